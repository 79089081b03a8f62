/* benchmarks/benchmark_common.hpp -- synthetic generator shared by the *_synth programs:
 * uniform bytes in [0, max_byte] from a seeded mt19937 (the shape of the reference's
 * gen_data, benchmarks/benchmark_common.h:158-175). */
#pragma once

#include <cstdint>
#include <random>
#include <vector>

namespace bench {

inline std::vector<uint8_t> gen_data(int max_byte, size_t size, std::mt19937& rng)
{
  std::uniform_int_distribution<uint16_t> dist(0, (uint16_t)max_byte);
  std::vector<uint8_t> v(size);
  for (auto& b : v) {
    b = (uint8_t)(dist(rng) & 0xff);
  }
  return v;
}

} // namespace bench
