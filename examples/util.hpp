/*
 * examples/util.hpp -- small helpers shared by the example and benchmark programs:
 * HIP error checking, file reading and the chunk splitting rule of the reference's
 * harness (each file is cut independently into <= chunk_size pieces, the last one
 * short; reference: examples/util.h:51-95, benchmarks/benchmark_template_chunked.cuh:313-356).
 */
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#define HIP_CHECK(expr)                                                                                     \
  do {                                                                                                      \
    const hipError_t err_ = (expr);                                                                         \
    if (err_ != hipSuccess) {                                                                               \
      throw std::runtime_error(std::string("HIP failure '") + hipGetErrorString(err_) + "' at " + __FILE__ \
                               + ":" + std::to_string(__LINE__));                                           \
    }                                                                                                       \
  } while (0)

namespace util {

inline std::vector<char> read_file(const std::string& path)
{
  std::ifstream in(path, std::ios::binary | std::ios::ate);
  if (!in) {
    throw std::runtime_error("ERROR: Unable to open \"" + path + "\" for reading.");
  }
  const std::streamoff size = in.tellg();
  in.seekg(0);
  std::vector<char> data((size_t)size);
  in.read(data.data(), size);
  if (!in) {
    throw std::runtime_error("ERROR: Unable to read all of file \"" + path + "\".");
  }
  return data;
}

/* file = repeated {uint64 size, bytes}: every page is one chunk (-s flag of the harness) */
inline std::vector<std::vector<char>> read_pages(const std::string& path)
{
  std::ifstream in(path, std::ios::binary);
  if (!in) {
    throw std::runtime_error("ERROR: Unable to open \"" + path + "\" for reading.");
  }
  std::vector<std::vector<char>> pages;
  for (;;) {
    uint64_t n = 0;
    in.read(reinterpret_cast<char*>(&n), sizeof(n));
    if (!in) {
      break;
    }
    pages.emplace_back((size_t)n);
    in.read(pages.back().data(), (std::streamsize)n);
  }
  return pages;
}

inline std::vector<std::vector<char>> split_chunks(
    const std::vector<std::string>& files, size_t chunk_size, bool pages, size_t duplicate)
{
  std::vector<std::vector<char>> chunks;
  for (const std::string& f : files) {
    if (pages) {
      for (auto& p : read_pages(f)) {
        chunks.push_back(std::move(p));
      }
      continue;
    }
    const std::vector<char> data = read_file(f);
    for (size_t off = 0; off < data.size(); off += chunk_size) {
      const size_t n = std::min(chunk_size, data.size() - off);
      chunks.emplace_back(data.begin() + (std::ptrdiff_t)off, data.begin() + (std::ptrdiff_t)(off + n));
    }
  }
  if (duplicate > 1) {
    const size_t base = chunks.size();
    chunks.reserve(base * duplicate);
    for (size_t d = 1; d < duplicate; ++d) {
      for (size_t i = 0; i < base; ++i) {
        chunks.push_back(chunks[i]);
      }
    }
  }
  return chunks;
}

} // namespace util
