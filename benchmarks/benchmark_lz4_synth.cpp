/*
 * benchmark_lz4_synth -- LZ4Manager on synthetic buffers of 64 KiB * 2^b bytes, b = 0..B:
 * all zeros and uniform random bytes, round trip verified (reference program:
 * benchmarks/benchmark_lz4_synth.cpp:64-72, rng mt19937(0); B = 13 there, -b to shorten).
 */
#include <iomanip>

#include "benchmark_common.hpp"
#include "nvcomp.hpp"
#include "../examples/util.hpp"

using namespace nvcomp;

static void run(LZ4Manager& m, hipStream_t stream, const std::vector<uint8_t>& data, const char* label)
{
  const size_t n = data.size();
  uint8_t *d_in, *d_comp, *d_out;
  HIP_CHECK(hipMalloc((void**)&d_in, n));
  HIP_CHECK(hipMalloc((void**)&d_out, n));
  HIP_CHECK(hipMemcpy(d_in, data.data(), n, hipMemcpyHostToDevice));
  CompressionConfig cc = m.configure_compression(n);
  HIP_CHECK(hipMalloc((void**)&d_comp, cc.max_compressed_buffer_size));
  hipEvent_t e0, e1, e2;
  HIP_CHECK(hipEventCreate(&e0));
  HIP_CHECK(hipEventCreate(&e1));
  HIP_CHECK(hipEventCreate(&e2));
  m.compress(d_in, d_comp, cc); /* warm-up */
  HIP_CHECK(hipEventRecord(e0, stream));
  m.compress(d_in, d_comp, cc);
  HIP_CHECK(hipEventRecord(e1, stream));
  DecompressionConfig dc = m.configure_decompression(cc);
  m.decompress(d_out, d_comp, dc);
  HIP_CHECK(hipEventRecord(e2, stream));
  const size_t comp = m.get_compressed_output_size(d_comp);
  float c_ms, d_ms;
  HIP_CHECK(hipEventElapsedTime(&c_ms, e0, e1));
  HIP_CHECK(hipEventElapsedTime(&d_ms, e1, e2));
  std::vector<uint8_t> back(n);
  HIP_CHECK(hipMemcpy(back.data(), d_out, n, hipMemcpyDeviceToHost));
  if (*dc.get_status() != nvcompSuccess || back != data) {
    throw std::runtime_error(std::string("ERROR: round trip failed for ") + label);
  }
  std::cout << std::fixed << std::setprecision(3) << label << " " << std::setw(10) << n << " B  ratio "
            << (double)n / (double)comp << "  comp " << n / 1.0e6 / c_ms << " GB/s  decomp " << n / 1.0e6 / d_ms << " GB/s"
            << std::endl;
  (void)hipFree(d_in);
  (void)hipFree(d_comp);
  (void)hipFree(d_out);
}

int main(int argc, char** argv)
{
  try {
    int max_b = 13, gpu = 0;
    for (int i = 1; i + 1 < argc; i += 2) {
      const std::string f = argv[i];
      if (f == "-b") max_b = std::atoi(argv[i + 1]);
      else if (f == "-g") gpu = std::atoi(argv[i + 1]);
    }
    HIP_CHECK(hipSetDevice(gpu));
    hipStream_t stream;
    HIP_CHECK(hipStreamCreate(&stream));
    {
      LZ4Manager manager{1 << 16, nvcompBatchedLZ4DefaultOpts, stream, gpu};
      std::mt19937 rng(0);
      for (int b = 0; b <= max_b; ++b) {
        const size_t n = (size_t)65536 << b;
        run(manager, stream, std::vector<uint8_t>(n, 0), "zeros ");
        run(manager, stream, bench::gen_data(255, n, rng), "random");
      }
    }
    HIP_CHECK(hipStreamDestroy(stream));
    return 0;
  } catch (const std::exception& e) {
    std::cerr << e.what() << std::endl;
    return 1;
  }
}
