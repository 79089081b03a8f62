/*
 * examples/lz4_cpu_decompression.cpp -- compress on the GPU with
 * nvcompBatchedLZ4CompressAsync, decompress every chunk on the CPU with liblz4's
 * LZ4_decompress_safe, byte-compare (reference: examples/lz4_cpu_decompression.cu:47-160:
 * the GPU compressor's output must be a legal LZ4 block for the standard CPU decoder).
 * Usage: lz4_cpu_decompression -f FILE [FILE...]
 */
#include <cstring>
#include <iomanip>

#include <lz4.h>

#include "nvcomp/lz4.h"
#include "util.hpp"

int main(int argc, char** argv)
{
  try {
    std::vector<std::string> files;
    for (int i = 1; i < argc; ++i) {
      if (std::string(argv[i]) == "-f") {
        while (i + 1 < argc && argv[i + 1][0] != '-') files.push_back(argv[++i]);
      }
    }
    if (files.empty()) {
      throw std::runtime_error("Usage: lz4_cpu_decompression -f FILE [FILE...]");
    }
    const size_t chunk = 1 << 16;
    const auto chunks = util::split_chunks(files, chunk, false, 0);
    const size_t n = chunks.size();
    size_t total = 0;
    for (const auto& c : chunks) total += c.size();
    std::cout << "----------" << std::endl;
    std::cout << "files: " << files.size() << std::endl;
    std::cout << "uncompressed (B): " << total << std::endl;
    std::cout << "chunks: " << n << std::endl;
    size_t temp_bytes = 0, max_out = 0;
    if (nvcompBatchedLZ4CompressGetTempSize(n, chunk, nvcompBatchedLZ4DefaultOpts, &temp_bytes) != nvcompSuccess
        || nvcompBatchedLZ4CompressGetMaxOutputChunkSize(chunk, nvcompBatchedLZ4DefaultOpts, &max_out) != nvcompSuccess) {
      throw std::runtime_error("size query failed");
    }
    char *d_in, *d_comp;
    HIP_CHECK(hipMalloc((void**)&d_in, n * chunk));
    HIP_CHECK(hipMalloc((void**)&d_comp, n * max_out));
    std::vector<void*> in_ptrs(n), comp_ptrs(n);
    std::vector<size_t> in_sizes(n);
    for (size_t i = 0; i < n; ++i) {
      HIP_CHECK(hipMemcpy(d_in + i * chunk, chunks[i].data(), chunks[i].size(), hipMemcpyHostToDevice));
      in_ptrs[i] = d_in + i * chunk;
      comp_ptrs[i] = d_comp + i * max_out;
      in_sizes[i] = chunks[i].size();
    }
    void **d_in_ptrs, **d_comp_ptrs, *d_temp;
    size_t *d_in_sizes, *d_comp_sizes;
    HIP_CHECK(hipMalloc((void**)&d_in_ptrs, n * 8));
    HIP_CHECK(hipMalloc((void**)&d_comp_ptrs, n * 8));
    HIP_CHECK(hipMalloc((void**)&d_in_sizes, n * 8));
    HIP_CHECK(hipMalloc((void**)&d_comp_sizes, n * 8));
    HIP_CHECK(hipMalloc(&d_temp, temp_bytes ? temp_bytes : 1));
    HIP_CHECK(hipMemcpy(d_in_ptrs, in_ptrs.data(), n * 8, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_comp_ptrs, comp_ptrs.data(), n * 8, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_in_sizes, in_sizes.data(), n * 8, hipMemcpyHostToDevice));
    hipStream_t stream;
    HIP_CHECK(hipStreamCreate(&stream));
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    HIP_CHECK(hipEventRecord(e0, stream));
    if (nvcompBatchedLZ4CompressAsync(d_in_ptrs, d_in_sizes, chunk, n, d_temp, temp_bytes, d_comp_ptrs, d_comp_sizes,
                                      nvcompBatchedLZ4DefaultOpts, stream) != nvcompSuccess) {
      throw std::runtime_error("nvcompBatchedLZ4CompressAsync() failed.");
    }
    HIP_CHECK(hipEventRecord(e1, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    std::vector<size_t> comp_sizes(n);
    HIP_CHECK(hipMemcpy(comp_sizes.data(), d_comp_sizes, n * 8, hipMemcpyDeviceToHost));
    size_t comp_total = 0;
    std::vector<char> comp(max_out), out(chunk);
    for (size_t i = 0; i < n; ++i) {
      comp_total += comp_sizes[i];
      HIP_CHECK(hipMemcpy(comp.data(), d_comp + i * max_out, comp_sizes[i], hipMemcpyDeviceToHost));
      const int got = chunks[i].empty() ? 0
                                        : LZ4_decompress_safe(comp.data(), out.data(), (int)comp_sizes[i], (int)chunk);
      if (got < 0 || (size_t)got != chunks[i].size() || std::memcmp(out.data(), chunks[i].data(), chunks[i].size()) != 0) {
        throw std::runtime_error("LZ4 CPU failed to decompress chunk " + std::to_string(i) + ".");
      }
    }
    float ms;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::cout << "comp_size: " << comp_total << ", compressed ratio: " << std::fixed << std::setprecision(2)
              << (double)total / (double)comp_total << std::endl;
    std::cout << "compression throughput (GB/s): " << (double)total / 1.0e9 / (ms * 1.0e-3) << std::endl;
    std::cout << "decompression validated :)" << std::endl;
    return 0;
  } catch (const std::exception& e) {
    std::cerr << e.what() << std::endl;
    return 1;
  }
}
