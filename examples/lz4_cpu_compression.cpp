/*
 * examples/lz4_cpu_compression.cpp -- compress on the CPU with liblz4 (LZ4_compress_HC,
 * level 12), decompress on the GPU with nvcompBatchedLZ4DecompressAsync, byte-compare.
 * This is the format pin of the reference (examples/lz4_cpu_compression.cu:35-160): the GPU
 * decoder must accept what the standard CPU compressor writes, from a tight-packed
 * (unaligned) compressed slab. Usage: lz4_cpu_compression -f FILE [FILE...]
 */
#include <cstring>
#include <iomanip>

#include <lz4.h>
#include <lz4hc.h>

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
      throw std::runtime_error("Usage: lz4_cpu_compression -f FILE [FILE...]");
    }
    const size_t chunk = 1 << 16;
    const auto chunks = util::split_chunks(files, chunk, false, 0);
    const size_t n = chunks.size();
    size_t total = 0, comp_total = 0;
    std::vector<std::vector<char>> comp(n);
    for (size_t i = 0; i < n; ++i) {
      total += chunks[i].size();
      comp[i].resize((size_t)LZ4_compressBound((int)chunks[i].size()));
      const int sz = LZ4_compress_HC(chunks[i].data(), comp[i].data(), (int)chunks[i].size(), (int)comp[i].size(), 12);
      if (sz <= 0) {
        throw std::runtime_error("LZ4 CPU failed to compress chunk " + std::to_string(i));
      }
      comp[i].resize((size_t)sz);
      comp_total += (size_t)sz;
    }
    std::cout << "----------" << std::endl;
    std::cout << "files: " << files.size() << std::endl;
    std::cout << "uncompressed (B): " << total << std::endl;
    std::cout << "chunks: " << n << std::endl;
    std::cout << "comp_size: " << comp_total << ", compressed ratio: " << std::fixed << std::setprecision(2)
              << (double)total / (double)comp_total << std::endl;
    /* tight-packed device slabs: compressed chunks start at arbitrary byte offsets */
    char *d_comp, *d_out;
    HIP_CHECK(hipMalloc((void**)&d_comp, comp_total));
    HIP_CHECK(hipMalloc((void**)&d_out, total));
    std::vector<void*> comp_ptrs(n), out_ptrs(n);
    std::vector<size_t> comp_sizes(n), out_sizes(n);
    size_t co = 0, oo = 0;
    for (size_t i = 0; i < n; ++i) {
      HIP_CHECK(hipMemcpy(d_comp + co, comp[i].data(), comp[i].size(), hipMemcpyHostToDevice));
      comp_ptrs[i] = d_comp + co;
      comp_sizes[i] = comp[i].size();
      out_ptrs[i] = d_out + oo;
      out_sizes[i] = chunks[i].size();
      co += comp[i].size();
      oo += chunks[i].size();
    }
    void **d_comp_ptrs, **d_out_ptrs, *d_temp;
    size_t *d_comp_sizes, *d_out_sizes, *d_actual, temp_bytes = 0;
    nvcompStatus_t* d_status;
    HIP_CHECK(hipMalloc((void**)&d_comp_ptrs, n * 8));
    HIP_CHECK(hipMalloc((void**)&d_out_ptrs, n * 8));
    HIP_CHECK(hipMalloc((void**)&d_comp_sizes, n * 8));
    HIP_CHECK(hipMalloc((void**)&d_out_sizes, n * 8));
    HIP_CHECK(hipMalloc((void**)&d_actual, n * 8));
    HIP_CHECK(hipMalloc((void**)&d_status, n * sizeof(nvcompStatus_t)));
    HIP_CHECK(hipMemcpy(d_comp_ptrs, comp_ptrs.data(), n * 8, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_out_ptrs, out_ptrs.data(), n * 8, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_comp_sizes, comp_sizes.data(), n * 8, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_out_sizes, out_sizes.data(), n * 8, hipMemcpyHostToDevice));
    if (nvcompBatchedLZ4DecompressGetTempSize(n, chunk, &temp_bytes) != nvcompSuccess) {
      throw std::runtime_error("nvcompBatchedLZ4DecompressGetTempSize() failed.");
    }
    HIP_CHECK(hipMalloc(&d_temp, temp_bytes ? temp_bytes : 1));
    hipStream_t stream;
    HIP_CHECK(hipStreamCreate(&stream));
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    for (int pass = 0; pass < 2; ++pass) { /* second pass is the timed one */
      HIP_CHECK(hipEventRecord(e0, stream));
      if (nvcompBatchedLZ4DecompressAsync(d_comp_ptrs, d_comp_sizes, d_out_sizes, d_actual, n, d_temp, temp_bytes, d_out_ptrs,
                                          d_status, stream) != nvcompSuccess) {
        throw std::runtime_error("nvcompBatchedLZ4DecompressAsync() not successful");
      }
      HIP_CHECK(hipEventRecord(e1, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
    }
    std::vector<nvcompStatus_t> status(n);
    std::vector<char> back(total);
    HIP_CHECK(hipMemcpy(status.data(), d_status, n * sizeof(nvcompStatus_t), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(back.data(), d_out, total, hipMemcpyDeviceToHost));
    oo = 0;
    for (size_t i = 0; i < n; ++i) {
      if (status[i] != nvcompSuccess || std::memcmp(back.data() + oo, chunks[i].data(), chunks[i].size()) != 0) {
        throw std::runtime_error("Failed to validate decompressed data (chunk " + std::to_string(i) + ")");
      }
      oo += chunks[i].size();
    }
    std::cout << "decompression validated :)" << std::endl;
    float ms;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::cout << "decompression throughput (GB/s): " << (double)total / 1.0e9 / (ms * 1.0e-3) << std::endl;
    return 0;
  } catch (const std::exception& e) {
    std::cerr << e.what() << std::endl;
    return 1;
  }
}
