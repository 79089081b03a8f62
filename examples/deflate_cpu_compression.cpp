/*
 * examples/deflate_cpu_compression.cpp -- compress on the CPU with zlib, decompress on the GPU with
 * nvcompBatchedDeflateDecompressAsync (raw DEFLATE streams) or nvcompBatchedGzipDecompressAsync (gzip members),
 * byte-compare. The format pins of the reference: examples/deflate_cpu_compression.cu:58-104 (algo 1: compress2 with
 * the zlib wrapper cut off; algo 2: deflateInit2(9, windowBits -15)) and examples/gzip_gpu_decompression.cu:57-81
 * (deflateInit2(9, windowBits 15 | 16)); algo 0 is libdeflate at level 6 (:60-67), the reference's default.
 * Usage: deflate_cpu_compression [-a 0|1|2|gzip] -f FILE [FILE...]
 */
#include <cstring>
#include <iomanip>

#include <libdeflate.h>
#include <zlib.h>

#include "nvcomp/deflate.h"
#include "nvcomp/gzip.h"
#include "util.hpp"

static std::vector<char> zlib_stream(const std::vector<char>& in, int window_bits)
{
  z_stream zs;
  std::memset(&zs, 0, sizeof(zs));
  if (deflateInit2(&zs, 9, Z_DEFLATED, window_bits, 8, Z_DEFAULT_STRATEGY) != Z_OK) {
    throw std::runtime_error("Call to deflateInit2 failed");
  }
  std::vector<char> out(deflateBound(&zs, (uLong)in.size()) + 32);
  zs.next_in = (Bytef*)in.data();
  zs.avail_in = (uInt)in.size();
  zs.next_out = (Bytef*)out.data();
  zs.avail_out = (uInt)out.size();
  if (deflate(&zs, Z_FINISH) != Z_STREAM_END) {
    throw std::runtime_error("Deflate operation failed");
  }
  out.resize(zs.total_out);
  deflateEnd(&zs);
  return out;
}

int main(int argc, char** argv)
{
  try {
    std::vector<std::string> files;
    std::string algo = "2";
    for (int i = 1; i < argc; ++i) {
      if (std::string(argv[i]) == "-f") {
        while (i + 1 < argc && argv[i + 1][0] != '-') files.push_back(argv[++i]);
      } else if (std::string(argv[i]) == "-a" && i + 1 < argc) {
        algo = argv[++i];
      }
    }
    if (files.empty() || (algo != "0" && algo != "1" && algo != "2" && algo != "gzip")) {
      throw std::runtime_error("Usage: deflate_cpu_compression [-a 0|1|2|gzip] -f FILE [FILE...]");
    }
    const bool gz = algo == "gzip";
    const size_t chunk = 1 << 16;
    const auto chunks = util::split_chunks(files, chunk, false, 0);
    const size_t n = chunks.size();
    size_t total = 0, comp_total = 0;
    std::vector<std::vector<char>> comp(n);
    for (size_t i = 0; i < n; ++i) {
      total += chunks[i].size();
      if (algo == "0") { /* libdeflate, level 6 (examples/deflate_cpu_compression.cu:60-67) */
        libdeflate_compressor* c = libdeflate_alloc_compressor(6);
        if (c == nullptr) {
          throw std::runtime_error("libdeflate_alloc_compressor failed");
        }
        comp[i].resize(libdeflate_deflate_compress_bound(c, chunks[i].size()));
        const size_t len = libdeflate_deflate_compress(c, chunks[i].data(), chunks[i].size(), comp[i].data(), comp[i].size());
        libdeflate_free_compressor(c);
        if (len == 0) {
          throw std::runtime_error("libdeflate_deflate_compress failed to compress chunk " + std::to_string(i) + ".");
        }
        comp[i].resize(len);
      } else if (algo == "1") { /* compress2, then the 2-byte header and the 4-byte Adler-32 are dropped */
        uLongf len = compressBound((uLong)chunks[i].size());
        std::vector<char> z(len);
        if (compress2((Bytef*)z.data(), &len, (const Bytef*)chunks[i].data(), (uLong)chunks[i].size(), 9) != Z_OK) {
          throw std::runtime_error("ZLIB compress() failed");
        }
        comp[i].assign(z.begin() + 2, z.begin() + (len - 4));
      } else {
        comp[i] = zlib_stream(chunks[i], gz ? (15 | 16) : -15);
      }
      comp_total += comp[i].size();
    }
    std::cout << "----------" << std::endl;
    std::cout << "files: " << files.size() << std::endl;
    std::cout << "uncompressed (B): " << total << std::endl;
    std::cout << "chunks: " << n << std::endl;
    std::cout << "comp_size: " << comp_total << ", compressed ratio: " << std::fixed << std::setprecision(2)
              << (double)total / (double)comp_total << std::endl;
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
    const nvcompStatus_t ts = gz ? nvcompBatchedGzipDecompressGetTempSize(n, chunk, &temp_bytes)
                                 : nvcompBatchedDeflateDecompressGetTempSize(n, chunk, &temp_bytes);
    if (ts != nvcompSuccess) {
      throw std::runtime_error("DecompressGetTempSize() failed.");
    }
    HIP_CHECK(hipMalloc(&d_temp, temp_bytes ? temp_bytes : 1));
    hipStream_t stream;
    HIP_CHECK(hipStreamCreate(&stream));
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    for (int pass = 0; pass < 2; ++pass) { /* second pass is the timed one */
      HIP_CHECK(hipEventRecord(e0, stream));
      const nvcompStatus_t rc =
          gz ? nvcompBatchedGzipDecompressAsync(d_comp_ptrs, d_comp_sizes, d_out_sizes, d_actual, n, d_temp, temp_bytes, d_out_ptrs,
                                                d_status, stream)
             : nvcompBatchedDeflateDecompressAsync(d_comp_ptrs, d_comp_sizes, d_out_sizes, d_actual, n, d_temp, temp_bytes,
                                                   d_out_ptrs, d_status, stream);
      if (rc != nvcompSuccess) {
        throw std::runtime_error("DecompressAsync() not successful");
      }
      HIP_CHECK(hipEventRecord(e1, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
    }
    std::vector<nvcompStatus_t> status(n);
    std::vector<size_t> actual(n);
    std::vector<char> back(total);
    HIP_CHECK(hipMemcpy(status.data(), d_status, n * sizeof(nvcompStatus_t), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(actual.data(), d_actual, n * 8, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(back.data(), d_out, total, hipMemcpyDeviceToHost));
    oo = 0;
    for (size_t i = 0; i < n; ++i) {
      if (status[i] != nvcompSuccess || actual[i] != chunks[i].size()
          || std::memcmp(back.data() + oo, chunks[i].data(), chunks[i].size()) != 0) {
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
