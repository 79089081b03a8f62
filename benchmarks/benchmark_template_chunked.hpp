/*
 * benchmarks/benchmark_template_chunked.hpp -- HIP-native harness behind the
 * benchmark_<format>_chunked programs. Same command line, same stdout and the same
 * measurement rules as the reference's harness
 * (benchmarks/benchmark_template_chunked.cuh; flag table doc/Benchmarks.md:43-54):
 *   - every file is cut into <= chunk_size pieces; all pieces form one batch
 *   - events bracket exactly one *Async call; throughput = uncompressed bytes / time
 *   - every iteration checks per-chunk status and size, the last one every byte
 * so scripts written against the reference (benchmarks/benchmark.sh:24-34 awk the
 * "compressed ratio:" / "compression throughput" / "decompression throughput" lines)
 * work unchanged.
 */
#pragma once

#include <algorithm>
#include <cstring>
#include <functional>
#include <iomanip>
#include <numeric>
#include <sstream>

#include "nvcomp.h"
#include "../examples/util.hpp"

namespace bench {

inline void require(bool ok, const std::string& msg)
{
  if (!ok) {
    throw std::runtime_error("ERROR: " + msg);
  }
}

inline void nv(nvcompStatus_t s, const char* what)
{
  require(s == nvcompSuccess, std::string(what) + " failed with status " + std::to_string((int)s));
}

/* One device slab holding a batch of chunks + device arrays of pointers and sizes. */
class DeviceBatch
{
public:
  /* upload host chunks; starts aligned to `align` bytes (the reference aligns to 8) */
  DeviceBatch(const std::vector<std::vector<char>>& chunks, size_t align)
  {
    count_ = chunks.size();
    std::vector<size_t> offs(count_ + 1, 0);
    std::vector<size_t> sizes(count_);
    for (size_t i = 0; i < count_; ++i) {
      sizes[i] = chunks[i].size();
      offs[i + 1] = (offs[i] + sizes[i] + align - 1) / align * align;
    }
    allocate(offs.back(), offs, sizes);
    for (size_t i = 0; i < count_; ++i) {
      if (sizes[i]) {
        HIP_CHECK(hipMemcpy(slab_ + offs[i], chunks[i].data(), sizes[i], hipMemcpyHostToDevice));
      }
    }
  }
  /* `count` empty slots of `slot` bytes each; sizes preset to the slot size */
  DeviceBatch(size_t slot, size_t count)
  {
    count_ = count;
    std::vector<size_t> offs(count + 1), sizes(count, slot);
    for (size_t i = 0; i <= count; ++i) {
      offs[i] = i * slot;
    }
    allocate(slot * count, offs, sizes);
  }
  ~DeviceBatch()
  {
    (void)hipFree(slab_);
    (void)hipFree(ptrs_);
    (void)hipFree(sizes_);
  }
  DeviceBatch(const DeviceBatch&) = delete;
  DeviceBatch& operator=(const DeviceBatch&) = delete;

  void** ptrs() { return ptrs_; }
  size_t* sizes() { return sizes_; }
  uint8_t* data() { return slab_; }
  size_t size() const { return count_; }
  const std::vector<size_t>& offsets() const { return offsets_; }

private:
  void allocate(size_t bytes, const std::vector<size_t>& offs, const std::vector<size_t>& sizes)
  {
    HIP_CHECK(hipMalloc((void**)&slab_, bytes ? bytes : 1));
    HIP_CHECK(hipMalloc((void**)&ptrs_, sizeof(void*) * (count_ ? count_ : 1)));
    HIP_CHECK(hipMalloc((void**)&sizes_, sizeof(size_t) * (count_ ? count_ : 1)));
    std::vector<void*> p(count_);
    for (size_t i = 0; i < count_; ++i) {
      p[i] = slab_ + offs[i];
    }
    offsets_.assign(offs.begin(), offs.end());
    if (count_) {
      HIP_CHECK(hipMemcpy(ptrs_, p.data(), sizeof(void*) * count_, hipMemcpyHostToDevice));
      HIP_CHECK(hipMemcpy(sizes_, sizes.data(), sizeof(size_t) * count_, hipMemcpyHostToDevice));
    }
  }
  uint8_t* slab_ = nullptr;
  void** ptrs_ = nullptr;
  size_t* sizes_ = nullptr;
  size_t count_ = 0;
  std::vector<size_t> offsets_;
};

/* The six entry points of one format, options bound. */
struct Codec
{
  std::function<nvcompStatus_t(size_t, size_t, size_t*)> compress_temp_size;
  std::function<nvcompStatus_t(size_t, size_t*)> max_output_chunk_size;
  std::function<nvcompStatus_t(const void* const*, const size_t*, size_t, size_t, void*, size_t, void* const*, size_t*,
                               hipStream_t)>
      compress_async;
  std::function<nvcompStatus_t(size_t, size_t, size_t*)> decompress_temp_size;
  std::function<nvcompStatus_t(const void* const*, const size_t*, const size_t*, size_t*, size_t, void*, size_t,
                               void* const*, nvcompStatus_t*, hipStream_t)>
      decompress_async;
  /* returns false (after printing why) when the inputs cannot be used with the options */
  std::function<bool(const std::vector<std::vector<char>>&)> input_valid;
};

struct Args
{
  int gpu = 0;
  std::vector<std::string> files;
  size_t warmup = 1;
  size_t iterations = 1;
  size_t duplicate = 0;
  bool csv = false;
  bool tab = false;
  bool pages = false;
  size_t chunk_size = 65536;
};

inline bool parse_bool(const std::string& v)
{
  if (v == "true") {
    return true;
  }
  if (v == "false") {
    return false;
  }
  throw std::runtime_error("ERROR: expected true or false, got \"" + v + "\"");
}

inline void usage(const char* prog, const std::string& extra)
{
  std::cout << "Usage: " << prog << " [OPTIONS]\n"
            << "  -f, --input_file F...        input file(s) (required)\n"
            << "  -g, --gpu N                  device to use (default 0)\n"
            << "  -w, --warmup_count N         unreported warm-up iterations (default 1)\n"
            << "  -i, --iteration_count N      timed iterations to average (default 1)\n"
            << "  -x, --duplicate_data K       clone the chunk list K times\n"
            << "  -c, --csv_output true|false  one CSV row instead of text\n"
            << "  -e, --tab_separator true|false\n"
            << "  -s, --file_with_page_sizes true|false  file = repeated {uint64 size, bytes}\n"
            << "  -p, --chunk_size N           split size in bytes (default 65536)\n"
            << extra;
}

/* handle_extra(flag_short, flag_long, value) consumes format-specific options. */
inline Args parse_args(int argc, char** argv, const std::string& extra_usage,
                       const std::function<bool(const std::string&, const std::string&)>& handle_extra)
{
  Args a;
  for (int i = 1; i < argc; ++i) {
    const std::string flag = argv[i];
    if (flag == "-?" || flag == "--help") {
      usage(argv[0], extra_usage);
      std::exit(0);
    }
    if (flag == "-f" || flag == "--input_file") {
      while (i + 1 < argc && argv[i + 1][0] != '-') {
        a.files.push_back(argv[++i]);
      }
      continue;
    }
    if (i + 1 >= argc) {
      throw std::runtime_error("ERROR: missing value for " + flag);
    }
    const std::string val = argv[++i];
    if (flag == "-g" || flag == "--gpu") {
      a.gpu = std::atoi(val.c_str());
    } else if (flag == "-w" || flag == "--warmup_count") {
      a.warmup = std::strtoull(val.c_str(), nullptr, 10);
    } else if (flag == "-i" || flag == "--iteration_count") {
      a.iterations = std::strtoull(val.c_str(), nullptr, 10);
    } else if (flag == "-x" || flag == "--duplicate_data") {
      a.duplicate = std::strtoull(val.c_str(), nullptr, 10);
    } else if (flag == "-c" || flag == "--csv_output") {
      a.csv = parse_bool(val);
    } else if (flag == "-e" || flag == "--tab_separator") {
      a.tab = parse_bool(val);
    } else if (flag == "-s" || flag == "--file_with_page_sizes") {
      a.pages = parse_bool(val);
    } else if (flag == "-p" || flag == "--chunk_size") {
      a.chunk_size = std::strtoull(val.c_str(), nullptr, 10);
    } else if (!handle_extra(flag, val)) {
      usage(argv[0], extra_usage);
      throw std::runtime_error("ERROR: unknown option " + flag);
    }
  }
  if (a.files.empty()) {
    usage(argv[0], extra_usage);
    throw std::runtime_error("ERROR: Must specify at least one input file.");
  }
  return a;
}

/* One pass: `count` iterations of compress + decompress; prints unless warm-up. */
inline void run(const Codec& codec, const std::vector<std::vector<char>>& chunks, const Args& a, bool warmup,
                size_t count)
{
  if (count == 0) {
    return;
  }
  require(codec.input_valid(chunks), "Input is not valid for the chosen format options.");
  size_t total_bytes = 0, max_chunk = 0;
  for (const auto& c : chunks) {
    total_bytes += c.size();
    max_chunk = std::max(max_chunk, c.size());
  }
  const size_t n = chunks.size();
  hipStream_t stream;
  HIP_CHECK(hipStreamCreate(&stream));
  hipEvent_t start, stop;
  HIP_CHECK(hipEventCreate(&start));
  HIP_CHECK(hipEventCreate(&stop));
  DeviceBatch input(chunks, 8);
  std::vector<float> comp_ms, decomp_ms;
  std::vector<size_t> comp_sizes(n);
  size_t comp_bytes = 0;
  for (size_t it = 0; it < count; ++it) {
    /* ---- compression ---- */
    size_t ctemp_bytes = 0;
    nv(codec.compress_temp_size(n, max_chunk, &ctemp_bytes), "CompressGetTempSize");
    void* ctemp = nullptr;
    HIP_CHECK(hipMalloc(&ctemp, ctemp_bytes ? ctemp_bytes : 1));
    size_t max_out = 0;
    nv(codec.max_output_chunk_size(max_chunk, &max_out), "CompressGetMaxOutputChunkSize");
    max_out = (max_out + 7) / 8 * 8;
    DeviceBatch compressed(max_out, n);
    HIP_CHECK(hipEventRecord(start, stream));
    nv(codec.compress_async(input.ptrs(), input.sizes(), max_chunk, n, ctemp, ctemp_bytes, compressed.ptrs(),
                            compressed.sizes(), stream),
       "CompressAsync");
    HIP_CHECK(hipEventRecord(stop, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, start, stop));
    comp_ms.push_back(ms);
    HIP_CHECK(hipFree(ctemp));
    HIP_CHECK(hipMemcpy(comp_sizes.data(), compressed.sizes(), sizeof(size_t) * n, hipMemcpyDeviceToHost));
    comp_bytes = std::accumulate(comp_sizes.begin(), comp_sizes.end(), (size_t)0);
    for (size_t i = 0; i < n; ++i) {
      require(comp_sizes[i] <= max_out, "compressed chunk " + std::to_string(i) + " exceeds the declared bound");
    }
    /* ---- decompression: exact-size output slots, statuses and sizes requested ---- */
    size_t dtemp_bytes = 0;
    nv(codec.decompress_temp_size(n, max_chunk, &dtemp_bytes), "DecompressGetTempSize");
    void* dtemp = nullptr;
    HIP_CHECK(hipMalloc(&dtemp, dtemp_bytes ? dtemp_bytes : 1));
    size_t* d_actual = nullptr;
    nvcompStatus_t* d_status = nullptr;
    HIP_CHECK(hipMalloc((void**)&d_actual, sizeof(size_t) * n));
    HIP_CHECK(hipMalloc((void**)&d_status, sizeof(nvcompStatus_t) * n));
    std::vector<std::vector<char>> shape(n);
    for (size_t i = 0; i < n; ++i) {
      shape[i].resize(chunks[i].size());
    }
    DeviceBatch output(shape, 8); /* zero-filled slots of the exact sizes */
    HIP_CHECK(hipEventRecord(start, stream));
    nv(codec.decompress_async(compressed.ptrs(), compressed.sizes(), input.sizes(), d_actual, n, dtemp, dtemp_bytes,
                              output.ptrs(), d_status, stream),
       "DecompressAsync");
    HIP_CHECK(hipEventRecord(stop, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    HIP_CHECK(hipEventElapsedTime(&ms, start, stop));
    decomp_ms.push_back(ms);
    std::vector<size_t> actual(n);
    std::vector<nvcompStatus_t> status(n);
    HIP_CHECK(hipMemcpy(actual.data(), d_actual, sizeof(size_t) * n, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(status.data(), d_status, sizeof(nvcompStatus_t) * n, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) {
      require(status[i] == nvcompSuccess, "chunk " + std::to_string(i) + " status " + std::to_string((int)status[i]));
      require(actual[i] == chunks[i].size(), "chunk " + std::to_string(i) + " decompressed to the wrong size");
    }
    if (it + 1 == count) { /* last iteration: every byte */
      std::vector<char> back;
      for (size_t i = 0; i < n; ++i) {
        back.resize(chunks[i].size());
        if (!back.empty()) {
          HIP_CHECK(hipMemcpy(back.data(), output.data() + output.offsets()[i], back.size(), hipMemcpyDeviceToHost));
        }
        require(back == chunks[i], "chunk " + std::to_string(i) + " differs after the round trip");
      }
    }
    HIP_CHECK(hipFree(dtemp));
    HIP_CHECK(hipFree(d_actual));
    HIP_CHECK(hipFree(d_status));
  }
  HIP_CHECK(hipEventDestroy(start));
  HIP_CHECK(hipEventDestroy(stop));
  HIP_CHECK(hipStreamDestroy(stream));
  if (warmup) {
    return;
  }
  const double mean_c = std::accumulate(comp_ms.begin(), comp_ms.end(), 0.0) / comp_ms.size();
  const double mean_d = std::accumulate(decomp_ms.begin(), decomp_ms.end(), 0.0) / decomp_ms.size();
  const double ratio = (double)total_bytes / (double)comp_bytes;
  const double comp_gbs = (double)total_bytes / 1.0e9 / (mean_c * 1.0e-3);
  const double decomp_gbs = (double)total_bytes / 1.0e9 / (mean_d * 1.0e-3);
  if (!a.csv) {
    std::cout << "----------" << std::endl;
    std::cout << "files: " << a.files.size() << std::endl;
    std::cout << "uncompressed (B): " << total_bytes << std::endl;
    std::cout << "comp_size: " << comp_bytes << ", compressed ratio: " << std::fixed << std::setprecision(4) << ratio
              << std::endl;
    std::cout << "compression throughput (GB/s): " << comp_gbs << std::endl;
    std::cout << "decompression throughput (GB/s): " << decomp_gbs << std::endl;
  } else {
    const char* sep = a.tab ? "\t" : ", ";
    std::cout << "Files" << sep << "Duplicate data" << sep << "Size in MB" << sep << "Pages" << sep
              << "Avg page size in KB" << sep << "Max page size in KB" << sep << "Ucompressed size in bytes" << sep
              << "Compressed size in bytes" << sep << "Compression ratio" << sep
              << "Compression throughput (uncompressed) in GB/s" << sep
              << "Decompression throughput (uncompressed) in GB/s" << std::endl;
    std::cout << a.files.size() << sep << a.duplicate << sep << total_bytes / 1.0e6 << sep << n << sep
              << (n ? (double)total_bytes / n / 1.0e3 : 0.0) << sep << max_chunk / 1.0e3 << sep << total_bytes << sep
              << comp_bytes << sep << std::fixed << std::setprecision(2) << ratio << sep << comp_gbs << sep << decomp_gbs
              << std::endl;
  }
}

inline int main_chunked(int argc, char** argv, const std::string& extra_usage,
                        const std::function<bool(const std::string&, const std::string&)>& handle_extra,
                        const std::function<Codec(size_t chunk_size)>& make_codec)
{
  try {
    const Args a = parse_args(argc, argv, extra_usage, handle_extra);
    HIP_CHECK(hipSetDevice(a.gpu));
    const auto chunks = util::split_chunks(a.files, a.chunk_size, a.pages, a.duplicate);
    const Codec codec = make_codec(a.chunk_size);
    run(codec, chunks, a, true, a.warmup);
    run(codec, chunks, a, false, a.iterations);
    return 0;
  } catch (const std::exception& e) {
    std::cerr << e.what() << std::endl;
    return 1;
  }
}

} // namespace bench
