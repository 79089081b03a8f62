/*
 * examples/low_level_quickstart_example.cpp -- the batched low-level C API end to end on
 * 1,000,000 pseudo-random bytes in 64 KiB chunks: compress, ask the library for the
 * decompressed sizes (nvcompBatchedLZ4GetDecompressSizeAsync), then decompress IN PLACE
 * over the original input buffers and compare with the host copy. Mirrors the flow of the
 * reference's example (examples/low_level_quickstart_example.cpp:36-160).
 */
#include <cstring>
#include <random>

#include "nvcomp/lz4.h"
#include "util.hpp"

static void check(nvcompStatus_t s, const char* what)
{
  if (s != nvcompSuccess) {
    throw std::runtime_error(std::string(what) + " failed with status " + std::to_string((int)s));
  }
}

int main()
{
  try {
    const size_t in_bytes = 1000000, chunk = 1 << 16;
    std::vector<char> host(in_bytes);
    std::mt19937 gen(42);
    std::uniform_int_distribution<short> dist(0, 255);
    for (size_t i = 0; i < in_bytes; ++i) {
      host[i] = (i / 3000) % 2 ? (char)dist(gen) : (char)('a' + (i % 7));
    }
    hipStream_t stream;
    HIP_CHECK(hipStreamCreate(&stream));
    const size_t batch = (in_bytes + chunk - 1) / chunk;
    char* d_in;
    HIP_CHECK(hipMalloc((void**)&d_in, in_bytes));
    HIP_CHECK(hipMemcpyAsync(d_in, host.data(), in_bytes, hipMemcpyHostToDevice, stream));
    std::vector<void*> h_in_ptrs(batch);
    std::vector<size_t> h_in_sizes(batch);
    for (size_t i = 0; i < batch; ++i) {
      h_in_ptrs[i] = d_in + i * chunk;
      h_in_sizes[i] = std::min(chunk, in_bytes - i * chunk);
    }
    void** d_in_ptrs;
    size_t* d_in_sizes;
    HIP_CHECK(hipMalloc((void**)&d_in_ptrs, batch * sizeof(void*)));
    HIP_CHECK(hipMalloc((void**)&d_in_sizes, batch * sizeof(size_t)));
    HIP_CHECK(hipMemcpyAsync(d_in_ptrs, h_in_ptrs.data(), batch * sizeof(void*), hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(d_in_sizes, h_in_sizes.data(), batch * sizeof(size_t), hipMemcpyHostToDevice, stream));
    /* compress */
    size_t temp_bytes = 0, max_out = 0;
    check(nvcompBatchedLZ4CompressGetTempSize(batch, chunk, nvcompBatchedLZ4DefaultOpts, &temp_bytes), "CompressGetTempSize");
    check(nvcompBatchedLZ4CompressGetMaxOutputChunkSize(chunk, nvcompBatchedLZ4DefaultOpts, &max_out), "CompressGetMaxOutputChunkSize");
    void* d_temp;
    HIP_CHECK(hipMalloc(&d_temp, temp_bytes ? temp_bytes : 1));
    char* d_comp;
    HIP_CHECK(hipMalloc((void**)&d_comp, batch * max_out));
    std::vector<void*> h_comp_ptrs(batch);
    for (size_t i = 0; i < batch; ++i) {
      h_comp_ptrs[i] = d_comp + i * max_out;
    }
    void** d_comp_ptrs;
    size_t* d_comp_sizes;
    HIP_CHECK(hipMalloc((void**)&d_comp_ptrs, batch * sizeof(void*)));
    HIP_CHECK(hipMalloc((void**)&d_comp_sizes, batch * sizeof(size_t)));
    HIP_CHECK(hipMemcpyAsync(d_comp_ptrs, h_comp_ptrs.data(), batch * sizeof(void*), hipMemcpyHostToDevice, stream));
    check(nvcompBatchedLZ4CompressAsync(d_in_ptrs, d_in_sizes, chunk, batch, d_temp, temp_bytes, d_comp_ptrs, d_comp_sizes,
                                        nvcompBatchedLZ4DefaultOpts, stream),
          "CompressAsync");
    /* the chunks carry no metadata and could be stored or shuffled here; ask for their decompressed sizes */
    size_t* d_sizes_back;
    HIP_CHECK(hipMalloc((void**)&d_sizes_back, batch * sizeof(size_t)));
    check(nvcompBatchedLZ4GetDecompressSizeAsync(d_comp_ptrs, d_comp_sizes, d_sizes_back, batch, stream), "GetDecompressSizeAsync");
    size_t dtemp_bytes = 0;
    check(nvcompBatchedLZ4DecompressGetTempSize(batch, chunk, &dtemp_bytes), "DecompressGetTempSize");
    void* d_dtemp;
    HIP_CHECK(hipMalloc(&d_dtemp, dtemp_bytes ? dtemp_bytes : 1));
    nvcompStatus_t* d_status;
    HIP_CHECK(hipMalloc((void**)&d_status, batch * sizeof(nvcompStatus_t)));
    /* wipe the input, then decompress in place over it */
    HIP_CHECK(hipMemsetAsync(d_in, 0, in_bytes, stream));
    check(nvcompBatchedLZ4DecompressAsync(d_comp_ptrs, d_comp_sizes, d_sizes_back, d_sizes_back, batch, d_dtemp, dtemp_bytes,
                                          d_in_ptrs, d_status, stream),
          "DecompressAsync");
    HIP_CHECK(hipStreamSynchronize(stream));
    std::vector<size_t> sizes_back(batch);
    std::vector<nvcompStatus_t> status(batch);
    std::vector<char> back(in_bytes);
    HIP_CHECK(hipMemcpy(sizes_back.data(), d_sizes_back, batch * sizeof(size_t), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(status.data(), d_status, batch * sizeof(nvcompStatus_t), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(back.data(), d_in, in_bytes, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < batch; ++i) {
      if (status[i] != nvcompSuccess || sizes_back[i] != h_in_sizes[i]) {
        throw std::runtime_error("chunk " + std::to_string(i) + " did not decompress correctly");
      }
    }
    if (back != host) {
      throw std::runtime_error("decompressed data differs from the input");
    }
    std::cout << "low_level_quickstart_example: " << batch << " chunks round-tripped in place" << std::endl;
    return 0;
  } catch (const std::exception& e) {
    std::cerr << e.what() << std::endl;
    return 1;
  }
}
