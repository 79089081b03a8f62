/*
 * examples/nvcomp_storage.cpp -- the storage round trip of the reference's examples/nvcomp_gds.cu:97-283: generate
 * data on the device, compress it with nvcomp::LZ4Manager, write the (4 KiB-aligned) compressed buffer to a file opened
 * O_DIRECT, wipe the device buffer, read the file back, decompress, compare on the device.
 *
 * The reference moves the bytes between NVMe and device memory with cuFile (GPUDirect Storage). This image has no
 * hipFile / GDS driver, so the file side goes through two pinned host buffers: O_DIRECT pread / pwrite of one 8 MiB
 * piece overlaps the hipMemcpyAsync of the other (the page cache is bypassed either way; a peer-to-peer DMA path
 * drops in where `transfer()` is). The phases carry roctx ranges where the reference has NVTX ranges
 * (nvcomp_gds.cu:54,129-280): `rocprofv3 --marker-trace -- nvcomp_storage ...` shows them; libroctx64 is looked up at
 * run time, without it the ranges are no-ops. Usage: nvcomp_storage <filename> [bytes]
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <dlfcn.h>
#include <fcntl.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "nvcomp/lz4.hpp"
#include "util.hpp"

using namespace nvcomp;

__global__ void initialize(uint8_t* data, size_t n)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    data[i] = (uint8_t)((i >> 6) * 31 + (i & 7)); /* short repeats a byte-oriented LZ finds */
  }
}

__global__ void compare(const uint8_t* a, const uint8_t* b, int* invalid, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    if (a[i] != b[i]) {
      *invalid = 1;
    }
  }
}

/* roctxRangePushA / roctxRangePop of libroctx64.so (ROCm's NVTX), resolved at run time */
struct Ranges
{
  int (*push_fn)(const char*) = nullptr;
  int (*pop_fn)() = nullptr;
  Ranges()
  {
    if (void* h = dlopen("libroctx64.so", RTLD_LAZY | RTLD_GLOBAL)) {
      push_fn = (int (*)(const char*))dlsym(h, "roctxRangePushA");
      pop_fn = (int (*)())dlsym(h, "roctxRangePop");
    }
  }
  void push(const char* name) const
  {
    if (push_fn && pop_fn) {
      push_fn(name);
    }
  }
  void pop() const
  {
    if (push_fn && pop_fn) {
      pop_fn();
    }
  }
};

constexpr size_t kPiece = 8u << 20; /* multiple of 4096 */

/* device <-> file through two pinned buffers; `to_file`: device -> file. Returns the bytes moved, -1 on an I/O error. */
static ssize_t transfer(int fd, uint8_t* d_buf, size_t bytes, bool to_file, hipStream_t streams[2], uint8_t* pinned[2])
{
  size_t done = 0;
  int slot = 0;
  size_t pending[2] = {0, 0}, pending_at[2] = {0, 0};
  while (done < bytes || pending[0] || pending[1]) {
    if (pending[slot]) { /* finish what this slot started two turns ago */
      HIP_CHECK(hipStreamSynchronize(streams[slot]));
      if (to_file && pwrite(fd, pinned[slot], pending[slot], (off_t)pending_at[slot]) != (ssize_t)pending[slot]) {
        return -1;
      }
      pending[slot] = 0;
    }
    if (done < bytes) {
      const size_t now = bytes - done < kPiece ? bytes - done : kPiece;
      if (to_file) {
        HIP_CHECK(hipMemcpyAsync(pinned[slot], d_buf + done, now, hipMemcpyDeviceToHost, streams[slot]));
      } else {
        if (pread(fd, pinned[slot], now, (off_t)done) != (ssize_t)now) {
          return -1;
        }
        HIP_CHECK(hipMemcpyAsync(d_buf + done, pinned[slot], now, hipMemcpyHostToDevice, streams[slot]));
      }
      pending[slot] = now;
      pending_at[slot] = done;
      done += now;
    }
    slot ^= 1;
  }
  return (ssize_t)bytes;
}

int main(int argc, char** argv)
{
  if (argc < 2) {
    printf("Argument: %s <filename> [bytes]\n", argv[0]);
    return -1;
  }
  try {
    const char* filename = argv[1];
    int fd = open(filename, O_RDWR | O_TRUNC | O_CREAT | O_DIRECT, 0666);
    bool direct = true;
    if (fd == -1) { /* tmpfs and some overlay file systems refuse O_DIRECT */
      fd = open(filename, O_RDWR | O_TRUNC | O_CREAT, 0666);
      direct = false;
    }
    if (fd == -1) {
      printf("Error, cannot create the file: %s\n", filename);
      return -1;
    }
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, 0));
    printf("Using device: %s%s\n", prop.name, direct ? "" : " (file system without O_DIRECT: buffered I/O)");
    const size_t n = argc > 2 ? strtoull(argv[2], nullptr, 10) : 100000000;

    const Ranges ranges;
    ranges.push("Compressor setup");
    uint8_t *d_input, *d_output, *d_compressed;
    hipStream_t stream, io[2];
    HIP_CHECK(hipMalloc((void**)&d_input, n));
    HIP_CHECK(hipMalloc((void**)&d_output, n));
    HIP_CHECK(hipStreamCreate(&stream));
    HIP_CHECK(hipStreamCreate(&io[0]));
    HIP_CHECK(hipStreamCreate(&io[1]));
    uint8_t* pinned[2];
    HIP_CHECK(hipHostMalloc((void**)&pinned[0], kPiece, hipHostMallocDefault)); /* page-aligned: fine for O_DIRECT */
    HIP_CHECK(hipHostMalloc((void**)&pinned[1], kPiece, hipHostMallocDefault));
    hipLaunchKernelGGL(initialize, dim3((unsigned)((n - 1) / 512 + 1)), dim3(512), 0, stream, d_input, n);

    LZ4Manager compressor(1 << 16, nvcompBatchedLZ4Opts_t{NVCOMP_TYPE_CHAR}, stream, 0);
    const CompressionConfig comp_config = compressor.configure_compression(n);
    size_t lcompbuf = comp_config.max_compressed_buffer_size;
    lcompbuf = ((lcompbuf - 1) / 4096 + 1) * 4096; /* O_DIRECT wants whole 4 KiB blocks */
    HIP_CHECK(hipMalloc((void**)&d_compressed, lcompbuf));
    ranges.pop();

    ranges.push("Compression");
    compressor.compress(d_input, d_compressed, comp_config);
    const size_t compressed_size = compressor.get_compressed_output_size(d_compressed);
    const size_t aligned = ((compressed_size - 1) / 4096 + 1) * 4096;
    printf("Data compressed from %zu Bytes to %zu Bytes, aligned to %zu Bytes\n", n, compressed_size, aligned);
    ranges.pop();

    ranges.push("File write");
    ssize_t nb = transfer(fd, d_compressed, aligned, true, io, pinned);
    if (nb != (ssize_t)aligned) {
      printf("Error, write returned %zd instead of %zu\n", nb, aligned);
      return -1;
    }
    printf("Wrote %zd bytes to file %s\n", nb, filename);
    ranges.pop();

    HIP_CHECK(hipMemsetAsync(d_compressed, 0xff, compressed_size, stream)); /* nothing of the compressed data survives on the device */
    HIP_CHECK(hipStreamSynchronize(stream));

    ranges.push("File read");
    nb = transfer(fd, d_compressed, aligned, false, io, pinned);
    if (nb != (ssize_t)aligned) {
      printf("Error, read returned %zd instead of %zu\n", nb, aligned);
      return -1;
    }
    HIP_CHECK(hipStreamSynchronize(io[0]));
    HIP_CHECK(hipStreamSynchronize(io[1]));
    printf("Read %zd bytes from file %s\n", nb, filename);
    ranges.pop();

    ranges.push("Decompression");
    /* a fresh manager, configured from the bytes that came back from the file */
    const DecompressionConfig decomp_config = compressor.configure_decompression(d_compressed);
    if (decomp_config.decomp_data_size != n) {
      printf("Error: Uncompressed size does not match the original size\n");
      return -1;
    }
    int* dh_invalid;
    HIP_CHECK(hipHostMalloc((void**)&dh_invalid, sizeof(int), hipHostMallocDefault));
    *dh_invalid = 0;
    printf("Decompressing\n");
    compressor.decompress(d_output, d_compressed, decomp_config);
    hipLaunchKernelGGL(compare, dim3(2 * (unsigned)prop.multiProcessorCount), dim3(1024), 0, stream, d_input, d_output, dh_invalid, n);
    HIP_CHECK(hipStreamSynchronize(stream));
    const bool ok = *dh_invalid == 0;
    ranges.pop();
    printf(ok ? "PASSED: Uncompressed data is identical to the input\n" : "FAILED: Uncompressed data does not match the original\n");
    close(fd);
    unlink(filename);
    HIP_CHECK(hipHostFree(dh_invalid));
    HIP_CHECK(hipHostFree(pinned[0]));
    HIP_CHECK(hipHostFree(pinned[1]));
    HIP_CHECK(hipFree(d_input));
    HIP_CHECK(hipFree(d_output));
    HIP_CHECK(hipFree(d_compressed));
    printf("All done, exiting...\n");
    return ok ? 0 : 1;
  } catch (const std::exception& e) {
    fprintf(stderr, "%s\n", e.what());
    return 1;
  }
}
