/*
 * api/pack_api.hip -- nvcompAmdBatchedPackAsync (include/nvcomp/amd_ext.h): the chunks of a batch, which the
 * compressors leave one per worst-case-sized slot, moved into ONE contiguous buffer with a device-side prefix sum of
 * their sizes. It is what a caller does between "compress" and "send": the reference's all-gather benchmark ships the
 * whole slots (benchmarks/benchmark_allgather.cpp:322-361 copies max-size buffers), the managers gather inside the
 * container (hlif/manager.hip); this entry point gives the low-level interface the same step without a host round
 * trip or a host-side loop.
 */
#include <hip/hip_runtime.h>

#include "nvcomp/amd_ext.h"

#include "common/log.h"
#include "common/wave.h"

namespace {

/* One workgroup: offsets[i] = sum of bytes[0..i), offsets[n] = total. */
__global__ void __launch_bounds__(256) pack_scan_kernel(const size_t* __restrict__ bytes, size_t n, size_t* offsets)
{
  __shared__ unsigned long long partial[256];
  const size_t per = (n + 255) / 256;
  const size_t lo = (size_t)threadIdx.x * per < n ? (size_t)threadIdx.x * per : n;
  const size_t hi = lo + per < n ? lo + per : n;
  unsigned long long sum = 0;
  for (size_t i = lo; i < hi; ++i) {
    sum += bytes[i];
  }
  partial[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long run = 0;
    for (int t = 0; t < 256; ++t) {
      const unsigned long long v = partial[t];
      partial[t] = run;
      run += v;
    }
    offsets[n] = run;
  }
  __syncthreads();
  unsigned long long run = partial[threadIdx.x];
  for (size_t i = lo; i < hi; ++i) {
    offsets[i] = run;
    run += bytes[i];
  }
}

/* One wavefront per chunk: 16-byte lane loads, stores at whatever alignment the running offset has. A chunk that
 * would end behind `capacity` is not copied (the caller sized the buffer from the declared bound: cannot happen
 * with sizes a compressor wrote). */
__global__ void __launch_bounds__(256) pack_copy_kernel(
    const void* const* __restrict__ ptrs, const size_t* __restrict__ bytes, size_t n, uint8_t* packed, size_t capacity,
    const size_t* __restrict__ offsets)
{
  const size_t i = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) {
    return;
  }
  const uint32_t lane = threadIdx.x & 63;
  const uint8_t* src = (const uint8_t*)ptrs[i];
  const size_t len = bytes[i];
  const size_t at = offsets[i];
  if (at + len > capacity) {
    return;
  }
  uint8_t* dst = packed + at;
  size_t k = (size_t)lane * 16;
  if (((uintptr_t)src & 15u) == 0) {
    for (; k + 16 <= len; k += 1024) {
      const wave::u32x4 v = wave::gload_u32x4_aligned(src + k);
      __builtin_memcpy(dst + k, &v, 16);
    }
  } else {
    for (; k + 16 <= len; k += 1024) {
      const wave::u32x4 v = wave::gload_u32x4(src + k);
      __builtin_memcpy(dst + k, &v, 16);
    }
  }
  /* the last len % 16 bytes (k of the lane that owns them points at them; every other lane is past the end) */
  const size_t tail = len & ~(size_t)15;
  for (size_t b = tail + lane; b < len; b += 64) {
    dst[b] = src[b];
  }
}

} // namespace

extern "C" nvcompStatus_t nvcompAmdBatchedPackAsync(
    const void* const* device_chunk_ptrs, const size_t* device_chunk_bytes, size_t batch_size, void* device_packed,
    size_t packed_capacity, size_t* device_offsets, hipStream_t stream)
{
  nvlog::call(3, "nvcompAmdBatchedPackAsync(batch_size=%zu, capacity=%zu, stream=%p)", batch_size, packed_capacity, (void*)stream);
  if (device_offsets == nullptr || (batch_size != 0 && (device_chunk_ptrs == nullptr || device_chunk_bytes == nullptr || device_packed == nullptr))) {
    return nvcompErrorInvalidValue;
  }
  (void)hipGetLastError();
  hipLaunchKernelGGL(pack_scan_kernel, dim3(1), dim3(256), 0, stream, device_chunk_bytes, batch_size, device_offsets);
  if (batch_size != 0) {
    hipLaunchKernelGGL(pack_copy_kernel, dim3((unsigned)((batch_size + 3) / 4)), dim3(256), 0, stream, device_chunk_ptrs,
                       device_chunk_bytes, batch_size, (uint8_t*)device_packed, packed_capacity, device_offsets);
  }
  return hipGetLastError() == hipSuccess ? nvcompSuccess : nvcompErrorCudaError;
}
