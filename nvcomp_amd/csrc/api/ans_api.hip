/*
 * api/ans_api.hip -- C ABI of the batched ANS codec (include/nvcomp/ans.h) and the kernels
 * it launches: one wavefront per chunk, one launch per *Async call on the caller's stream;
 * nothing here allocates or synchronises.
 */
#include <hip/hip_runtime.h>

#include "nvcomp/ans.h"

#include "common/log.h"

#include "ans/ans.hip.h"

namespace {

constexpr unsigned kWavesPerBlock = 4;
constexpr uint32_t kMaxOutCap = 1u << 26;

__global__ void __launch_bounds__(64 * kWavesPerBlock) ans_compress_kernel(
    const void* const* __restrict__ in_ptrs,
    const size_t* __restrict__ in_bytes,
    size_t max_chunk_bytes,
    size_t batch_size,
    void* const* __restrict__ out_ptrs,
    size_t* out_bytes)
{
  __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock][ans::kEncodeLds];
  const uint32_t w = wave::uniform(threadIdx.x >> 6);
  const size_t chunk = (size_t)blockIdx.x * kWavesPerBlock + w;
  if (chunk >= batch_size) {
    return;
  }
  const uint8_t* src = wave::uniform_ptr((const uint8_t*)in_ptrs[chunk]);
  uint8_t* dst = wave::uniform_ptr((uint8_t*)out_ptrs[chunk]);
  const size_t n64 = wave::uniform64(in_bytes[chunk]);
  /* a chunk larger than the caller declared would overrun the slot sized from GetMaxOutputChunkSize: refused (size 0) */
  const uint32_t produced = n64 > max_chunk_bytes ? 0u : ans::encode_chunk(src, (uint32_t)n64, dst, lds[w]);
  if (wave::lane_id() == 0) {
    out_bytes[chunk] = produced;
  }
}

__global__ void __launch_bounds__(64 * kWavesPerBlock) ans_decompress_kernel(
    const void* const* __restrict__ comp_ptrs,
    const size_t* __restrict__ comp_bytes,
    const size_t* out_caps,
    size_t* actual_bytes,
    size_t batch_size,
    void* const* __restrict__ out_ptrs,
    nvcompStatus_t* statuses)
{
  __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerBlock][ans::kDecodeLds];
  const uint32_t w = wave::uniform(threadIdx.x >> 6);
  const size_t chunk = (size_t)blockIdx.x * kWavesPerBlock + w;
  if (chunk >= batch_size) {
    return;
  }
  const uint8_t* in = wave::uniform_ptr((const uint8_t*)comp_ptrs[chunk]);
  uint8_t* out = wave::uniform_ptr((uint8_t*)out_ptrs[chunk]);
  const size_t in_len64 = wave::uniform64(comp_bytes[chunk]);
  size_t cap64 = wave::uniform64(out_caps[chunk]);
  if (cap64 > kMaxOutCap) {
    cap64 = kMaxOutCap;
  }
  uint32_t err = ans::kErrNone;
  uint32_t produced = 0;
  if (in_len64 > 0xffffffffull - 64) {
    err = ans::kErrInput;
  } else {
    produced = ans::decode_chunk(in, (uint32_t)in_len64, out, (uint32_t)cap64, lds[w], err);
  }
  if (wave::lane_id() == 0) {
    if (actual_bytes != nullptr) {
      actual_bytes[chunk] = err ? 0 : produced;
    }
    if (statuses != nullptr) {
      statuses[chunk] = err ? nvcompErrorCannotDecompress : nvcompSuccess;
    }
  }
}

__global__ void __launch_bounds__(256) ans_decompress_size_kernel(
    const void* const* __restrict__ comp_ptrs, const size_t* __restrict__ comp_bytes, size_t* out_bytes, size_t batch_size)
{
  const size_t chunk = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (chunk >= batch_size) {
    return;
  }
  const uint8_t* in = (const uint8_t*)comp_ptrs[chunk];
  size_t n = 0;
  if (comp_bytes[chunk] >= ans::kHeaderBytes && ans::load_as<uint32_t>(in) == 0x01534e41u) {
    n = ans::load_as<uint32_t>(in + 4);
  }
  out_bytes[chunk] = n;
}

void clear_stale_error()
{
  (void)hipGetLastError();
}

nvcompStatus_t launch_status()
{
  return hipGetLastError() == hipSuccess ? nvcompSuccess : nvcompErrorCudaError;
}

unsigned grid_for(size_t batch_size)
{
  return (unsigned)((batch_size + kWavesPerBlock - 1) / kWavesPerBlock);
}

bool opts_ok(nvcompBatchedANSOpts_t o)
{
  return o.type == nvcomp_rANS;
}

} // namespace

extern "C" {

nvcompStatus_t nvcompBatchedANSCompressGetTempSize(
    size_t /*batch_size*/, size_t max_uncompressed_chunk_bytes, nvcompBatchedANSOpts_t format_opts, size_t* temp_bytes)
{
  if (temp_bytes == nullptr || !opts_ok(format_opts)) {
    return nvcompErrorInvalidValue;
  }
  if (max_uncompressed_chunk_bytes > nvcompANSCompressionMaxAllowedChunkSize) {
    return nvcompErrorChunkSizeTooLarge;
  }
  *temp_bytes = 0; /* histogram and symbol table live in LDS */
  return nvcompSuccess;
}

nvcompStatus_t nvcompBatchedANSCompressGetMaxOutputChunkSize(
    size_t max_uncompressed_chunk_bytes, nvcompBatchedANSOpts_t format_opts, size_t* max_compressed_bytes)
{
  if (max_compressed_bytes == nullptr || !opts_ok(format_opts)) {
    return nvcompErrorInvalidValue;
  }
  if (max_uncompressed_chunk_bytes > nvcompANSCompressionMaxAllowedChunkSize) {
    return nvcompErrorChunkSizeTooLarge;
  }
  *max_compressed_bytes = ans::max_compressed_bytes(max_uncompressed_chunk_bytes);
  return nvcompSuccess;
}

nvcompStatus_t nvcompBatchedANSCompressAsync(
    const void* const* device_uncompressed_ptrs,
    const size_t* device_uncompressed_bytes,
    size_t max_uncompressed_chunk_bytes,
    size_t batch_size,
    void* /*device_temp_ptr*/,
    size_t /*temp_bytes*/,
    void* const* device_compressed_ptrs,
    size_t* device_compressed_bytes,
    nvcompBatchedANSOpts_t format_opts,
    hipStream_t stream)
{
  nvlog::call(3, "nvcompBatchedANSCompressAsync(batch_size=%zu, max_uncompressed_chunk_bytes=%zu, stream=%p)", batch_size,
              max_uncompressed_chunk_bytes, (void*)stream);
  if (!opts_ok(format_opts)) {
    return nvcompErrorInvalidValue;
  }
  if (max_uncompressed_chunk_bytes > nvcompANSCompressionMaxAllowedChunkSize) {
    return nvcompErrorChunkSizeTooLarge;
  }
  if (batch_size == 0) {
    return nvcompSuccess;
  }
  if (device_uncompressed_ptrs == nullptr || device_uncompressed_bytes == nullptr || device_compressed_ptrs == nullptr
      || device_compressed_bytes == nullptr) {
    return nvcompErrorInvalidValue;
  }
  clear_stale_error();
  hipLaunchKernelGGL(ans_compress_kernel, dim3(grid_for(batch_size)), dim3(64 * kWavesPerBlock), 0, stream,
                     device_uncompressed_ptrs, device_uncompressed_bytes, max_uncompressed_chunk_bytes, batch_size,
                     device_compressed_ptrs, device_compressed_bytes);
  return launch_status();
}

nvcompStatus_t nvcompBatchedANSDecompressGetTempSize(
    size_t /*num_chunks*/, size_t /*max_uncompressed_chunk_bytes*/, size_t* temp_bytes)
{
  if (temp_bytes == nullptr) {
    return nvcompErrorInvalidValue;
  }
  *temp_bytes = 0; /* the decode table lives in LDS */
  return nvcompSuccess;
}

nvcompStatus_t nvcompBatchedANSDecompressAsync(
    const void* const* device_compressed_ptrs,
    const size_t* device_compressed_bytes,
    const size_t* device_uncompressed_bytes,
    size_t* device_actual_uncompressed_bytes,
    size_t batch_size,
    void* const /*device_temp_ptr*/,
    size_t /*temp_bytes*/,
    void* const* device_uncompressed_ptrs,
    nvcompStatus_t* device_statuses,
    hipStream_t stream)
{
  nvlog::call(3, "nvcompBatchedANSDecompressAsync(batch_size=%zu, statuses=%s, actual_sizes=%s, stream=%p)", batch_size,
              device_statuses ? "yes" : "null", device_actual_uncompressed_bytes ? "yes" : "null", (void*)stream);
  if (batch_size == 0) {
    return nvcompSuccess;
  }
  if (device_compressed_ptrs == nullptr || device_compressed_bytes == nullptr || device_uncompressed_bytes == nullptr
      || device_uncompressed_ptrs == nullptr) {
    return nvcompErrorInvalidValue;
  }
  clear_stale_error();
  hipLaunchKernelGGL(ans_decompress_kernel, dim3(grid_for(batch_size)), dim3(64 * kWavesPerBlock), 0, stream,
                     device_compressed_ptrs, device_compressed_bytes, device_uncompressed_bytes,
                     device_actual_uncompressed_bytes, batch_size, device_uncompressed_ptrs, device_statuses);
  return launch_status();
}

nvcompStatus_t nvcompBatchedANSGetDecompressSizeAsync(
    const void* const* device_compressed_ptrs,
    const size_t* device_compressed_bytes,
    size_t* device_uncompressed_bytes,
    size_t batch_size,
    hipStream_t stream)
{
  if (batch_size == 0) {
    return nvcompSuccess;
  }
  if (device_compressed_ptrs == nullptr || device_compressed_bytes == nullptr || device_uncompressed_bytes == nullptr) {
    return nvcompErrorInvalidValue;
  }
  clear_stale_error();
  hipLaunchKernelGGL(ans_decompress_size_kernel, dim3((unsigned)((batch_size + 255) / 256)), dim3(256), 0, stream,
                     device_compressed_ptrs, device_compressed_bytes, device_uncompressed_bytes, batch_size);
  return launch_status();
}

nvcompStatus_t nvcompBatchedANSCompressGetTempSizeEx(
    size_t batch_size,
    size_t max_uncompressed_chunk_bytes,
    nvcompBatchedANSOpts_t format_opts,
    size_t* temp_bytes,
    const size_t /*max_total_uncompressed_bytes*/)
{
  /* the scratch does not depend on the batch's total size */
  return nvcompBatchedANSCompressGetTempSize(batch_size, max_uncompressed_chunk_bytes, format_opts, temp_bytes);
}

nvcompStatus_t nvcompBatchedANSDecompressGetTempSizeEx(
    size_t num_chunks, size_t max_uncompressed_chunk_bytes, size_t* temp_bytes, size_t /*max_total_uncompressed_bytes*/)
{
  return nvcompBatchedANSDecompressGetTempSize(num_chunks, max_uncompressed_chunk_bytes, temp_bytes);
}

} // extern "C"
