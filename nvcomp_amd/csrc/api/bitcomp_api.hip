/*
 * api/bitcomp_api.hip -- C ABI of the batched Bitcomp codec (include/nvcomp/bitcomp.h) and
 * the kernels it launches: one wavefront per chunk, one launch per *Async call on the
 * caller's stream; nothing here allocates or synchronises.
 */
#include <hip/hip_runtime.h>

#include "nvcomp/bitcomp.h"

#include "common/log.h"

#include "bitcomp/bitcomp.hip.h"

namespace {

constexpr unsigned kWavesPerBlock = 4;
constexpr uint32_t kMaxOutCap = 1u << 26;

/* Workgroups per CU the register allocation aims at: 8 (64 registers) spills 12-20 registers of the 8-byte element types,
 * whose values travel as two parts; 6 (85 registers) does not and is 5-24 % faster on them (round 3, 1 GiB: int64 key column
 * 2 809 -> 3 015 GB/s, 64-bit view of the float columns 1 928 -> 2 390; 7: slower than both). */
#ifndef NVCOMP_BITCOMP_WGS64
#define NVCOMP_BITCOMP_WGS64 6
#endif
template <class T, bool DELTA>
__global__ void __launch_bounds__(64 * kWavesPerBlock, sizeof(T) == 8 ? NVCOMP_BITCOMP_WGS64 : 8) bitcomp_compress_kernel(
    const void* const* __restrict__ in_ptrs,
    const size_t* __restrict__ in_bytes,
    size_t max_chunk_bytes,
    size_t batch_size,
    void* const* __restrict__ out_ptrs,
    size_t* out_bytes)
{
  const size_t chunk = (size_t)blockIdx.x * kWavesPerBlock + wave::uniform(threadIdx.x >> 6);
  if (chunk >= batch_size) {
    return;
  }
  const uint8_t* src = wave::uniform_ptr((const uint8_t*)in_ptrs[chunk]);
  uint8_t* dst = wave::uniform_ptr((uint8_t*)out_ptrs[chunk]);
  const size_t n64 = wave::uniform64(in_bytes[chunk]);
  /* a chunk larger than the caller declared would overrun the slot sized from GetMaxOutputChunkSize: refused (size 0) */
  const uint32_t produced = n64 > max_chunk_bytes ? 0u : bitcomp::encode_chunk<T, DELTA>(src, (uint32_t)n64, dst);
  if (wave::lane_id() == 0) {
    out_bytes[chunk] = produced;
  }
}

template <bool CHECKED>
__global__ void __launch_bounds__(64 * kWavesPerBlock) bitcomp_decompress_kernel(
    const void* const* __restrict__ comp_ptrs,
    const size_t* __restrict__ comp_bytes,
    const size_t* out_caps,
    size_t* actual_bytes,
    size_t batch_size,
    void* const* __restrict__ out_ptrs,
    nvcompStatus_t* statuses)
{
  const size_t chunk = (size_t)blockIdx.x * kWavesPerBlock + wave::uniform(threadIdx.x >> 6);
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
  uint32_t err = bitcomp::kErrNone;
  uint32_t produced = 0;
  if (in_len64 > 0xffffffffull - 64) {
    err = bitcomp::kErrInput;
  } else {
    /* the stream is validated whether or not the caller wants statuses: the checks are a few scalar compares per block */
    produced = bitcomp::decode_chunk<true>(in, (uint32_t)in_len64, out, (uint32_t)cap64, err);
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

__global__ void __launch_bounds__(256) bitcomp_decompress_size_kernel(
    const void* const* __restrict__ comp_ptrs, const size_t* __restrict__ comp_bytes, size_t* out_bytes, size_t batch_size)
{
  const size_t chunk = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (chunk >= batch_size) {
    return;
  }
  const uint8_t* in = (const uint8_t*)comp_ptrs[chunk];
  size_t n = 0;
  if (comp_bytes[chunk] >= bitcomp::kHeaderBytes && bitcomp::load_u32(in) == 0x01435442u) {
    n = bitcomp::load_u32(in + 8);
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

/* element size for a type code, 0 when the code is not a Bitcomp type */
unsigned elem_size(nvcompType_t t)
{
  switch (t) {
  case NVCOMP_TYPE_CHAR:
  case NVCOMP_TYPE_UCHAR: return 1;
  case NVCOMP_TYPE_SHORT:
  case NVCOMP_TYPE_USHORT: return 2;
  case NVCOMP_TYPE_INT:
  case NVCOMP_TYPE_UINT: return 4;
  case NVCOMP_TYPE_LONGLONG:
  case NVCOMP_TYPE_ULONGLONG: return 8;
  default: return 0;
  }
}

bool opts_ok(nvcompBatchedBitcompFormatOpts o)
{
  return (o.algorithm_type == 0 || o.algorithm_type == 1) && elem_size(o.data_type) != 0;
}

} // namespace

extern "C" {

nvcompStatus_t nvcompBatchedBitcompCompressGetTempSize(
    size_t /*batch_size*/, size_t max_uncompressed_chunk_bytes, nvcompBatchedBitcompFormatOpts format_opts, size_t* temp_bytes)
{
  if (temp_bytes == nullptr || !opts_ok(format_opts)) {
    return nvcompErrorInvalidValue;
  }
  if (max_uncompressed_chunk_bytes > nvcompBitcompCompressionMaxAllowedChunkSize) {
    return nvcompErrorChunkSizeTooLarge;
  }
  *temp_bytes = 0; /* a block's values stay in registers */
  return nvcompSuccess;
}

nvcompStatus_t nvcompBatchedBitcompCompressGetMaxOutputChunkSize(
    size_t max_uncompressed_chunk_bytes, nvcompBatchedBitcompFormatOpts format_opts, size_t* max_compressed_bytes)
{
  if (max_compressed_bytes == nullptr || !opts_ok(format_opts)) {
    return nvcompErrorInvalidValue;
  }
  if (max_uncompressed_chunk_bytes > nvcompBitcompCompressionMaxAllowedChunkSize) {
    return nvcompErrorChunkSizeTooLarge;
  }
  const size_t bound = bitcomp::max_compressed_bytes(max_uncompressed_chunk_bytes, elem_size(format_opts.data_type));
  *max_compressed_bytes = (bound + 7) & ~(size_t)7; /* keeps consecutive output slots 8-byte aligned */
  return nvcompSuccess;
}

nvcompStatus_t nvcompBatchedBitcompCompressAsync(
    const void* const* device_uncompressed_ptrs,
    const size_t* device_uncompressed_bytes,
    size_t max_uncompressed_chunk_bytes,
    size_t batch_size,
    void* /*device_temp_ptr*/,
    size_t /*temp_bytes*/,
    void* const* device_compressed_ptrs,
    size_t* device_compressed_bytes,
    nvcompBatchedBitcompFormatOpts format_opts,
    hipStream_t stream)
{
  nvlog::call(3, "nvcompBatchedBitcompCompressAsync(batch_size=%zu, max_uncompressed_chunk_bytes=%zu, algo=%d, type=%d, stream=%p)",
              batch_size, max_uncompressed_chunk_bytes, format_opts.algorithm_type, (int)format_opts.data_type, (void*)stream);
  if (!opts_ok(format_opts)) {
    return nvcompErrorInvalidValue;
  }
  if (max_uncompressed_chunk_bytes > nvcompBitcompCompressionMaxAllowedChunkSize) {
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
  const dim3 grid(grid_for(batch_size));
  const dim3 block(64 * kWavesPerBlock);
#define NVCOMP_BITCOMP_LAUNCH(T, D)                                                                            \
  hipLaunchKernelGGL((bitcomp_compress_kernel<T, D>), grid, block, 0, stream, device_uncompressed_ptrs,        \
                     device_uncompressed_bytes, max_uncompressed_chunk_bytes, batch_size, device_compressed_ptrs, \
                     device_compressed_bytes)
  const bool delta = format_opts.algorithm_type == 0;
  switch (elem_size(format_opts.data_type)) {
  case 1:
    if (delta) {
      NVCOMP_BITCOMP_LAUNCH(uint8_t, true);
    } else {
      NVCOMP_BITCOMP_LAUNCH(uint8_t, false);
    }
    break;
  case 2:
    if (delta) {
      NVCOMP_BITCOMP_LAUNCH(uint16_t, true);
    } else {
      NVCOMP_BITCOMP_LAUNCH(uint16_t, false);
    }
    break;
  case 4:
    if (delta) {
      NVCOMP_BITCOMP_LAUNCH(uint32_t, true);
    } else {
      NVCOMP_BITCOMP_LAUNCH(uint32_t, false);
    }
    break;
  default:
    if (delta) {
      NVCOMP_BITCOMP_LAUNCH(uint64_t, true);
    } else {
      NVCOMP_BITCOMP_LAUNCH(uint64_t, false);
    }
    break;
  }
#undef NVCOMP_BITCOMP_LAUNCH
  return launch_status();
}

nvcompStatus_t nvcompBatchedBitcompDecompressGetTempSize(
    size_t /*num_chunks*/, size_t /*max_uncompressed_chunk_bytes*/, size_t* temp_bytes)
{
  if (temp_bytes == nullptr) {
    return nvcompErrorInvalidValue;
  }
  *temp_bytes = 0;
  return nvcompSuccess;
}

nvcompStatus_t nvcompBatchedBitcompDecompressAsync(
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
  nvlog::call(3, "nvcompBatchedBitcompDecompressAsync(batch_size=%zu, statuses=%s, actual_sizes=%s, stream=%p)", batch_size,
              device_statuses ? "yes" : "null", device_actual_uncompressed_bytes ? "yes" : "null", (void*)stream);
  if (batch_size == 0) {
    return nvcompSuccess;
  }
  if (device_compressed_ptrs == nullptr || device_compressed_bytes == nullptr || device_uncompressed_bytes == nullptr
      || device_uncompressed_ptrs == nullptr) {
    return nvcompErrorInvalidValue;
  }
  clear_stale_error();
  const dim3 grid(grid_for(batch_size));
  const dim3 block(64 * kWavesPerBlock);
  if (device_statuses != nullptr) {
    hipLaunchKernelGGL((bitcomp_decompress_kernel<true>), grid, block, 0, stream, device_compressed_ptrs,
                       device_compressed_bytes, device_uncompressed_bytes, device_actual_uncompressed_bytes, batch_size,
                       device_uncompressed_ptrs, device_statuses);
  } else {
    hipLaunchKernelGGL((bitcomp_decompress_kernel<false>), grid, block, 0, stream, device_compressed_ptrs,
                       device_compressed_bytes, device_uncompressed_bytes, device_actual_uncompressed_bytes, batch_size,
                       device_uncompressed_ptrs, device_statuses);
  }
  return launch_status();
}

nvcompStatus_t nvcompBatchedBitcompGetDecompressSizeAsync(
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
  hipLaunchKernelGGL(bitcomp_decompress_size_kernel, dim3((unsigned)((batch_size + 255) / 256)), dim3(256), 0, stream,
                     device_compressed_ptrs, device_compressed_bytes, device_uncompressed_bytes, batch_size);
  return launch_status();
}

nvcompStatus_t nvcompBatchedBitcompCompressGetTempSizeEx(
    size_t batch_size,
    size_t max_uncompressed_chunk_bytes,
    nvcompBatchedBitcompFormatOpts format_opts,
    size_t* temp_bytes,
    const size_t /*max_total_uncompressed_bytes*/)
{
  /* the scratch does not depend on the batch's total size */
  return nvcompBatchedBitcompCompressGetTempSize(batch_size, max_uncompressed_chunk_bytes, format_opts, temp_bytes);
}

nvcompStatus_t nvcompBatchedBitcompDecompressGetTempSizeEx(
    size_t num_chunks, size_t max_uncompressed_chunk_bytes, size_t* temp_bytes, size_t /*max_total_uncompressed_bytes*/)
{
  return nvcompBatchedBitcompDecompressGetTempSize(num_chunks, max_uncompressed_chunk_bytes, temp_bytes);
}

} // extern "C"
