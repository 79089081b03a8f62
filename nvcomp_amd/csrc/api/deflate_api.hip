/*
 * api/deflate_api.hip -- C ABI of the batched DEFLATE and gzip codecs (include/nvcomp/deflate.h, include/nvcomp/gzip.h)
 * and the kernels it launches. Host side does argument checks and one launch per *Async call on the caller's
 * stream; nothing here allocates or synchronises.
 */
#include <hip/hip_runtime.h>

#include <stdlib.h>

#include "nvcomp/deflate.h"
#include "nvcomp/gzip.h"

#include "common/log.h"

#include "deflate/deflate_decode.hip.h"
#include "deflate/deflate_encode_dynamic.hip.h"

namespace {

/* One chunk per wave. A wave's LDS (window, stream ring, literal ring, decoding and jump tables:
 * deflate::kLdsPerWave = 9.1 KiB) allows 17 waves per CU. Workgroups of ONE wave: chunks of a mixed batch take very
 * different times and a workgroup's LDS is only released when its last wave ends (1 wave: 71.6 GB/s, 2: 66.1). */
#ifndef NVCOMP_DEFLATE_WAVES_PER_BLOCK
#define NVCOMP_DEFLATE_WAVES_PER_BLOCK 1
#endif
constexpr unsigned kDecWaves = NVCOMP_DEFLATE_WAVES_PER_BLOCK;
constexpr unsigned kDecWavesPerSimd = deflate::kLdsPerWave <= 8192 ? 5 : deflate::kLdsPerWave <= 10240 ? 4 : 3;
constexpr unsigned kEncWaves = 2; /* 10 / 11.3 KiB of LDS per wave: 16 / 14 waves per CU in workgroups of two */
constexpr uint32_t kMaxOutCap = 1u << 26;

/* The bounds and match-offset checks always run: DEFLATE / gzip streams come from outside (CPU producers, files), and a
 * corrupt one must never write outside its output slot or read in front of it -- whether or not the caller asked for
 * statuses (they cost two ballots per batch of 64 records). Only the status WRITE depends on `statuses`. */
template <uint32_t FLAGS>
__global__ void __launch_bounds__(64 * kDecWaves, kDecWavesPerSimd) deflate_decompress_kernel(
    const void* const* __restrict__ comp_ptrs,
    const size_t* __restrict__ comp_bytes,
    const size_t* out_caps,
    size_t* actual_bytes,
    size_t batch_size,
    void* const* __restrict__ out_ptrs,
    nvcompStatus_t* statuses)
{
  __shared__ __attribute__((aligned(16))) uint8_t lds[kDecWaves][deflate::kLdsPerWave];
  const uint32_t w = wave::uniform(threadIdx.x >> 6);
  const size_t chunk = (size_t)blockIdx.x * kDecWaves + w;
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
  uint32_t err = lz::kErrNone;
  uint32_t produced = 0;
  if (in_len64 > (1u << 28)) {
    err = lz::kErrInput;
  } else {
#ifdef NVCOMP_LZW_PROF
    lzw::prof_begin();
#endif
    produced = deflate::decode_chunk<true, false>(in, (uint32_t)in_len64, out, (uint32_t)cap64, lds[w], FLAGS, err);
#ifdef NVCOMP_LZW_PROF
    lzw::prof_end();
#endif
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

/* Size query of raw DEFLATE: the symbols are decoded and counted (the format carries no length). */
__global__ void __launch_bounds__(64 * kDecWaves, kDecWavesPerSimd) deflate_size_kernel(
    const void* const* __restrict__ comp_ptrs, const size_t* __restrict__ comp_bytes, size_t* uncompressed_bytes, size_t batch_size)
{
  __shared__ __attribute__((aligned(16))) uint8_t lds[kDecWaves][deflate::kLdsPerWave];
  const uint32_t w = wave::uniform(threadIdx.x >> 6);
  const size_t chunk = (size_t)blockIdx.x * kDecWaves + w;
  if (chunk >= batch_size) {
    return;
  }
  const uint8_t* in = wave::uniform_ptr((const uint8_t*)comp_ptrs[chunk]);
  const size_t in_len64 = wave::uniform64(comp_bytes[chunk]);
  uint32_t err = lz::kErrInput;
  uint32_t produced = 0;
  if (in_len64 <= (1u << 28)) {
    produced = deflate::decode_chunk<true, true>(in, (uint32_t)in_len64, nullptr, 0, lds[w], 0, err);
  }
  if (wave::lane_id() == 0) {
    uncompressed_bytes[chunk] = err ? 0 : produced;
  }
}

/* Size query of gzip: ISIZE, the last four bytes of a member. */
__global__ void __launch_bounds__(256) gzip_size_kernel(
    const void* const* __restrict__ comp_ptrs, const size_t* __restrict__ comp_bytes, size_t* uncompressed_bytes, size_t batch_size)
{
  const size_t chunk = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (chunk >= batch_size) {
    return;
  }
  const uint8_t* in = (const uint8_t*)comp_ptrs[chunk];
  const size_t n = comp_bytes[chunk];
  size_t size = 0;
  if (n >= 18 && in[0] == 0x1f && in[1] == 0x8b && in[2] == 8) {
    size = (size_t)in[n - 4] | ((size_t)in[n - 3] << 8) | ((size_t)in[n - 2] << 16) | ((size_t)in[n - 1] << 24);
  }
  uncompressed_bytes[chunk] = size;
}

/* DYNAMIC: per-chunk Huffman codes (nvcompBatchedDeflateOpts_t.algo >= 1: two runs of the match finder and a code
 * construction per chunk) instead of the fixed code (algo 0). */
template <bool DYNAMIC>
__global__ void __launch_bounds__(64 * kEncWaves, DYNAMIC ? 3 : 4) deflate_compress_kernel(
    const void* const* __restrict__ in_ptrs,
    const size_t* __restrict__ in_bytes,
    size_t max_chunk_bytes,
    size_t batch_size,
    void* const* __restrict__ out_ptrs,
    size_t* out_bytes)
{
  __shared__ __attribute__((aligned(16))) uint8_t lds[kEncWaves][DYNAMIC ? deflate::kDynLdsPerWave : deflate::kEncLdsPerWave];
  const uint32_t w = wave::uniform(threadIdx.x >> 6);
  const size_t chunk = (size_t)blockIdx.x * kEncWaves + w;
  if (chunk >= batch_size) {
    return;
  }
  const uint8_t* src = wave::uniform_ptr((const uint8_t*)in_ptrs[chunk]);
  uint8_t* dst = wave::uniform_ptr((uint8_t*)out_ptrs[chunk]);
  const size_t n64 = wave::uniform64(in_bytes[chunk]);
  /* a chunk larger than the caller declared would overrun the output slot sized from GetMaxOutputChunkSize: it is
   * not compressed, its size reads 0 */
  const uint32_t produced = n64 > max_chunk_bytes ? 0u
                            : DYNAMIC ? deflate::encode_chunk_dynamic(src, (uint32_t)n64, dst, lds[w])
                                      : deflate::encode_chunk(src, (uint32_t)n64, dst, lds[w]);
  if (wave::lane_id() == 0) {
    out_bytes[chunk] = produced;
  }
}

/* hipGetLastError() is sticky per host thread: an unrelated earlier runtime call of the application must not be
 * reported as this launch's failure, so the slate is cleared before launching. */
void clear_stale_error()
{
  (void)hipGetLastError();
}

nvcompStatus_t launch_status()
{
  return hipGetLastError() == hipSuccess ? nvcompSuccess : nvcompErrorCudaError;
}

bool deflate_opts_ok(nvcompBatchedDeflateOpts_t opts)
{
  return opts.algo >= 0 && opts.algo <= 2;
}

template <uint32_t FLAGS>
nvcompStatus_t decompress_async(
    const char* name,
    const void* const* device_compressed_ptrs,
    const size_t* device_compressed_bytes,
    const size_t* device_uncompressed_bytes,
    size_t* device_actual_uncompressed_bytes,
    size_t batch_size,
    void* const* device_uncompressed_ptrs,
    nvcompStatus_t* device_statuses,
    hipStream_t stream)
{
  nvlog::call(3, "%s(batch_size=%zu, statuses=%s, actual_sizes=%s, stream=%p)", name, batch_size,
              device_statuses ? "yes" : "null", device_actual_uncompressed_bytes ? "yes" : "null", (void*)stream);
  if (batch_size == 0) {
    return nvcompSuccess;
  }
  if (device_compressed_ptrs == nullptr || device_compressed_bytes == nullptr || device_uncompressed_bytes == nullptr
      || device_uncompressed_ptrs == nullptr) {
    return nvcompErrorInvalidValue;
  }
  clear_stale_error();
  const dim3 grid((unsigned)((batch_size + kDecWaves - 1) / kDecWaves));
  const dim3 block(64 * kDecWaves);
  hipLaunchKernelGGL((deflate_decompress_kernel<FLAGS>), grid, block, 0, stream, device_compressed_ptrs,
                     device_compressed_bytes, device_uncompressed_bytes, device_actual_uncompressed_bytes, batch_size,
                     device_uncompressed_ptrs, device_statuses);
  return launch_status();
}

} // namespace

extern "C" {

nvcompStatus_t nvcompBatchedDeflateDecompressGetTempSize(
    size_t /*num_chunks*/, size_t /*max_uncompressed_chunk_bytes*/, size_t* temp_bytes)
{
  if (temp_bytes == nullptr) {
    return nvcompErrorInvalidValue;
  }
  *temp_bytes = 0; /* tables, rings and window live in LDS */
  return nvcompSuccess;
}

nvcompStatus_t nvcompBatchedDeflateDecompressGetTempSizeEx(
    size_t num_chunks, size_t max_uncompressed_chunk_bytes, size_t* temp_bytes, size_t /*max_total_uncompressed_bytes*/)
{
  return nvcompBatchedDeflateDecompressGetTempSize(num_chunks, max_uncompressed_chunk_bytes, temp_bytes);
}

nvcompStatus_t nvcompBatchedDeflateDecompressAsync(
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
  return decompress_async<0>("nvcompBatchedDeflateDecompressAsync", device_compressed_ptrs, device_compressed_bytes,
                             device_uncompressed_bytes, device_actual_uncompressed_bytes, batch_size,
                             device_uncompressed_ptrs, device_statuses, stream);
}

nvcompStatus_t nvcompBatchedDeflateGetDecompressSizeAsync(
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
  hipLaunchKernelGGL(deflate_size_kernel, dim3((unsigned)((batch_size + kDecWaves - 1) / kDecWaves)), dim3(64 * kDecWaves), 0,
                     stream, device_compressed_ptrs, device_compressed_bytes, device_uncompressed_bytes, batch_size);
  return launch_status();
}

nvcompStatus_t nvcompBatchedGzipDecompressGetTempSize(
    size_t num_chunks, size_t max_uncompressed_chunk_bytes, size_t* temp_bytes)
{
  return nvcompBatchedDeflateDecompressGetTempSize(num_chunks, max_uncompressed_chunk_bytes, temp_bytes);
}

nvcompStatus_t nvcompBatchedGzipDecompressGetTempSizeEx(
    size_t num_chunks, size_t max_uncompressed_chunk_bytes, size_t* temp_bytes, size_t /*max_total_uncompressed_bytes*/)
{
  return nvcompBatchedDeflateDecompressGetTempSize(num_chunks, max_uncompressed_chunk_bytes, temp_bytes);
}

nvcompStatus_t nvcompBatchedGzipDecompressAsync(
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
  return decompress_async<deflate::kGzip>("nvcompBatchedGzipDecompressAsync", device_compressed_ptrs, device_compressed_bytes,
                                          device_uncompressed_bytes, device_actual_uncompressed_bytes, batch_size,
                                          device_uncompressed_ptrs, device_statuses, stream);
}

nvcompStatus_t nvcompBatchedGzipGetDecompressSizeAsync(
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
  hipLaunchKernelGGL(gzip_size_kernel, dim3((unsigned)((batch_size + 255) / 256)), dim3(256), 0, stream,
                     device_compressed_ptrs, device_compressed_bytes, device_uncompressed_bytes, batch_size);
  return launch_status();
}

nvcompStatus_t nvcompBatchedDeflateCompressGetTempSize(
    size_t /*batch_size*/, size_t max_uncompressed_chunk_bytes, nvcompBatchedDeflateOpts_t format_opts, size_t* temp_bytes)
{
  if (temp_bytes == nullptr || !deflate_opts_ok(format_opts)) {
    return nvcompErrorInvalidValue;
  }
  if (max_uncompressed_chunk_bytes > nvcompDeflateCompressionMaxAllowedChunkSize) {
    return nvcompErrorChunkSizeTooLarge;
  }
  *temp_bytes = 0; /* hash table, input image and bit staging live in LDS */
  return nvcompSuccess;
}

nvcompStatus_t nvcompBatchedDeflateCompressGetTempSizeEx(
    size_t batch_size,
    size_t max_uncompressed_chunk_bytes,
    nvcompBatchedDeflateOpts_t format_opts,
    size_t* temp_bytes,
    const size_t /*max_total_uncompressed_bytes*/)
{
  return nvcompBatchedDeflateCompressGetTempSize(batch_size, max_uncompressed_chunk_bytes, format_opts, temp_bytes);
}

nvcompStatus_t nvcompBatchedDeflateCompressGetMaxOutputChunkSize(
    size_t max_uncompressed_chunk_bytes, nvcompBatchedDeflateOpts_t format_opts, size_t* max_compressed_bytes)
{
  if (max_compressed_bytes == nullptr || !deflate_opts_ok(format_opts)) {
    return nvcompErrorInvalidValue;
  }
  if (max_uncompressed_chunk_bytes > nvcompDeflateCompressionMaxAllowedChunkSize) {
    return nvcompErrorChunkSizeTooLarge;
  }
  *max_compressed_bytes = deflate::max_compressed_size(max_uncompressed_chunk_bytes);
  return nvcompSuccess;
}

nvcompStatus_t nvcompBatchedDeflateCompressAsync(
    const void* const* device_uncompressed_ptrs,
    const size_t* device_uncompressed_bytes,
    size_t max_uncompressed_chunk_bytes,
    size_t batch_size,
    void* /*device_temp_ptr*/,
    size_t /*temp_bytes*/,
    void* const* device_compressed_ptrs,
    size_t* device_compressed_bytes,
    nvcompBatchedDeflateOpts_t format_opts,
    hipStream_t stream)
{
  nvlog::call(3, "nvcompBatchedDeflateCompressAsync(batch_size=%zu, max_uncompressed_chunk_bytes=%zu, stream=%p)", batch_size,
              max_uncompressed_chunk_bytes, (void*)stream);
  if (!deflate_opts_ok(format_opts)) {
    return nvcompErrorInvalidValue;
  }
  if (max_uncompressed_chunk_bytes > nvcompDeflateCompressionMaxAllowedChunkSize) {
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
  const dim3 grid((unsigned)((batch_size + kEncWaves - 1) / kEncWaves)), block(64 * kEncWaves);
  if (format_opts.algo == 0) {
    hipLaunchKernelGGL(deflate_compress_kernel<false>, grid, block, 0, stream, device_uncompressed_ptrs, device_uncompressed_bytes,
                       max_uncompressed_chunk_bytes, batch_size, device_compressed_ptrs, device_compressed_bytes);
  } else {
    hipLaunchKernelGGL(deflate_compress_kernel<true>, grid, block, 0, stream, device_uncompressed_ptrs, device_uncompressed_bytes,
                       max_uncompressed_chunk_bytes, batch_size, device_compressed_ptrs, device_compressed_bytes);
  }
  return launch_status();
}

} // extern "C"

#ifdef NVCOMP_LZW_PROF
/* Profiling builds only (scripts/build_deflate_variant.sh dprof -DNVCOMP_LZW_PROF): read (and clear) the per-phase cycle sums of
 * the DEFLATE decoder. Slots 4-9, 11-14: the batch executor's (bench.py names them); 0 = block headers and code tables,
 * 1 = window tables, 2 = enumerations, 3 = a round's decode + records, 10 = symbols decoded one at a time, 15 = the rest. */
extern "C" int nvcompAmdProfReadDeflate(unsigned long long* host_slots, int n)
{
  unsigned long long v[lzw::kProfSlots] = {};
  if (hipMemcpyFromSymbol(v, HIP_SYMBOL(lzw::g_prof), sizeof(v)) != hipSuccess) {
    return -1;
  }
  for (int i = 0; i < n && i < (int)lzw::kProfSlots; ++i) {
    host_slots[i] = v[i];
  }
  unsigned long long z[lzw::kProfSlots] = {};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(lzw::g_prof), z, sizeof(z));
  return (int)lzw::kProfSlots;
}
#endif
