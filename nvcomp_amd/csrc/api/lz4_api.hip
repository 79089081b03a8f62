/*
 * api/lz4_api.hip -- C ABI of the batched LZ4 codec (include/nvcomp/lz4.h) and
 * the kernels it launches. Host side does argument checks and one launch per
 * *Async call on the caller's stream; nothing here allocates or synchronises.
 */
#include <hip/hip_runtime.h>

#include <stdlib.h>

#include "nvcomp/lz4.h"

#include "common/log.h"

#include "nvcomp/amd_ext.h"

#include "common/tuning.h"

#include "lz4/lz4_decode.hip.h"
#include "lz4/lz4_decode_window.hip.h"
#include "lz4/lz4_encode.hip.h"
#include "lz4/lz4_index.hip.h"

namespace {

constexpr unsigned kWavesPerBlock = 4; /* 256-thread workgroups, one chunk per wave */
/* The decoders' workgroup size is a tuning parameter of its own: a workgroup's LDS is released when its LAST wave ends, and
 * chunks of a mixed batch take very different times. */
#ifndef NVCOMP_LZ_DEC_WAVES_PER_BLOCK
#define NVCOMP_LZ_DEC_WAVES_PER_BLOCK 4
#endif
constexpr unsigned kDecWaves = NVCOMP_LZ_DEC_WAVES_PER_BLOCK;
#ifndef NVCOMP_LZM_WAVES_PER_BLOCK
#define NVCOMP_LZM_WAVES_PER_BLOCK 4
#endif
constexpr unsigned kEncWaves = NVCOMP_LZM_WAVES_PER_BLOCK; /* the compressors' workgroup size, same reasoning */
constexpr uint32_t kMaxOutCap = 1u << 26;

/* A/B and ablation kernels exist in measurement builds only (scripts/build_variants.sh passes
 * -DNVCOMP_AMD_LZ4_VARIANT=1 direct | 2 serial | 11 / 12 chase-only / chase+parse ablations, the last two with
 * wrong output by construction); the shipped library has exactly one decoder and no run-time switch. */
#ifndef NVCOMP_AMD_LZ4_VARIANT
#define NVCOMP_AMD_LZ4_VARIANT 0
#endif

/* WAVES = chunks (waves) per workgroup. A workgroup's LDS is released when its last wave ends and the chunks of a batch
 * take very different times: one-wave workgroups keep a full card fuller on the mix (65 536 chunks: 507 -> 518 GB/s;
 * Snappy 350 -> 372) -- but batches of uniformly fast chunks lose 7 % to the four times as many workgroup launches
 * (int32 column, 16 384 chunks: 639 -> 585; Snappy 608 -> 598), and a batch that does not fill the card spreads
 * better in workgroups of four (4 096 chunks: 282 against 263). Snappy takes the trade from 8 192 chunks on
 * (api/snappy_api.hip), LZ4 does not: kSingleWaveFromBatch is out of reach. */
constexpr size_t kSingleWaveFromBatch = ~(size_t)0;

template <bool CHECKED, int ABLATE = 0, unsigned WAVES = kDecWaves>
__global__ void __launch_bounds__(64 * WAVES, NVCOMP_LZW_WAVES_PER_SIMD) lz4_decompress_window_kernel(
    const void* const* __restrict__ comp_ptrs,
    const size_t* __restrict__ comp_bytes,
    const size_t* out_caps,
    size_t* actual_bytes,
    size_t batch_size,
    void* const* __restrict__ out_ptrs,
    nvcompStatus_t* statuses,
    const uint32_t* __restrict__ index_counts /* non-null: only the chunks the indexer left out (lzi::kNotIndexed) */)
{
  __shared__ __attribute__((aligned(16))) uint8_t lds[WAVES][lzg::kLdsPerWave];
  const uint32_t w = wave::uniform(threadIdx.x >> 6);
  const size_t chunk = (size_t)blockIdx.x * WAVES + w;
  if (chunk >= batch_size) {
    return;
  }
  if (index_counts != nullptr && wave::uniform(index_counts[chunk]) != lzi::kNotIndexed) {
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
  if (in_len64 > 0xffffffffull - 64) {
    err = lz::kErrInput;
  } else {
#ifdef NVCOMP_LZW_PROF
    lzw::prof_begin();
#endif
    produced = lz4w::decode_chunk<CHECKED, ABLATE>(in, (uint32_t)in_len64, out, (uint32_t)cap64, lds[w], err);
#ifdef NVCOMP_LZW_PROF
    lzw::prof_end();
#endif
  }
  if (wave::lane_id() == 0) {
    if (actual_bytes != nullptr) {
      actual_bytes[chunk] = err ? 0 : produced;
    }
    if (CHECKED) {
      statuses[chunk] = err ? nvcompErrorCannotDecompress : nvcompSuccess;
    }
  }
}

#if !NVCOMP_LZ_GATHER
/* Small batches: two waves per chunk, a producer (chase + parse) and a consumer (execute), lz4w::pair. */
template <bool CHECKED>
__global__ void __launch_bounds__(128, 4) lz4_decompress_pair_kernel(
    const void* const* __restrict__ comp_ptrs,
    const size_t* __restrict__ comp_bytes,
    const size_t* out_caps,
    size_t* actual_bytes,
    size_t batch_size,
    void* const* __restrict__ out_ptrs,
    nvcompStatus_t* statuses)
{
  __shared__ __attribute__((aligned(16))) uint8_t lds[lz4w::pair::kLdsPerChunk];
  const uint32_t w = wave::uniform(threadIdx.x >> 6);
  const size_t chunk = blockIdx.x;
  if (chunk >= batch_size) {
    return;
  }
  if (threadIdx.x < 4) {
    ((uint32_t*)(lds + lz4w::pair::kLdsPerChunk - lz4w::pair::kCtrlBytes))[threadIdx.x] = 0; /* both slots empty, no abort */
  }
  __syncthreads();
  const uint8_t* in = wave::uniform_ptr((const uint8_t*)comp_ptrs[chunk]);
  uint8_t* out = wave::uniform_ptr((uint8_t*)out_ptrs[chunk]);
  const size_t in_len64 = wave::uniform64(comp_bytes[chunk]);
  size_t cap64 = wave::uniform64(out_caps[chunk]);
  if (cap64 > kMaxOutCap) {
    cap64 = kMaxOutCap;
  }
  const bool too_long = in_len64 > 0xffffffffull - 64;
  const bool work = !too_long && in_len64 != 0;
  if (w == 0) {
    if (work) {
      lz4w::pair::produce(in, (uint32_t)in_len64, lds);
    }
    return;
  }
  uint32_t err = too_long ? lz::kErrInput : lz::kErrNone;
  uint32_t produced = 0;
  if (work) {
    produced = lz4w::pair::consume<CHECKED>(in, (uint32_t)in_len64, out, (uint32_t)cap64, lds, err);
  }
  if (wave::lane_id() == 0) {
    if (actual_bytes != nullptr) {
      actual_bytes[chunk] = err ? 0 : produced;
    }
    if (CHECKED) {
      statuses[chunk] = err ? nvcompErrorCannotDecompress : nvcompSuccess;
    }
  }
}
#endif

/* Token indexer: one LANE per chunk, 64 chunks per single-wave workgroup (common/lz_index.hip.h). */
__global__ void __launch_bounds__(64) lz4_index_kernel(
    const void* const* __restrict__ comp_ptrs, const size_t* __restrict__ comp_bytes, size_t batch_size, lzi::Layout lay)
{
  __shared__ __attribute__((aligned(16))) uint32_t lds[lzi::kLdsDwords];
  lzi::index_chunks<lz4i::Format>(comp_ptrs, comp_bytes, batch_size, lay, lds);
}

/* The window decoder fed from the token index: no chase tables in LDS, one wave per chunk. */
template <bool CHECKED>
__global__ void __launch_bounds__(64 * kDecWaves, NVCOMP_LZW_INDEXED_WAVES_PER_SIMD) lz4_decompress_indexed_kernel(
    const void* const* __restrict__ comp_ptrs,
    const size_t* __restrict__ comp_bytes,
    const size_t* out_caps,
    size_t* actual_bytes,
    size_t batch_size,
    void* const* __restrict__ out_ptrs,
    nvcompStatus_t* statuses,
    lzi::Layout lay)
{
  __shared__ __attribute__((aligned(16))) uint8_t lds[kDecWaves][lzg::kLdsPerWaveIndexed];
  const uint32_t w = wave::uniform(threadIdx.x >> 6);
  const size_t chunk = (size_t)blockIdx.x * kDecWaves + w;
  if (chunk >= batch_size) {
    return;
  }
  const uint32_t n_tok = wave::uniform(lay.counts[chunk]);
  if (n_tok == lzi::kNotIndexed) {
    return; /* the chase decoder takes this chunk */
  }
  const uint8_t* in = wave::uniform_ptr((const uint8_t*)comp_ptrs[chunk]);
  uint8_t* out = wave::uniform_ptr((uint8_t*)out_ptrs[chunk]);
  const uint32_t in_len = (uint32_t)wave::uniform64(comp_bytes[chunk]); /* <= lzi::kMaxInput */
  size_t cap64 = wave::uniform64(out_caps[chunk]);
  if (cap64 > kMaxOutCap) {
    cap64 = kMaxOutCap;
  }
  uint32_t err = lz::kErrNone;
  const uint32_t produced = lz4w::decode_chunk_indexed<CHECKED>(
      in, in_len, out, (uint32_t)cap64, lds[w], lay.table + chunk * (size_t)lay.stride, n_tok, err);
  if (wave::lane_id() == 0) {
    if (actual_bytes != nullptr) {
      actual_bytes[chunk] = err ? 0 : produced;
    }
    if (CHECKED) {
      statuses[chunk] = err ? nvcompErrorCannotDecompress : nvcompSuccess;
    }
  }
}

template <bool CHECKED, bool LANE_PARALLEL>
__global__ void __launch_bounds__(64 * kDecWaves) lz4_decompress_kernel(
    const void* const* __restrict__ comp_ptrs,
    const size_t* __restrict__ comp_bytes,
    const size_t* out_caps,
    size_t* actual_bytes,
    size_t batch_size,
    void* const* __restrict__ out_ptrs,
    nvcompStatus_t* statuses)
{
  const size_t chunk = (size_t)blockIdx.x * kDecWaves + wave::uniform(threadIdx.x >> 6);
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
  if (in_len64 > 0xffffffffull - 8) {
    err = lz::kErrInput;
  } else {
    produced = lz4::decode_chunk<CHECKED, LANE_PARALLEL, false>(in, (uint32_t)in_len64, out, (uint32_t)cap64, err);
  }
  if (wave::lane_id() == 0) {
    if (actual_bytes != nullptr) {
      actual_bytes[chunk] = err ? 0 : produced;
    }
    if (CHECKED) {
      statuses[chunk] = err ? nvcompErrorCannotDecompress : nvcompSuccess;
    }
  }
}

__global__ void __launch_bounds__(64 * kWavesPerBlock) lz4_decompress_size_kernel(
    const void* const* __restrict__ comp_ptrs,
    const size_t* __restrict__ comp_bytes,
    size_t* uncompressed_bytes,
    size_t batch_size)
{
  const size_t chunk = (size_t)blockIdx.x * kWavesPerBlock + wave::uniform(threadIdx.x >> 6);
  if (chunk >= batch_size) {
    return;
  }
  const uint8_t* in = wave::uniform_ptr((const uint8_t*)comp_ptrs[chunk]);
  const size_t in_len64 = wave::uniform64(comp_bytes[chunk]);
  uint32_t err = lz::kErrNone;
  uint32_t produced = 0;
  if (in_len64 <= 0xffffffffull - 8) {
    produced = lz4::decode_chunk<false, true, true>(in, (uint32_t)in_len64, nullptr, 0, err);
  }
  if (wave::lane_id() == 0) {
    uncompressed_bytes[chunk] = err ? 0 : produced;
  }
}

/* STRIDE: the element size the caller declared (nvcompBatchedLZ4Opts_t.data_type): matches are searched at element
 * boundaries only, a step covers 64 elements (common/lz_match.hip.h). */
template <uint32_t STRIDE>
__global__ void __launch_bounds__(64 * kEncWaves, NVCOMP_LZM_WAVES_PER_SIMD) lz4_compress_kernel(
    const void* const* __restrict__ in_ptrs,
    const size_t* __restrict__ in_bytes,
    size_t max_chunk_bytes,
    size_t batch_size,
    void* const* __restrict__ out_ptrs,
    size_t* out_bytes)
{
  __shared__ uint16_t tables[kEncWaves][lzm::kTableU16];
  __shared__ __attribute__((aligned(8))) uint8_t images[kEncWaves][lzm::kStageBytes];
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
  const uint32_t produced = n64 > max_chunk_bytes ? 0u : lz4::encode_chunk<STRIDE>(src, (uint32_t)n64, dst, tables[w], images[w]);
  if (wave::lane_id() == 0) {
    out_bytes[chunk] = produced;
  }
}

/* hipGetLastError() is sticky per host thread: an unrelated earlier runtime call
 * of the application (e.g. a failed pointer-attribute query) must not be
 * reported as this launch's failure, so the slate is cleared before launching. */
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

/* worst case of the block format: one length byte per 255 literals + token + slack (== LZ4_compressBound) */
size_t lz4_bound(size_t n)
{
  return n + n / 255 + 16;
}

bool lz4_type_ok(nvcompType_t t)
{
  return (t >= NVCOMP_TYPE_CHAR && t <= NVCOMP_TYPE_UINT) || t == NVCOMP_TYPE_BITS;
}

} // namespace

extern "C" {

nvcompStatus_t nvcompBatchedLZ4DecompressGetTempSize(
    size_t num_chunks, size_t max_uncompressed_chunk_bytes, size_t* temp_bytes)
{
  if (temp_bytes == nullptr) {
    return nvcompErrorInvalidValue;
  }
  /* only the opt-in two-kernel path (include/nvcomp/amd_ext.h) wants scratch: the token index, u32 count + one u16 per
   * possible token of every chunk (common/lz_index.hip.h) */
  *temp_bytes = num_chunks == 0 || num_chunks < nvcomp_amd_tuning::lz_index_min_batch
                    ? 0
                    : lzi::temp_bytes_for(num_chunks, lz4_bound(max_uncompressed_chunk_bytes));
  return nvcompSuccess;
}

nvcompStatus_t nvcompBatchedLZ4DecompressGetTempSizeEx(
    size_t num_chunks, size_t max_uncompressed_chunk_bytes, size_t* temp_bytes, size_t /*max_total_uncompressed_bytes*/)
{
  return nvcompBatchedLZ4DecompressGetTempSize(num_chunks, max_uncompressed_chunk_bytes, temp_bytes);
}

nvcompStatus_t nvcompBatchedLZ4DecompressAsync(
    const void* const* device_compressed_ptrs,
    const size_t* device_compressed_bytes,
    const size_t* device_uncompressed_bytes,
    size_t* device_actual_uncompressed_bytes,
    size_t batch_size,
    void* const device_temp_ptr,
    size_t temp_bytes,
    void* const* device_uncompressed_ptrs,
    nvcompStatus_t* device_statuses,
    hipStream_t stream)
{
  nvlog::call(3, "nvcompBatchedLZ4DecompressAsync(batch_size=%zu, statuses=%s, actual_sizes=%s, temp_bytes=%zu, stream=%p)",
              batch_size, device_statuses ? "yes" : "null", device_actual_uncompressed_bytes ? "yes" : "null", temp_bytes,
              (void*)stream);
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
  const bool checked = device_statuses != nullptr;
#define NVCOMP_LZ4_ARGS                                                                                        \
  device_compressed_ptrs, device_compressed_bytes, device_uncompressed_bytes, device_actual_uncompressed_bytes, \
      batch_size, device_uncompressed_ptrs, device_statuses
#if NVCOMP_AMD_LZ4_VARIANT == 0 && !NVCOMP_LZ_GATHER
  /* Small batches cannot fill the card with one wave per chunk: two waves per chunk (producer / consumer). */
  if (batch_size <= nvcomp_amd_tuning::lz_pair_max_batch) {
    const dim3 pgrid((unsigned)batch_size), pblock(128);
    if (checked) {
      hipLaunchKernelGGL((lz4_decompress_pair_kernel<true>), pgrid, pblock, 0, stream, NVCOMP_LZ4_ARGS);
    } else {
      hipLaunchKernelGGL((lz4_decompress_pair_kernel<false>), pgrid, pblock, 0, stream, NVCOMP_LZ4_ARGS);
    }
    return launch_status();
  }
#endif
#if NVCOMP_AMD_LZ4_VARIANT == 0
  /* Two-kernel path when the caller's temp buffer holds the token index and the batch is large enough to fill
   * the indexer's lanes (one lane per chunk; below the threshold the chase decoder's latency is lower). */
  const lzi::Layout lay = batch_size >= nvcomp_amd_tuning::lz_index_min_batch
                              ? lzi::carve(device_temp_ptr, temp_bytes, batch_size)
                              : lzi::Layout{nullptr, nullptr, 0};
  const uint32_t* only = nullptr;
  if (lay.stride != 0) {
    hipLaunchKernelGGL(lz4_index_kernel, dim3((unsigned)((batch_size + 63) / 64)), dim3(64), 0, stream,
                       device_compressed_ptrs, device_compressed_bytes, batch_size, lay);
    if (checked) {
      hipLaunchKernelGGL((lz4_decompress_indexed_kernel<true>), grid, block, 0, stream, NVCOMP_LZ4_ARGS, lay);
    } else {
      hipLaunchKernelGGL((lz4_decompress_indexed_kernel<false>), grid, block, 0, stream, NVCOMP_LZ4_ARGS, lay);
    }
    only = lay.counts; /* chunks the index leaves out (> 65535 bytes, row overflow) fall to the chase decoder */
  }
  if (batch_size >= kSingleWaveFromBatch) {
    const dim3 grid1((unsigned)batch_size), block1(64);
    if (checked) {
      hipLaunchKernelGGL((lz4_decompress_window_kernel<true, 0, 1>), grid1, block1, 0, stream, NVCOMP_LZ4_ARGS, only);
    } else {
      hipLaunchKernelGGL((lz4_decompress_window_kernel<false, 0, 1>), grid1, block1, 0, stream, NVCOMP_LZ4_ARGS, only);
    }
  } else if (checked) {
    hipLaunchKernelGGL((lz4_decompress_window_kernel<true>), grid, block, 0, stream, NVCOMP_LZ4_ARGS, only);
  } else {
    hipLaunchKernelGGL((lz4_decompress_window_kernel<false>), grid, block, 0, stream, NVCOMP_LZ4_ARGS, only);
  }
#elif NVCOMP_AMD_LZ4_VARIANT == 11
  hipLaunchKernelGGL((lz4_decompress_window_kernel<false, 1>), grid, block, 0, stream, NVCOMP_LZ4_ARGS, nullptr);
#elif NVCOMP_AMD_LZ4_VARIANT == 12
  hipLaunchKernelGGL((lz4_decompress_window_kernel<false, 2>), grid, block, 0, stream, NVCOMP_LZ4_ARGS, nullptr);
#elif NVCOMP_AMD_LZ4_VARIANT == 3 /* the round-1 decoder: chase inside the decode kernel, whatever the batch size */
  if (checked) {
    hipLaunchKernelGGL((lz4_decompress_window_kernel<true>), grid, block, 0, stream, NVCOMP_LZ4_ARGS, nullptr);
  } else {
    hipLaunchKernelGGL((lz4_decompress_window_kernel<false>), grid, block, 0, stream, NVCOMP_LZ4_ARGS, nullptr);
  }
#else
  if (checked) {
    hipLaunchKernelGGL((lz4_decompress_kernel<true, NVCOMP_AMD_LZ4_VARIANT == 1>), grid, block, 0, stream, NVCOMP_LZ4_ARGS);
  } else {
    hipLaunchKernelGGL((lz4_decompress_kernel<false, NVCOMP_AMD_LZ4_VARIANT == 1>), grid, block, 0, stream, NVCOMP_LZ4_ARGS);
  }
#endif
#undef NVCOMP_LZ4_ARGS
  (void)temp_bytes;
  return launch_status();
}

nvcompStatus_t nvcompBatchedLZ4GetDecompressSizeAsync(
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
  hipLaunchKernelGGL(lz4_decompress_size_kernel, dim3(grid_for(batch_size)), dim3(64 * kWavesPerBlock), 0, stream,
                     device_compressed_ptrs, device_compressed_bytes, device_uncompressed_bytes, batch_size);
  return launch_status();
}

nvcompStatus_t nvcompBatchedLZ4CompressGetTempSize(
    size_t /*batch_size*/, size_t max_uncompressed_chunk_bytes, nvcompBatchedLZ4Opts_t format_opts, size_t* temp_bytes)
{
  if (temp_bytes == nullptr || !lz4_type_ok(format_opts.data_type)) {
    return nvcompErrorInvalidValue;
  }
  if (max_uncompressed_chunk_bytes > nvcompLZ4CompressionMaxAllowedChunkSize) {
    return nvcompErrorChunkSizeTooLarge;
  }
  *temp_bytes = 0; /* the per-chunk hash tables live in LDS */
  return nvcompSuccess;
}

nvcompStatus_t nvcompBatchedLZ4CompressGetTempSizeEx(
    size_t batch_size,
    size_t max_uncompressed_chunk_bytes,
    nvcompBatchedLZ4Opts_t format_opts,
    size_t* temp_bytes,
    const size_t /*max_total_uncompressed_bytes*/)
{
  return nvcompBatchedLZ4CompressGetTempSize(batch_size, max_uncompressed_chunk_bytes, format_opts, temp_bytes);
}

nvcompStatus_t nvcompBatchedLZ4CompressGetMaxOutputChunkSize(
    size_t max_uncompressed_chunk_bytes, nvcompBatchedLZ4Opts_t format_opts, size_t* max_compressed_bytes)
{
  if (max_compressed_bytes == nullptr || !lz4_type_ok(format_opts.data_type)) {
    return nvcompErrorInvalidValue;
  }
  if (max_uncompressed_chunk_bytes > nvcompLZ4CompressionMaxAllowedChunkSize) {
    return nvcompErrorChunkSizeTooLarge;
  }
  *max_compressed_bytes = lz4_bound(max_uncompressed_chunk_bytes);
  return nvcompSuccess;
}

nvcompStatus_t nvcompBatchedLZ4CompressAsync(
    const void* const* device_uncompressed_ptrs,
    const size_t* device_uncompressed_bytes,
    size_t max_uncompressed_chunk_bytes,
    size_t batch_size,
    void* /*device_temp_ptr*/,
    size_t /*temp_bytes*/,
    void* const* device_compressed_ptrs,
    size_t* device_compressed_bytes,
    nvcompBatchedLZ4Opts_t format_opts,
    hipStream_t stream)
{
  nvlog::call(3, "nvcompBatchedLZ4CompressAsync(batch_size=%zu, max_uncompressed_chunk_bytes=%zu, stream=%p)", batch_size,
              max_uncompressed_chunk_bytes, (void*)stream);
  if (!lz4_type_ok(format_opts.data_type)) {
    return nvcompErrorInvalidValue;
  }
  if (max_uncompressed_chunk_bytes > nvcompLZ4CompressionMaxAllowedChunkSize) {
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
#define NVCOMP_LZ4_COMPRESS(STRIDE)                                                                                  \
  hipLaunchKernelGGL((lz4_compress_kernel<STRIDE>), dim3((unsigned)((batch_size + kEncWaves - 1) / kEncWaves)), dim3(64 * kEncWaves), 0, stream,   \
                     device_uncompressed_ptrs, device_uncompressed_bytes, max_uncompressed_chunk_bytes, batch_size,     \
                     device_compressed_ptrs, device_compressed_bytes)
  switch (format_opts.data_type) {
  case NVCOMP_TYPE_SHORT:
  case NVCOMP_TYPE_USHORT: NVCOMP_LZ4_COMPRESS(2); break;
  case NVCOMP_TYPE_INT:
  case NVCOMP_TYPE_UINT: NVCOMP_LZ4_COMPRESS(4); break;
  default: NVCOMP_LZ4_COMPRESS(1); break;
  }
#undef NVCOMP_LZ4_COMPRESS
  return launch_status();
}

} // extern "C"

#ifdef NVCOMP_LZM_PROF
/* Profiling builds only: read (and clear) the per-phase cycle sums of the LZ4 compressor. */
extern "C" int nvcompAmdCompProfRead(unsigned long long* host_slots, int n)
{
  unsigned long long v[lzm::kProfSlots] = {};
  if (hipMemcpyFromSymbol(v, HIP_SYMBOL(lzm::g_prof), sizeof(v)) != hipSuccess) {
    return -1;
  }
  for (int i = 0; i < n && i < (int)lzm::kProfSlots; ++i) {
    host_slots[i] = v[i];
  }
  unsigned long long z[lzm::kProfSlots] = {};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(lzm::g_prof), z, sizeof(z));
  return (int)lzm::kProfSlots;
}
#endif

#ifdef NVCOMP_LZW_PROF
/* Profiling builds only: read (and clear) the per-phase cycle sums of the window decoder. */
extern "C" int nvcompAmdProfRead(unsigned long long* host_slots, int n)
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
