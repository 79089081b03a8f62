/*
 * api/snappy_api.hip -- C ABI of the batched Snappy codec (include/nvcomp/snappy.h) and
 * the kernels it launches. Host side does argument checks and one launch per
 * *Async call on the caller's stream; nothing here allocates or synchronises.
 */
#include <hip/hip_runtime.h>

#include <stdlib.h>

#include "nvcomp/snappy.h"

#include "nvcomp/amd_ext.h"

#include "common/log.h"

#include "common/lz_launch.hip.h"
#include "snappy/snappy_decode.hip.h"
#include "snappy/snappy_decode_window.hip.h"
#include "common/lz_team.hip.h"
#include "snappy/snappy_encode.hip.h"

namespace {

constexpr unsigned kWavesPerBlock = 4; /* 256-thread workgroups, one chunk per wave */
#ifndef NVCOMP_LZ_DEC_WAVES_PER_BLOCK
#define NVCOMP_LZ_DEC_WAVES_PER_BLOCK 4
#endif
constexpr unsigned kDecWaves = NVCOMP_LZ_DEC_WAVES_PER_BLOCK;
#ifndef NVCOMP_LZM_WAVES_PER_BLOCK
#define NVCOMP_LZM_WAVES_PER_BLOCK 4
#endif
constexpr unsigned kEncWaves = NVCOMP_LZM_WAVES_PER_BLOCK; /* the compressors' workgroup size */
/* Untyped data takes the 256-position steps of common/lz_match_wide.hip.h (0: the one-window compressor, A/B build). */
#ifndef NVCOMP_LZM_WIDE
#define NVCOMP_LZM_WIDE 1
#endif
#ifndef NVCOMP_LZMW_WAVES_PER_BLOCK
#define NVCOMP_LZMW_WAVES_PER_BLOCK 1
#endif
constexpr unsigned kWideWaves = NVCOMP_LZMW_WAVES_PER_BLOCK;
#ifndef NVCOMP_LZMW_WAVES_PER_SIMD
#define NVCOMP_LZMW_WAVES_PER_SIMD 4 /* what the wave's LDS allows (15-16 waves per CU): a budget of 128 registers */
#endif
using lzl::kMaxOutCap;
#ifndef NVCOMP_SNAPPY_RUNS
#define NVCOMP_SNAPPY_RUNS 1 /* A/B: 0 = no run executor in the Snappy decoder */
#endif
#ifndef NVCOMP_SNAPPY_RUNS_RATIO
#define NVCOMP_SNAPPY_RUNS_RATIO 8
#endif
constexpr size_t kRunsRatio = NVCOMP_SNAPPY_RUNS_RATIO;

/* Decode chunk `chunk` of the batch with the calling wave and report its size and status. */
template <bool CHECKED, class BatchPtr>
__device__ __forceinline__ void decode_one(BatchPtr b, size_t chunk, uint8_t* lds)
{
  const uint8_t* in = wave::uniform_ptr((const uint8_t*)b->comp_ptrs[chunk]);
  uint8_t* out = wave::uniform_ptr((uint8_t*)b->out_ptrs[chunk]);
  const size_t in_len64 = wave::uniform64(b->comp_bytes[chunk]);
  size_t cap64 = wave::uniform64(b->out_caps[chunk]);
  if (cap64 > kMaxOutCap) {
    cap64 = kMaxOutCap;
  }
  uint32_t err = lz::kErrNone;
  uint32_t produced = 0;
  if (in_len64 > 0xffffffffull - 64) {
    err = lz::kErrInput;
  } else {
    /* a chunk that shrank 8 x or more takes the instance of the loop that tries the run executor (snappyw::decode_chunk) */
    if (NVCOMP_LZW_RUNS && NVCOMP_SNAPPY_RUNS && in_len64 * kRunsRatio <= cap64) {
      produced = snappyw::decode_chunk<CHECKED, true>(in, (uint32_t)in_len64, out, (uint32_t)cap64, lds, err);
    } else {
      produced = snappyw::decode_chunk<CHECKED, false>(in, (uint32_t)in_len64, out, (uint32_t)cap64, lds, err);
    }
  }
  if (wave::lane_id() == 0) {
    size_t* actual_bytes = b->actual_bytes;
    if (actual_bytes != nullptr) {
      actual_bytes[chunk] = err ? 0 : produced;
    }
    if (CHECKED && b->statuses != nullptr) {
      b->statuses[chunk] = err ? nvcompErrorCannotDecompress : nvcompSuccess;
    }
  }
}

/* One wave per chunk at a time; with a ticket counter the waves are persistent (common/lz_launch.hip.h). */
template <bool CHECKED>
__global__ void __launch_bounds__(64 * kDecWaves, NVCOMP_LZW_WAVES_PER_SIMD) snappy_decompress_window_kernel(const lzl::Launch launch)
{
  __shared__ __attribute__((aligned(16))) uint8_t lds[kDecWaves][lzw::kLdsPerWave];
  const uint32_t w = wave::uniform(threadIdx.x >> 6);
  size_t place = (size_t)blockIdx.x * kDecWaves + w; /* the wave's place in the launch = its first chunk */
#ifdef NVCOMP_LZW_PROF
  lzw::prof_begin();
#endif
  for (;;) {
    /* the arguments are read where they are used, not held in scalar registers across the decode (wave::kernel_args) */
    const auto* a = wave::kernel_args(launch);
    if (place >= a->b.batch_size) {
      break;
    }
    const size_t chunk = place;
    decode_one<CHECKED>(&a->b, chunk, lds[w]);
    a = wave::kernel_args(launch);
    uint32_t* ticket = a->ticket;
    if (ticket == nullptr) {
      break;
    }
    place = lzl::next_chunk(ticket, a->first_dynamic);
  }
#ifdef NVCOMP_LZW_PROF
  lzw::prof_end();
#endif
}

/* A workgroup per chunk (common/lz_team.hip.h): batches that cannot fill the card with one wave per chunk. Persistent
 * workgroups when the caller's temp buffer holds a ticket counter, one workgroup per chunk otherwise. */
template <bool CHECKED, uint32_t WAVES>
__global__ void __launch_bounds__(64 * WAVES, 4) snappy_decompress_team_kernel(const lzl::Launch launch)
{
  __shared__ __attribute__((aligned(16))) uint8_t lds[lzt::Geo<WAVES>::kLds];
  size_t chunk = blockIdx.x;
  for (;;) {
    const auto* a = wave::kernel_args(launch);
    if (chunk >= a->b.batch_size) {
      break;
    }
    const uint8_t* in = wave::uniform_ptr((const uint8_t*)a->b.comp_ptrs[chunk]);
    uint8_t* out = wave::uniform_ptr((uint8_t*)a->b.out_ptrs[chunk]);
    const size_t in_len64 = wave::uniform64(a->b.comp_bytes[chunk]);
    size_t cap64 = wave::uniform64(a->b.out_caps[chunk]);
    if (cap64 > kMaxOutCap) {
      cap64 = kMaxOutCap;
    }
    uint32_t err = lz::kErrNone;
    uint32_t produced = 0;
    if (in_len64 > 0xffffffffull - 64) {
      err = lz::kErrInput;
    } else {
      produced = lzt::decode_chunk<snappyw::TeamFrontEnd, WAVES>(
          in, (uint32_t)in_len64, out, (uint32_t)cap64, lds, err,
          [](uint32_t role, const uint8_t* i, uint32_t n, uint8_t* o, uint32_t cap, uint8_t* scratch, uint32_t& e) -> uint32_t {
            const bool solo = NVCOMP_LZW_RUNS && NVCOMP_SNAPPY_RUNS && NVCOMP_LZ_PAIR_SOLO && (size_t)n * kRunsRatio <= cap;
            if (role == 0) {
              if (!solo) {
                snappyw::pair::produce<true>(i, n, scratch);
              }
              return 0u;
            }
            if (solo) {
              return snappyw::decode_chunk<true, true>(i, n, o, cap, scratch, e); /* api/lz4_api.hip */
            }
            return snappyw::pair::consume<true>(i, n, o, cap, scratch, e);
          });
    }
    a = wave::kernel_args(launch);
    if (threadIdx.x == 0) {
      size_t* actual_bytes = a->b.actual_bytes;
      if (actual_bytes != nullptr) {
        actual_bytes[chunk] = err ? 0 : produced;
      }
      if (CHECKED && a->b.statuses != nullptr) {
        a->b.statuses[chunk] = err ? nvcompErrorCannotDecompress : nvcompSuccess;
      }
    }
    uint32_t* ticket = a->ticket;
    if (ticket == nullptr) {
      break;
    }
    uint32_t* slot = (uint32_t*)(lds + lzt::Geo<WAVES>::kLds - 4 * lzt::kCtlWords) + lzt::kCtlTicket;
    if (threadIdx.x == 0) {
      *slot = atomicAdd(ticket, 1u);
    }
    __syncthreads();
    chunk = a->first_dynamic + wave::uniform(*slot);
    __syncthreads();
  }
}

/* Small batches: two waves per chunk, a producer (chase + parse) and a consumer (execute), snappyw::pair. */
template <bool CHECKED>
__global__ void __launch_bounds__(128, 7) snappy_decompress_pair_kernel(const lzl::Batch b)
{
  __shared__ __attribute__((aligned(16))) uint8_t lds[lzw::pair::kLdsPerChunk];
  const uint32_t w = wave::uniform(threadIdx.x >> 6);
  const size_t chunk = blockIdx.x;
  if (chunk >= b.batch_size) {
    return;
  }
  if (threadIdx.x < 4) {
    ((uint32_t*)(lds + lzw::pair::kLdsPerChunk - lzw::pair::kCtrlBytes))[threadIdx.x] = 0; /* both slots empty, no abort */
  }
  __syncthreads();
  const uint8_t* in = wave::uniform_ptr((const uint8_t*)b.comp_ptrs[chunk]);
  uint8_t* out = wave::uniform_ptr((uint8_t*)b.out_ptrs[chunk]);
  const size_t in_len64 = wave::uniform64(b.comp_bytes[chunk]);
  size_t cap64 = wave::uniform64(b.out_caps[chunk]);
  if (cap64 > kMaxOutCap) {
    cap64 = kMaxOutCap;
  }
  const bool work = in_len64 != 0 && in_len64 <= 0xffffffffull - 64; /* an empty stream has no preamble: malformed */
  /* a chunk that shrank 8 x or more: the second wave alone, with the one-wave loop that holds the run executor (api/lz4_api.hip) */
  static_assert(lzw::kLdsPerWave <= lzw::pair::kLdsPerChunk, "the lone wave's LDS is the pair's");
  const bool solo = NVCOMP_LZW_RUNS && NVCOMP_SNAPPY_RUNS && NVCOMP_LZ_PAIR_SOLO && work && in_len64 * kRunsRatio <= cap64;
  if (w == 0) {
    if (work && !solo) {
      snappyw::pair::produce<CHECKED>(in, (uint32_t)in_len64, lds);
    }
    return;
  }
  uint32_t err = work ? lz::kErrNone : lz::kErrInput;
  uint32_t produced = 0;
  if (solo) {
    produced = snappyw::decode_chunk<CHECKED, true>(in, (uint32_t)in_len64, out, (uint32_t)cap64, lds, err);
  } else if (work) {
    produced = snappyw::pair::consume<CHECKED>(in, (uint32_t)in_len64, out, (uint32_t)cap64, lds, err);
  }
  if (wave::lane_id() == 0) {
    if (b.actual_bytes != nullptr) {
      b.actual_bytes[chunk] = err ? 0 : produced;
    }
    if (CHECKED && b.statuses != nullptr) {
      b.statuses[chunk] = err ? nvcompErrorCannotDecompress : nvcompSuccess;
    }
  }
}

__global__ void __launch_bounds__(64 * kWavesPerBlock) snappy_decompress_size_kernel(
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
  uint32_t produced = 0;
  if (in_len64 <= 0xffffffffull - 8) {
    bool ok;
    produced = snappy::decoded_size(in, (uint32_t)in_len64, ok); /* preamble only */
  }
  if (wave::lane_id() == 0) {
    uncompressed_bytes[chunk] = produced;
  }
}

__global__ void __launch_bounds__(64 * kEncWaves, NVCOMP_LZM_WAVES_PER_SIMD) snappy_compress_kernel(const lzl::CompressLaunch launch)
{
  __shared__ uint16_t tables[kEncWaves][lzm::kTableU16];
  __shared__ __attribute__((aligned(8))) uint8_t images[kEncWaves][lzm::kImageBytes];
  const uint32_t w = wave::uniform(threadIdx.x >> 6);
  size_t chunk = (size_t)blockIdx.x * kEncWaves + w;
  /* persistent waves, as in the decoders (common/lz_launch.hip.h): chunks of a batch compress at very different speeds */
  for (;;) {
    const auto* a = wave::kernel_args(launch);
    if (chunk >= a->batch_size) {
      break;
    }
    const uint8_t* src = wave::uniform_ptr((const uint8_t*)a->in_ptrs[chunk]);
    uint8_t* dst = wave::uniform_ptr((uint8_t*)a->out_ptrs[chunk]);
    const size_t n64 = wave::uniform64(a->in_bytes[chunk]);
    /* a chunk larger than the caller declared would overrun the output slot sized from GetMaxOutputChunkSize: it is
     * not compressed, its size reads 0 */
    const uint32_t produced = n64 > a->max_chunk_bytes ? 0u : snappy::encode_chunk(src, (uint32_t)n64, dst, tables[w], images[w]);
    a = wave::kernel_args(launch);
    if (wave::lane_id() == 0) {
      a->out_bytes[chunk] = produced;
    }
    uint32_t* ticket = a->ticket;
    if (ticket == nullptr) {
      break;
    }
    chunk = lzl::next_chunk(ticket, a->first_dynamic);
  }
}

/* Untyped data: 256-position steps (common/lz_match_wide.hip.h); a wave's LDS is lzm::wide::kLdsPerWave bytes. */
__global__ void __launch_bounds__(64 * kWideWaves, NVCOMP_LZMW_WAVES_PER_SIMD) snappy_compress_wide_kernel(const lzl::CompressLaunch launch)
{
  __shared__ uint16_t tables[kWideWaves][lzm::wide::kEntries];
  __shared__ __attribute__((aligned(16))) uint8_t images[kWideWaves][lzm::wide::kImage];
  __shared__ __attribute__((aligned(16))) uint8_t scratch[kWideWaves][lzm::wide::kScratch];
  const uint32_t w = wave::uniform(threadIdx.x >> 6);
  size_t chunk = (size_t)blockIdx.x * kWideWaves + w;
  for (;;) {
    const auto* a = wave::kernel_args(launch);
    if (chunk >= a->batch_size) {
      break;
    }
    const uint8_t* src = wave::uniform_ptr((const uint8_t*)a->in_ptrs[chunk]);
    uint8_t* dst = wave::uniform_ptr((uint8_t*)a->out_ptrs[chunk]);
    const size_t n64 = wave::uniform64(a->in_bytes[chunk]);
    const uint32_t produced = n64 > a->max_chunk_bytes ? 0u : snappy::encode_chunk_wide(src, (uint32_t)n64, dst, tables[w], images[w], scratch[w]);
    a = wave::kernel_args(launch);
    if (wave::lane_id() == 0) {
      a->out_bytes[chunk] = produced;
    }
    uint32_t* ticket = a->ticket;
    if (ticket == nullptr) {
      break;
    }
    chunk = lzl::next_chunk(ticket, a->first_dynamic);
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

bool snappy_opts_ok(nvcompBatchedSnappyOpts_t opts)
{
  return opts.reserved == 0;
}

} // namespace

extern "C" {

nvcompStatus_t nvcompBatchedSnappyDecompressGetTempSize(
    size_t num_chunks, size_t /*max_uncompressed_chunk_bytes*/, size_t* temp_bytes)
{
  if (temp_bytes == nullptr) {
    return nvcompErrorInvalidValue;
  }
  /* the ticket counter of the persistent waves / workgroups (common/lz_launch.hip.h); the decoder itself keeps all state in
   * registers and LDS */
  *temp_bytes = num_chunks == 0 ? 0 : lzl::kTicketBytes;
  return nvcompSuccess;
}

nvcompStatus_t nvcompBatchedSnappyDecompressGetTempSizeEx(
    size_t num_chunks, size_t max_uncompressed_chunk_bytes, size_t* temp_bytes, size_t /*max_total_uncompressed_bytes*/)
{
  return nvcompBatchedSnappyDecompressGetTempSize(num_chunks, max_uncompressed_chunk_bytes, temp_bytes);
}

nvcompStatus_t nvcompBatchedSnappyDecompressAsync(
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
  nvlog::call(3, "nvcompBatchedSnappyDecompressAsync(batch_size=%zu, statuses=%s, actual_sizes=%s, temp_bytes=%zu, stream=%p)",
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
  /* Bounds are checked whether or not the caller asked for statuses (round 4): the kernels without the checks were no
   * faster (655-668 against 675 GB/s on the headline batch over three evidence runs: the checks are a handful of
   * wave-uniform tests per batch), and a corrupt stream decoded with statuses == NULL could write past its output slot.
   * A NULL status array now only means that nobody is told: a failed chunk still reads 0 in
   * device_actual_uncompressed_bytes. */
  (void)device_statuses; /* (only the kernels look at it) */
  const lzl::Batch b = {device_compressed_ptrs, device_compressed_bytes, device_uncompressed_bytes,
                        device_actual_uncompressed_bytes, batch_size, device_uncompressed_ptrs, (int*)device_statuses};
  /* Small batches cannot fill the card with one wave per chunk: a workgroup per chunk (common/lz_team.hip.h), persistent
   * when there are more chunks than workgroups stay resident and the caller's temp buffer holds the ticket counter. */
  if (batch_size <= lzl::kTeamMaxBatch) {
    unsigned groups = (unsigned)batch_size;
    uint32_t* ticket = nullptr;
    const lzl::Launch one_each = {b, nullptr, (size_t)groups, nullptr};
    if (batch_size <= lzl::kTeam16MaxBatch) {
      /* at most one chunk per CU: sixteen waves a chunk (one team holds a whole CU's LDS budget for two) */
      hipLaunchKernelGGL((snappy_decompress_team_kernel<true, 16>), dim3(groups), dim3(1024), 0, stream, one_each);
      return launch_status();
    }
    if (device_temp_ptr != nullptr && temp_bytes >= sizeof(uint32_t) && ((uintptr_t)device_temp_ptr & 3u) == 0) {
      static lzl::ResidentCache resident; /* per device ordinal */
      const unsigned fit = resident.get(snappy_decompress_team_kernel<true, 8>, 512, 0);
      if (fit != 0 && fit < groups && hipMemsetAsync(device_temp_ptr, 0, sizeof(uint32_t), stream) == hipSuccess) {
        ticket = (uint32_t*)device_temp_ptr;
        groups = fit;
      }
    }
    const lzl::Launch launch = {b, ticket, (size_t)groups, nullptr};
    hipLaunchKernelGGL((snappy_decompress_team_kernel<true, 8>), dim3(groups), dim3(512), 0, stream, launch);
    return launch_status();
  }
  /* (round 2's path for small batches: two waves per chunk, producer / consumer) */
  if (batch_size <= lzl::kPairMaxBatch) {
    const dim3 pgrid((unsigned)batch_size), pblock(128);
    hipLaunchKernelGGL((snappy_decompress_pair_kernel<true>), pgrid, pblock, 0, stream, b);
    return launch_status();
  }
  /* Persistent waves when the caller's temp buffer holds the ticket counter: as many workgroups as stay resident. */
  unsigned groups = (unsigned)((batch_size + kDecWaves - 1) / kDecWaves);
  uint32_t* ticket = nullptr;
#if NVCOMP_LZ_PERSISTENT
  if (device_temp_ptr != nullptr && temp_bytes >= sizeof(uint32_t) && ((uintptr_t)device_temp_ptr & 3u) == 0) {
    static lzl::ResidentCache resident; /* per device ordinal */
    const unsigned fit = resident.get(snappy_decompress_window_kernel<true>, 64 * kDecWaves);
    if (fit != 0 && fit < groups && hipMemsetAsync(device_temp_ptr, 0, sizeof(uint32_t), stream) == hipSuccess) {
      ticket = (uint32_t*)device_temp_ptr;
      groups = fit;
    }
  }
#endif
  const lzl::Launch launch = {b, ticket, (size_t)groups * kDecWaves, nullptr};
  hipLaunchKernelGGL((snappy_decompress_window_kernel<true>), dim3(groups), dim3(64 * kDecWaves), 0, stream, launch);
  return launch_status();
}

nvcompStatus_t nvcompBatchedSnappyGetDecompressSizeAsync(
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
  hipLaunchKernelGGL(snappy_decompress_size_kernel, dim3(grid_for(batch_size)), dim3(64 * kWavesPerBlock), 0, stream,
                     device_compressed_ptrs, device_compressed_bytes, device_uncompressed_bytes, batch_size);
  return launch_status();
}

nvcompStatus_t nvcompBatchedSnappyCompressGetTempSize(
    size_t batch_size, size_t max_uncompressed_chunk_bytes, nvcompBatchedSnappyOpts_t format_opts, size_t* temp_bytes)
{
  if (temp_bytes == nullptr || !snappy_opts_ok(format_opts)) {
    return nvcompErrorInvalidValue;
  }
  if (max_uncompressed_chunk_bytes > nvcompSnappyCompressionMaxAllowedChunkSize) {
    return nvcompErrorChunkSizeTooLarge;
  }
  /* the per-chunk hash tables live in LDS; the scratch is the persistent waves' ticket counter (common/lz_launch.hip.h) */
  *temp_bytes = batch_size != 0 ? lzl::kTicketBytes : 0;
  return nvcompSuccess;
}

nvcompStatus_t nvcompBatchedSnappyCompressGetTempSizeEx(
    size_t batch_size,
    size_t max_uncompressed_chunk_bytes,
    nvcompBatchedSnappyOpts_t format_opts,
    size_t* temp_bytes,
    const size_t /*max_total_uncompressed_bytes*/)
{
  return nvcompBatchedSnappyCompressGetTempSize(batch_size, max_uncompressed_chunk_bytes, format_opts, temp_bytes);
}

nvcompStatus_t nvcompBatchedSnappyCompressGetMaxOutputChunkSize(
    size_t max_uncompressed_chunk_bytes, nvcompBatchedSnappyOpts_t format_opts, size_t* max_compressed_bytes)
{
  if (max_compressed_bytes == nullptr || !snappy_opts_ok(format_opts)) {
    return nvcompErrorInvalidValue;
  }
  if (max_uncompressed_chunk_bytes > nvcompSnappyCompressionMaxAllowedChunkSize) {
    return nvcompErrorChunkSizeTooLarge;
  }
  /* the raw format's classic bound: preamble + literal headers */
  *max_compressed_bytes = 32 + max_uncompressed_chunk_bytes + max_uncompressed_chunk_bytes / 6;
  return nvcompSuccess;
}

nvcompStatus_t nvcompBatchedSnappyCompressAsync(
    const void* const* device_uncompressed_ptrs,
    const size_t* device_uncompressed_bytes,
    size_t max_uncompressed_chunk_bytes,
    size_t batch_size,
    void* device_temp_ptr,
    size_t temp_bytes,
    void* const* device_compressed_ptrs,
    size_t* device_compressed_bytes,
    nvcompBatchedSnappyOpts_t format_opts,
    hipStream_t stream)
{
  nvlog::call(3, "nvcompBatchedSnappyCompressAsync(batch_size=%zu, max_uncompressed_chunk_bytes=%zu, stream=%p)", batch_size,
              max_uncompressed_chunk_bytes, (void*)stream);
  if (!snappy_opts_ok(format_opts)) {
    return nvcompErrorInvalidValue;
  }
  if (max_uncompressed_chunk_bytes > nvcompSnappyCompressionMaxAllowedChunkSize) {
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
  /* persistent waves when the caller's temp buffer holds the ticket counter: as many workgroups as stay resident */
#if NVCOMP_LZM_WIDE
  constexpr unsigned kWaves = kWideWaves;
  const auto kernel = snappy_compress_wide_kernel;
#else
  constexpr unsigned kWaves = kEncWaves;
  const auto kernel = snappy_compress_kernel;
#endif
  unsigned groups = (unsigned)((batch_size + kWaves - 1) / kWaves);
  uint32_t* ticket = nullptr;
  if (NVCOMP_LZ_PERSISTENT && device_temp_ptr != nullptr && temp_bytes >= sizeof(uint32_t) && ((uintptr_t)device_temp_ptr & 3u) == 0) {
    static lzl::ResidentCache resident; /* per device ordinal */
    const unsigned fit = resident.get(kernel, 64 * kWaves, 0);
    if (fit != 0 && fit < groups && hipMemsetAsync(device_temp_ptr, 0, sizeof(uint32_t), stream) == hipSuccess) {
      ticket = (uint32_t*)device_temp_ptr;
      groups = fit;
    }
  }
  const lzl::CompressLaunch launch = {device_uncompressed_ptrs, device_uncompressed_bytes, max_uncompressed_chunk_bytes,
                                      batch_size, device_compressed_ptrs, device_compressed_bytes, ticket,
                                      (size_t)groups * kWaves};
  hipLaunchKernelGGL(kernel, dim3(groups), dim3(64 * kWaves), 0, stream, launch);
  return launch_status();
}

} // extern "C"

#ifdef NVCOMP_LZW_PROF
/* Profiling builds only: read (and clear) the per-phase cycle sums of the Snappy window decoder. */
extern "C" int nvcompAmdProfReadSnappy(unsigned long long* host_slots, int n)
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
