/*
 * common/lz_launch.hip.h -- how the batched LZ decoders (LZ4, Snappy) put a batch on the card.
 *
 * PERSISTENT WAVES. One wavefront decodes one chunk at a time, but a launch has only as many workgroups as the
 * card keeps resident; a wave that finishes its chunk takes the next one from a ticket counter in the caller's temp
 * buffer. Chunks of one batch take very different times (the mix: 5x), and with one chunk per wave and four waves per
 * workgroup a workgroup's LDS and wave slots stay allocated until its SLOWEST chunk ends; one-wave workgroups fix that
 * and pay four times as many workgroup launches instead (measured in round 2: +2 % on the mix, -7 % on uniformly fast
 * chunks). Persistent waves have neither cost. The counter is cleared by a 4-byte hipMemsetAsync on the caller's
 * stream in front of the kernel; without a temp buffer (a caller that passes NULL) the launch falls back to one
 * wave per chunk, statically.
 *
 * PATH BY BATCH SIZE. Batches of at most kTeamMaxBatch chunks cannot fill the card with one wave per chunk and run a
 * WORKGROUP of eight waves per chunk (common/lz_team.hip.h); kPairMaxBatch is the same for round 2's two waves per chunk
 * (producer / consumer, lz4_decode_window.hip.h: pair), which the team supersedes where both apply. The thresholds are
 * compile-time constants: the library has no run-time tuning state (tests force each path with an A/B build of this
 * file's macros).
 */
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "common/wave.h"

#ifndef NVCOMP_LZ_PAIR_MAX_BATCH
#define NVCOMP_LZ_PAIR_MAX_BATCH 4096 /* two waves per chunk up to here: 319 against 310 GB/s at 4 096 chunks since the pair kernels lost their scratch
                                        * (round 4, profiles/r04_final_nsweep.jsonl; 3 072 before); 412 against 454 at 8 192. Round 6: the kernels hold
                                        * the one-wave loop with the run executor for the chunks that shrank 8 x (NVCOMP_LZ_PAIR_SOLO), whose
                                        * registers cost them the eighth wave per SIMD -- 7 168 resident waves for 8 192 at 4 096 chunks -- and the
                                        * mix at 4 096 chunks is FASTER so: 328 against 317 GB/s (and 301 in the persistent kernel; gpurun r6ak) */
#endif
#ifndef NVCOMP_LZ_PAIR_SOLO
#define NVCOMP_LZ_PAIR_SOLO 1 /* A/B: 0 = every chunk of the two-wave kernels by producer and consumer */
#endif
#ifndef NVCOMP_LZ_TEAM_MAX_BATCH
#define NVCOMP_LZ_TEAM_MAX_BATCH 512 /* a WORKGROUP per chunk up to this many chunks (common/lz_team.hip.h): 263 us per chunk against 467 (two waves) and 677 (one); from 1 024 chunks on two waves per chunk keep more chunks in flight (profiles/r04_team.jsonl) */
#endif
#ifndef NVCOMP_LZ_TEAM16_MAX_BATCH
#define NVCOMP_LZ_TEAM16_MAX_BATCH 256 /* ... of which batches of at most one chunk per CU get teams of sixteen waves */
#endif
#ifndef NVCOMP_LZ_MAX_WG_PER_CU
#define NVCOMP_LZ_MAX_WG_PER_CU 7 /* cap on the persistent workgroups (of four waves) per CU; 0 = as many as stay resident.
                                   * Measured on MI355X (profiles/archive/r03_ab_g.jsonl): the decoders fit 8 waves/SIMD since they stopped
                                   * spilling, and run SLOWER there than at 7 (LZ4 mix 632 vs 644 GB/s, sorted-key column 686 vs
                                   * 750): 28 waves per CU already keep the vector, scalar and LDS pipes two thirds busy, four more
                                   * only add contention for L1 / L2 and the LDS pipe; 6 is worse again (607). */
#endif
#ifndef NVCOMP_LZ_PERSISTENT
#define NVCOMP_LZ_PERSISTENT 1 /* A/B: 0 = one wave per chunk, static mapping (round 2) */
#endif

namespace lzl {

constexpr size_t kPairMaxBatch = (size_t)(NVCOMP_LZ_PAIR_MAX_BATCH);
constexpr size_t kTeamMaxBatch = (size_t)(NVCOMP_LZ_TEAM_MAX_BATCH);
constexpr size_t kTeam16MaxBatch = (size_t)(NVCOMP_LZ_TEAM16_MAX_BATCH) < kTeamMaxBatch ? (size_t)(NVCOMP_LZ_TEAM16_MAX_BATCH) : kTeamMaxBatch;
constexpr uint32_t kMaxOutCap = 1u << 26;
constexpr size_t kTicketBytes = 16; /* what the temp-size queries ask for: one u32 counter, padded */
/* ... and, for the persistent one-wave-per-chunk launches, room for every wave's token index (common/lz_index.hip.h:
 * 64 lists of 344 positions of 16 bits) behind it. The queries cannot ask the device how many waves stay resident: they assume at
 * most kIndexMaxWaves (MI355X: 256 CUs x 28 = 7 168); a launch with more waves, or a caller with a smaller buffer, decodes
 * without the index (same bytes, the round-5 speed). */
constexpr size_t kIndexOffset = 64;
constexpr size_t kIndexBytesPerWave = 45056;
constexpr size_t kIndexMaxWaves = 8192;
inline size_t index_temp_bytes(size_t num_chunks)
{
  const size_t waves = (num_chunks + 3) & ~(size_t)3;
  return kIndexOffset + (waves < kIndexMaxWaves ? waves : kIndexMaxWaves) * kIndexBytesPerWave;
}
/* the index slices of a launch of `waves` waves inside the caller's temp buffer, or NULL when it has no room for them */
inline uint8_t* index_base(void* temp, size_t temp_bytes, size_t waves)
{
  if (temp == nullptr || ((uintptr_t)temp & 7u) != 0 || temp_bytes < kIndexOffset + waves * kIndexBytesPerWave) {
    return nullptr;
  }
  return (uint8_t*)temp + kIndexOffset;
}

/* The caller's arrays of one nvcompBatched<Fmt>DecompressAsync call. */
struct Batch
{
  const void* const* comp_ptrs;
  const size_t* comp_bytes;
  const size_t* out_caps;
  size_t* actual_bytes;
  size_t batch_size;
  void* const* out_ptrs;
  int* statuses; /* nvcompStatus_t* */
};

/* The single parameter of the persistent kernels: the batch, the ticket counter (NULL: one chunk per wave, statically) and
 * the first chunk index handed out by ticket (= waves of the launch). Read through wave::kernel_args where needed. */
struct Launch
{
  Batch b;
  uint32_t* ticket;
  size_t first_dynamic;
  uint8_t* index;        /* kIndexBytesPerWave bytes per wave of the launch for the token index (common/lz_index.hip.h); NULL: no index */
};

/* The same for the compressors: the caller's arrays of one nvcompBatched<Fmt>CompressAsync call. */
struct CompressLaunch
{
  const void* const* in_ptrs;
  const size_t* in_bytes;
  size_t max_chunk_bytes;
  size_t batch_size;
  void* const* out_ptrs;
  size_t* out_bytes;
  uint32_t* ticket;
  size_t first_dynamic;
};

/* The wave's next chunk: `first_dynamic` + a ticket (lane 0 draws it, the wave shares it). */
__device__ __forceinline__ size_t next_chunk(uint32_t* ticket, size_t first_dynamic)
{
  uint32_t t = 0;
  if (wave::lane_id() == 0) {
    t = atomicAdd(ticket, 1u);
  }
  return first_dynamic + wave::uniform(wave::read_lane(t, 0));
}

/* Workgroups of `kernel` (block threads, static LDS) the current device keeps resident at once; 0 when the runtime
 * cannot tell (the launch is then static). */
template <class Kernel>
inline unsigned resident_workgroups(Kernel kernel, unsigned block_threads, int max_per_cu = NVCOMP_LZ_MAX_WG_PER_CU, int* device = nullptr)
{
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess
      || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess
      || hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, (int)block_threads, 0) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  if (device != nullptr) {
    *device = dev;
  }
  if (max_per_cu > 0 && per_cu > max_per_cu) {
    per_cu = max_per_cu;
  }
  return cus > 0 && per_cu > 0 ? (unsigned)cus * (unsigned)per_cu : 0u;
}

/* The same, remembered PER DEVICE ORDINAL (one instance per kernel, a function-local static of the caller): a process
 * that drives several cards from one thread (benchmarks/benchmark_allgather.cpp: hipSetDevice in a loop) gets each
 * card's own geometry, and the two runtime queries are made once per kernel and card. Lock-free; two threads racing on
 * the first call both ask and store the same answer. */
struct ResidentCache
{
  static constexpr int kDevices = 64;
  std::atomic<unsigned> known[kDevices] = {}; /* answer + 1; 0 = not asked yet */
  template <class Kernel>
  unsigned get(Kernel kernel, unsigned block_threads, int max_per_cu = NVCOMP_LZ_MAX_WG_PER_CU)
  {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
      (void)hipGetLastError();
      return 0;
    }
    if (dev < 0 || dev >= kDevices) {
      return resident_workgroups(kernel, block_threads, max_per_cu);
    }
    const unsigned seen = known[dev].load(std::memory_order_relaxed);
    if (seen != 0) {
      return seen - 1;
    }
    const unsigned fit = resident_workgroups(kernel, block_threads, max_per_cu);
    known[dev].store(fit + 1, std::memory_order_relaxed);
    return fit;
  }
};

} // namespace lzl
