/*
 * api/cascaded_api.hip -- C ABI of the batched Cascaded codec
 * (include/nvcomp/cascaded.h) and the kernels it launches.
 *
 * Chunk container (little endian, 4-byte aligned; DESIGN.md "Cascaded stream layout"):
 *   u32 'CASC' | u8 type, num_RLEs, num_deltas, use_bp | u32 uncompressed bytes
 *   | u32 sub-chunk bytes | u32 num_sub | u32 sub_end[num_sub] | sub-chunk payloads
 */
#include <hip/hip_runtime.h>

#include "nvcomp/cascaded.h"

#include "common/log.h"

#include "cascaded/cascaded.hip.h"

namespace {

constexpr uint32_t kHeaderBytes = 20;
/* Decode runs up to three passes with growing LDS per wave; a sub-chunk's need follows from its actual stream counts
 * (casc::decompress_sub), so compressible data is decoded by the first pass at full occupancy. The launches of a call
 * (round 6; nvcompBatchedCascadedDecompressAsync):
 *   up to 512 chunks      ONE: a workgroup per chunk at 16 KiB a wave, the last pass folded in (decode_one_chunk: fold)
 *   up to 4 096 chunks    two: a workgroup per chunk at 5 KiB a wave, then the chunks it flagged, folded as above
 *   beyond                three: a wave per chunk at 5 KiB, 16 KiB, 64 KiB
 * and every launch behind the first is as many workgroups as the card holds, looping over the flagged chunks -- not a grid
 * over the batch whose workgroups read a flag and leave. Float columns, old -> new on one box, alternating
 * (profiles/r06_ab_cascaded_launches.jsonl): 256 chunks 417 -> 464 GB/s, 1 024 chunks 1 240 -> 1 321, 4 096 chunks
 * 1 652 -> 1 706, 16 384 chunks 2 094 -> 2 149, 65 536 chunks 2 472 -> 2 496. */
/* The first pass of each direction: 4 waves per workgroup, its own LDS slice per wave and register budget (workgroups
 * per CU in __launch_bounds__). Swept on hardware in round 3 (profiles/archive/r03_cascaded_passes.jsonl, 1 GiB, compress /
 * decompress GB/s):
 *   compress   8 workgroups x 5 KiB (64 VGPRs, 16 spilled): float columns 690, int32 column 1 670, int64 key column 2 485
 *              6 workgroups x 6.25 KiB (85 VGPRs, 2 spilled): 785 / 1 666 / 2 712 -- and the float columns, whose sub-chunks
 *              have ~1 000 runs (2 KiB of run pool), stay in the first pass instead of being bounced to the last, every chunk
 *              5 workgroups x 7.75 KiB: 715 / 1 525 / 2 507
 *   decompress 8 x 5 KiB: 1 594 / 1 883 / 2 382;  7 x 5.5 KiB: 1 454 / 1 883 / 2 360;  6: 1 418 / 1 917;  5: 1 223 / 1 744 */
#ifndef NVCOMP_CASC_COMP_SMALL
#define NVCOMP_CASC_COMP_SMALL 6400
#endif
#ifndef NVCOMP_CASC_COMP_WGS
#define NVCOMP_CASC_COMP_WGS 6
#endif
constexpr uint32_t kCompSmallBudget = NVCOMP_CASC_COMP_SMALL;
#ifndef NVCOMP_CASC_DEC_TEAM_MAX_BATCH
#define NVCOMP_CASC_DEC_TEAM_MAX_BATCH 4096
#endif
constexpr size_t kDecTeamMaxBatch = NVCOMP_CASC_DEC_TEAM_MAX_BATCH;
#ifndef NVCOMP_CASC_DEC_SMALL
#define NVCOMP_CASC_DEC_SMALL (5 * 1024)
#endif
#ifndef NVCOMP_CASC_DEC_WGS
#define NVCOMP_CASC_DEC_WGS 8
#endif
constexpr uint32_t kDecSmallBudget = NVCOMP_CASC_DEC_SMALL;
constexpr uint32_t kMidBudget = 8 * 1024 + 512;   /* compress: a worst case up to here (4 KiB sub-chunks of >= 2-byte elements: value buffer +
                                                     one run pool) is the last pass itself; beyond it an intermediate pass of this size runs first */
constexpr uint32_t kFastBudget = 16 * 1024;       /* decode pass 1: 4 waves per workgroup (two value buffers + pools + marks: sub-chunks with
                                                     long runs; short-run ones expand in place inside pass 0, casc::rle_expand_inplace) */
constexpr uint32_t kBigBudget = 64 * 1024;        /* last pass: one wave per workgroup; also the compressor's limit */
#ifndef NVCOMP_CASC_CUS
#define NVCOMP_CASC_CUS 256 /* the passes behind the first are launched with the workgroups the card HOLDS (tests/emu models a card of four) */
#endif
constexpr size_t kCus = NVCOMP_CASC_CUS;
constexpr size_t kLdsPerCu = 160 * 1024;

void clear_stale_error()
{
  (void)hipGetLastError();
}

nvcompStatus_t launch_status()
{
  return hipGetLastError() == hipSuccess ? nvcompSuccess : nvcompErrorCudaError;
}

bool opts_ok(const nvcompBatchedCascadedOpts_t& o)
{
  if (o.type < NVCOMP_TYPE_CHAR || o.type > NVCOMP_TYPE_ULONGLONG) {
    return false;
  }
  const size_t w = (size_t)1 << ((unsigned)o.type >> 1);
  return o.num_RLEs >= 0 && o.num_RLEs <= 7 && o.num_deltas >= 0 && o.num_deltas <= 7 && (o.use_bp == 0 || o.use_bp == 1)
         && o.chunk_size >= 256 && o.chunk_size <= 16384 && o.chunk_size % w == 0;
}

/* One chunk, by the calling wave. */
__device__ __forceinline__ void compress_one_chunk(
    size_t chunk,
    uint32_t wv,
    uint8_t* lds,
    const void* const* __restrict__ in_ptrs,
    const size_t* __restrict__ in_bytes,
    void* const* __restrict__ out_ptrs,
    size_t* out_bytes,
    const casc::Params& p,
    uint32_t* todo,
    uint32_t pass,
    uint32_t last_pass,
    uint32_t lds_per_wave)
{
  const uint8_t* src = wave::uniform_ptr((const uint8_t*)in_ptrs[chunk]);
  uint8_t* dst = wave::uniform_ptr((uint8_t*)out_ptrs[chunk]);
  const uint32_t n_bytes = (uint32_t)wave::uniform64(in_bytes[chunk]);
  const uint32_t w = casc::type_width(p.type);
  const uint32_t lane = (uint32_t)wave::lane_id();
  if (n_bytes % w != 0 || wave::uniform64(in_bytes[chunk]) > p.max_bytes) { /* contract: whole elements, within the declared size */
    if (lane == 0) {
      out_bytes[chunk] = 0;
      if (pass == 0 && todo != nullptr) {
        todo[chunk] = 0;
      }
    }
    return;
  }
  const uint32_t num_sub = (n_bytes + p.sub_bytes - 1) / p.sub_bytes;
  if (lane == 0) {
    uint32_t* h = (uint32_t*)dst;
    h[0] = casc::kMagic;
    h[1] = p.type | (p.num_rles << 8) | (p.num_deltas << 16) | ((p.use_bp ? 1u : 0u) << 24);
    h[2] = n_bytes;
    h[3] = p.sub_bytes;
    h[4] = num_sub;
  }
  uint32_t* table = (uint32_t*)(dst + kHeaderBytes);
  uint8_t* payload = dst + kHeaderBytes + 4 * (size_t)num_sub;
  uint8_t* slice = lds + (size_t)wv * lds_per_wave;
  uint32_t pay = 0;
  bool deferred = false;
  for (uint32_t s = 0; s < num_sub; ++s) {
    const uint32_t off = s * p.sub_bytes;
    const uint32_t bytes = n_bytes - off < p.sub_bytes ? n_bytes - off : p.sub_bytes;
    uint32_t sz;
    switch (w) {
    case 1: sz = casc::compress_sub<uint8_t>(src + off, bytes, payload + pay, p, slice, lds_per_wave); break;
    case 2: sz = casc::compress_sub<uint16_t>(src + off, bytes, payload + pay, p, slice, lds_per_wave); break;
    case 4: sz = casc::compress_sub<uint32_t>(src + off, bytes, payload + pay, p, slice, lds_per_wave); break;
    default: sz = casc::compress_sub<uint64_t>(src + off, bytes, payload + pay, p, slice, lds_per_wave); break;
    }
    if (sz == casc::kSubNeedsLds) {
      deferred = true; /* the whole chunk is compressed again by the next pass; the host sized the last pass for the worst case */
      LZ_STAT("casc_compress_chunks_deferred", 1);
      LZ_STAT("casc_compress_subs_wasted", s);
      break;
    }
    LZ_STAT("casc_compress_subs", 1);
    pay += sz;
    if (lane == 0) {
      table[s] = pay;
    }
    wave::sync();
  }
  if (lane == 0) {
    if (deferred && pass < last_pass) {
      todo[chunk] = pass + 1;
    } else {
      out_bytes[chunk] = deferred ? 0 : kHeaderBytes + 4 * (size_t)num_sub + pay;
      if (pass == 0 && todo != nullptr) {
        todo[chunk] = 0;
      }
    }
  }
}

/* `later` = false: the grid covers the batch, a chunk per wave (pass 0, and the one launch of a call without flag words).
 * `later` = true, the passes behind it: as many workgroups as the card holds at their LDS size, wave g of G takes the chunks
 * g, g + G, ... whose todo word names the pass -- the scheme of cascaded_decompress_kernel below, for the same reason: on
 * the bench's columns the passes behind the first find nothing to do. */
template <bool later>
__global__ void __launch_bounds__(256, later ? 4 : NVCOMP_CASC_COMP_WGS) cascaded_compress_kernel(
    const void* const* __restrict__ in_ptrs,
    const size_t* __restrict__ in_bytes,
    size_t batch_size,
    void* const* __restrict__ out_ptrs,
    size_t* out_bytes,
    casc::Params p,
    uint32_t* todo,
    uint32_t pass,
    uint32_t last_pass,
    uint32_t lds_per_wave,
    uint32_t waves_per_block)
{
  WAVE_DYNAMIC_LDS(lds);
  const uint32_t wv = wave::uniform(threadIdx.x >> 6);
  const size_t first = (size_t)blockIdx.x * waves_per_block + wv;
  if (!later) {
    if (first < batch_size) {
      compress_one_chunk(first, wv, lds, in_ptrs, in_bytes, out_ptrs, out_bytes, p, todo, pass, last_pass, lds_per_wave);
    }
    return;
  }
  const uint32_t lane = (uint32_t)wave::lane_id();
  const size_t stride = (size_t)gridDim.x * waves_per_block;
  for (size_t base = first; base < batch_size; base += 64 * stride) {
    const size_t mine = base + lane * stride;
    const uint64_t work = wave::ballot(mine < batch_size && todo[mine] == pass);
    for (uint64_t m = work; m != 0; m &= m - 1) {
      compress_one_chunk(base + wave::ctz64(m) * stride, wv, lds, in_ptrs, in_bytes, out_ptrs, out_bytes, p, todo, pass, last_pass,
                         lds_per_wave);
    }
  }
}

/* pass p decodes the chunks with todo == p (pass 0: all) whose streams fit its LDS budget and hands the others
 * on by setting todo = p + 1.
 * `team` (round 4, batches of up to kDecTeamMaxBatch chunks): a WORKGROUP per chunk. The sub-chunks of a chunk are independent
 * streams (their ends are in the chunk's table), so the workgroup's four waves take them in turn -- wave v the sub-chunks
 * v, v + 4, ... -- instead of one wave walking all sixteen: a quarter of a chunk's latency and four times the waves for
 * batches that cannot fill the card (1 024 chunks of the float columns: 401 -> 906 GB/s, 256 chunks: 108 -> 280). The
 * waves' verdicts (error, "needs the next pass") meet in two LDS words. Large batches keep a chunk per wave: with the
 * card full, four waves each paying a chunk's header round trips for four sub-chunks are slower than one paying them for
 * sixteen (16 384 chunks: 1 426 against 1 673; profiles/r04_cascaded_ab.jsonl). */
/* One chunk, by the calling wave (team: by the calling workgroup, every wave with the same arguments).
 * `fold` (team only): a chunk the waves' slices are too small for is not handed to another launch -- the workgroup's first
 * wave decodes it again, alone, with the slices of all four as its one (the last pass's 64 KiB: the launches that fold give
 * their waves 16 KiB each). For the batches that are decoded a workgroup per chunk the launches ARE the time: 256 chunks of
 * the float columns took 38 us in three launches, two of which found nothing to do. */
template <bool team, bool fold = false>
__device__ __forceinline__ void decode_one_chunk(
    size_t chunk,
    uint32_t wv,
    uint32_t lane,
    uint8_t* lds,
    uint32_t* verdict,
    const void* const* __restrict__ comp_ptrs,
    const size_t* __restrict__ comp_bytes,
    const size_t* out_caps,
    size_t* actual_bytes,
    void* const* __restrict__ out_ptrs,
    nvcompStatus_t* statuses,
    uint32_t* todo,
    uint32_t pass,
    uint32_t lds_per_wave,
    uint32_t waves_per_block)
{
  const uint32_t first_sub = team ? wv : 0u, sub_stride = team ? waves_per_block : 1u;
  if (team) {
    if (threadIdx.x == 0) { /* the thread that read the words of the workgroup's last chunk (a later pass takes several) */
      verdict[0] = 0;
      verdict[1] = 0;
    }
    __syncthreads();
  }
  const uint8_t* src = wave::uniform_ptr((const uint8_t*)comp_ptrs[chunk]);
  uint8_t* dst = wave::uniform_ptr((uint8_t*)out_ptrs[chunk]);
  const size_t src_len = wave::uniform64(comp_bytes[chunk]);
  const size_t cap = wave::uniform64(out_caps[chunk]);
  uint32_t err = casc::kOk;
  uint32_t produced = 0;
  bool deferred = false;
  do {
    if (((uintptr_t)src & 3u) != 0) {
      err = casc::kErrAlign;
      break;
    }
    if (src_len < kHeaderBytes) {
      err = casc::kErrInput;
      break;
    }
    const uint32_t* h = (const uint32_t*)src;
    const uint32_t magic = wave::uniform(h[0]);
    const uint32_t cfg = wave::uniform(h[1]);
    const uint32_t n_bytes = wave::uniform(h[2]);
    const uint32_t sub = wave::uniform(h[3]);
    const uint32_t num_sub = wave::uniform(h[4]);
    const uint32_t type = cfg & 0xffu, num_rles = (cfg >> 8) & 0xffu, num_deltas = (cfg >> 16) & 0xffu;
    if (magic != casc::kMagic || type > 7 || num_rles > 7 || num_deltas > 7) {
      err = casc::kErrInput;
      break;
    }
    const uint32_t w = casc::type_width(type);
    if (sub == 0 || sub % w != 0 || n_bytes % w != 0 || sub / w > casc::kMaxElems
        || num_sub != (uint32_t)(((uint64_t)n_bytes + sub - 1) / sub) || src_len < kHeaderBytes + 4 * (uint64_t)num_sub) {
      err = casc::kErrInput;
      break;
    }
    if (n_bytes > cap) {
      err = casc::kErrOutput;
      break;
    }
    if (((uintptr_t)dst & (w - 1)) != 0) {
      err = casc::kErrAlign;
      break;
    }
    const uint32_t* table = (const uint32_t*)(src + kHeaderBytes);
    const uint8_t* payload = src + kHeaderBytes + 4 * (size_t)num_sub;
    const uint32_t pay_len = (uint32_t)(src_len - kHeaderBytes - 4 * (size_t)num_sub);
    uint8_t* slice = lds + (size_t)wv * lds_per_wave;
    /* The sub-chunk table is read 64 entries at a time (one coalesced load, then v_readlane), and the first
     * 256 bytes of the wave's NEXT sub-chunk -- all of its header words -- are requested before the current one is
     * decoded: per sub-chunk one memory round trip is exposed (the packed words) instead of three. */
    uint32_t tab = 0, tab_base = 0xffffffffu;
    auto table_at = [&](uint32_t i) -> uint32_t { /* table[i], i < num_sub (uniform) */
      if ((i & ~63u) != tab_base) {
        tab_base = i & ~63u;
        tab = tab_base + lane < num_sub ? table[tab_base + lane] : 0u;
      }
      return wave::read_lane(tab, i & 63u);
    };
    auto load_head = [&](uint32_t b, uint32_t e) -> uint32_t {
      const uint32_t lim = e < pay_len ? e : pay_len;
      return (b <= lim && 4 * lane + 4 <= lim - b) ? *(const uint32_t*)(payload + b + 4 * lane) : 0u;
    };
    uint32_t begin = 0, end = 0, head = 0;
    if (first_sub < num_sub) {
      begin = first_sub ? table_at(first_sub - 1) : 0u;
      end = table_at(first_sub);
      head = load_head(begin, end);
    }
    for (uint32_t s = first_sub; s < num_sub && !err && !deferred; s += sub_stride) {
      const uint32_t off = s * sub;
      const uint32_t bytes = n_bytes - off < sub ? n_bytes - off : sub;
      if (end < begin || end > pay_len || end - begin < 4) {
        err = casc::kErrInput;
        break;
      }
      uint32_t next_begin = 0, next_end = 0, next_head = 0;
      if (s + sub_stride < num_sub) {
        next_begin = table_at(s + sub_stride - 1);
        next_end = table_at(s + sub_stride);
        next_head = load_head(next_begin, next_end);
      }
      uint32_t rc;
      switch (w) {
      case 1:
        rc = casc::decompress_sub<uint8_t>(payload + begin, end - begin, head, dst + off, bytes, num_rles, num_deltas, slice, lds_per_wave);
        break;
      case 2:
        rc = casc::decompress_sub<uint16_t>(payload + begin, end - begin, head, dst + off, bytes, num_rles, num_deltas, slice, lds_per_wave);
        break;
      case 4:
        rc = casc::decompress_sub<uint32_t>(payload + begin, end - begin, head, dst + off, bytes, num_rles, num_deltas, slice, lds_per_wave);
        break;
      default:
        rc = casc::decompress_sub<uint64_t>(payload + begin, end - begin, head, dst + off, bytes, num_rles, num_deltas, slice, lds_per_wave);
        break;
      }
      LZ_STAT("casc_decompress_subs", 1);
      if (rc == casc::kSubNeedLds) {
        LZ_STAT("casc_decompress_subs_deferred", 1);
        if (pass < 2) {
          deferred = true; /* whatever was already written is decoded again by the next pass */
        } else {
          err = casc::kErrInput; /* larger than anything the compressor accepts */
        }
      } else if (rc != casc::kSubOk) {
        err = casc::kErrInput;
      }
      begin = next_begin;
      end = next_end;
      head = next_head;
      wave::sync();
    }
    produced = n_bytes;
  } while (false);
  if (team) {
    if (lane == 0 && (err || deferred)) {
      atomicOr(verdict + 0, err);
      atomicOr(verdict + 1, deferred ? 1u : 0u);
    }
    __syncthreads();
    if (fold) {
      /* (every thread reads the words: the caller gives consecutive chunks of a workgroup different ones, so that the reset
       * for the next chunk does not meet these reads) */
      if (wave::uniform(verdict[0]) == 0 && wave::uniform(verdict[1]) != 0) {
        if (wv == 0) { /* (the other waves wait at the next chunk's first barrier, or leave) */
          decode_one_chunk<false>(chunk, 0u, lane, lds, nullptr, comp_ptrs, comp_bytes, out_caps, actual_bytes, out_ptrs, statuses, todo,
                                  2u, waves_per_block * lds_per_wave, 1u);
        }
        return;
      }
    }
  }
  if (team ? threadIdx.x == 0 : lane == 0) {
    const uint32_t all_err = team ? verdict[0] : err;
    const bool any_deferred = (team ? verdict[1] != 0 : deferred) && all_err == 0; /* (an error is final whatever the other waves wanted) */
    if (pass == 0 || any_deferred) {
      todo[chunk] = any_deferred ? pass + 1 : 0u;
    }
    if (!any_deferred) {
      if (actual_bytes != nullptr) {
        actual_bytes[chunk] = all_err ? 0 : produced;
      }
      if (statuses != nullptr) {
        statuses[chunk] = all_err == casc::kOk ? nvcompSuccess
                          : (all_err & casc::kErrAlign) ? nvcompErrorAlignment
                                                        : nvcompErrorCannotDecompress;
      }
    }
  }
}

/* `later` = false, pass 0: the grid covers the batch, a chunk per wave (team: per workgroup). `later` = true, passes 1 and 2:
 * as many workgroups as the card holds at this pass's LDS size, whatever the batch -- wave g of G takes the chunks g, g + G,
 * ... whose todo word names this pass, and reads up to 64 of its todo words with ONE load (lane j: its j-th chunk's). On data
 * the first pass decodes -- the float columns, every column of the bench -- both later passes find nothing: as grids over the
 * batch (16 384 workgroups holding 64 KiB of LDS each for the last one) they took 4.9 + 8.4 us of a 1 GiB call's 510 and
 * 17.5 + 33 us of a 4 GiB call's 1 900 (profiles/r06_final_kernel_stats_cascaded_*.csv) to read one word per chunk and leave. */
/* (the later passes' workgroups hold 64 KiB of LDS: two per CU, and registers to match) */
template <bool team, bool later, bool fold = false> /* team: the workgroup's waves share ONE chunk; otherwise a chunk per wave */
__global__ void __launch_bounds__(256, (later || fold) ? 2 : NVCOMP_CASC_DEC_WGS) cascaded_decompress_kernel(
    const void* const* __restrict__ comp_ptrs,
    const size_t* __restrict__ comp_bytes,
    const size_t* out_caps,
    size_t* actual_bytes,
    size_t batch_size,
    void* const* __restrict__ out_ptrs,
    nvcompStatus_t* statuses,
    uint32_t* todo,
    uint32_t pass,
    uint32_t lds_per_wave,
    uint32_t waves_per_block)
{
  WAVE_DYNAMIC_LDS(lds);
  /* team: err bits of the chunk's waves | any wave deferred, BEHIND the waves' slices (the launch adds 16 bytes for them: as
   * a static array they cost the chunk-per-wave launch its eighth workgroup per CU -- 8 x (4 x 5 KiB + 16 B) > 160 KiB --
   * and 8 % of its speed) */
  uint32_t* verdict = (uint32_t*)(lds + (size_t)waves_per_block * lds_per_wave);
  const uint32_t wv = wave::uniform(threadIdx.x >> 6);
  const size_t first = team ? (size_t)blockIdx.x : (size_t)blockIdx.x * waves_per_block + wv;
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  static_assert(team || !fold, "folding is the workgroup-per-chunk launches'");
  if (!later) {
    if (first < batch_size) {
      decode_one_chunk<team, fold>(first, wv, lane, lds, verdict, comp_ptrs, comp_bytes, out_caps, actual_bytes, out_ptrs, statuses,
                                   todo, pass, lds_per_wave, waves_per_block);
    }
    return;
  }
  const size_t stride = team ? (size_t)gridDim.x : (size_t)gridDim.x * waves_per_block;
  uint32_t pair = 0; /* the verdict words alternate between two pairs (see decode_one_chunk: fold) */
  for (size_t base = first; base < batch_size; base += 64 * stride) {
    const size_t mine = base + lane * stride;
    const uint64_t work = wave::ballot(mine < batch_size && todo[mine] == pass);
    for (uint64_t m = work; m != 0; m &= m - 1) { /* (team: the same words in every wave of the workgroup) */
      decode_one_chunk<team, fold>(base + wave::ctz64(m) * stride, wv, lane, lds, verdict + pair, comp_ptrs, comp_bytes, out_caps,
                                   actual_bytes, out_ptrs, statuses, todo, pass, lds_per_wave, waves_per_block);
      pair ^= 2u;
    }
  }
}

__global__ void cascaded_size_kernel(
    const void* const* __restrict__ comp_ptrs, const size_t* __restrict__ comp_bytes, size_t* sizes, size_t batch_size)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= batch_size) {
    return;
  }
  const uint8_t* src = (const uint8_t*)comp_ptrs[i];
  size_t n = 0;
  if (comp_bytes[i] >= kHeaderBytes && ((uintptr_t)src & 3u) == 0) {
    const uint32_t* h = (const uint32_t*)src;
    if (h[0] == casc::kMagic) {
      n = h[2];
    }
  }
  sizes[i] = n;
}

} // namespace

extern "C" {

nvcompStatus_t nvcompBatchedCascadedCompressGetTempSize(
    size_t batch_size, size_t max_uncompressed_chunk_bytes, nvcompBatchedCascadedOpts_t format_opts, size_t* temp_bytes)
{
  if (temp_bytes == nullptr || !opts_ok(format_opts)) {
    return nvcompErrorInvalidValue;
  }
  if (max_uncompressed_chunk_bytes > nvcompCascadedCompressionMaxAllowedChunkSize) {
    return nvcompErrorChunkSizeTooLarge;
  }
  *temp_bytes = 4 * batch_size; /* one "repeat with a larger LDS slice" word per chunk */
  return nvcompSuccess;
}

nvcompStatus_t nvcompBatchedCascadedCompressGetMaxOutputChunkSize(
    size_t max_uncompressed_chunk_bytes, nvcompBatchedCascadedOpts_t format_opts, size_t* max_compressed_bytes)
{
  if (max_compressed_bytes == nullptr || !opts_ok(format_opts)) {
    return nvcompErrorInvalidValue;
  }
  if (max_uncompressed_chunk_bytes > nvcompCascadedCompressionMaxAllowedChunkSize) {
    return nvcompErrorChunkSizeTooLarge;
  }
  /* worst case: every sub-chunk stored raw (4-byte marker + bytes padded to 4) */
  const size_t num_sub = (max_uncompressed_chunk_bytes + format_opts.chunk_size - 1) / format_opts.chunk_size;
  *max_compressed_bytes = kHeaderBytes + 4 * num_sub + 8 * num_sub + ((max_uncompressed_chunk_bytes + 3) & ~(size_t)3);
  return nvcompSuccess;
}

nvcompStatus_t nvcompBatchedCascadedCompressAsync(
    const void* const* device_uncompressed_ptrs,
    const size_t* device_uncompressed_bytes,
    size_t max_uncompressed_chunk_bytes,
    size_t batch_size,
    void* device_temp_ptr,
    size_t temp_bytes,
    void* const* device_compressed_ptrs,
    size_t* device_compressed_bytes,
    nvcompBatchedCascadedOpts_t format_opts,
    hipStream_t stream)
{
  nvlog::call(3, "nvcompBatchedCascadedCompressAsync(batch_size=%zu, max_uncompressed_chunk_bytes=%zu, stream=%p)", batch_size,
              max_uncompressed_chunk_bytes, (void*)stream);
  if (!opts_ok(format_opts)) {
    return nvcompErrorInvalidValue;
  }
  if (max_uncompressed_chunk_bytes > nvcompCascadedCompressionMaxAllowedChunkSize) {
    return nvcompErrorChunkSizeTooLarge;
  }
  if (batch_size == 0) {
    return nvcompSuccess;
  }
  if (device_uncompressed_ptrs == nullptr || device_uncompressed_bytes == nullptr || device_compressed_ptrs == nullptr
      || device_compressed_bytes == nullptr) {
    return nvcompErrorInvalidValue;
  }
  casc::Params p;
  p.sub_bytes = (uint32_t)format_opts.chunk_size;
  p.type = (uint32_t)format_opts.type;
  p.num_rles = (uint32_t)format_opts.num_RLEs;
  p.num_deltas = (uint32_t)format_opts.num_deltas;
  p.use_bp = (uint32_t)format_opts.use_bp;
  p.max_bytes = (uint32_t)max_uncompressed_chunk_bytes;
  const uint32_t w = 1u << (p.type >> 1);
  if (casc::decompress_lds_per_wave(p.sub_bytes, w, p.num_rles) + 64 > kBigBudget) { /* what the decoder may need */
    return nvcompErrorNotSupported;
  }
  const uint32_t per_wave = (casc::compress_lds_per_wave(p.sub_bytes, w, p.num_rles) + 15u) & ~15u;
  uint32_t waves = kBigBudget / per_wave;
  waves = waves > 4 ? 4 : waves;
  const unsigned grid = (unsigned)((batch_size + waves - 1) / waves);
  clear_stale_error();
  uint32_t* todo = (uint32_t*)device_temp_ptr;
  if (todo == nullptr || temp_bytes < 4 * batch_size || per_wave <= kCompSmallBudget) {
    /* no flag words (or nothing to gain): one launch sized for the worst case */
    hipLaunchKernelGGL(cascaded_compress_kernel<false>, dim3(grid), dim3(64 * waves), per_wave * waves, stream,
                       device_uncompressed_ptrs, device_uncompressed_bytes, batch_size, device_compressed_ptrs,
                       device_compressed_bytes, p, (uint32_t*)nullptr, 0u, 0u, per_wave, waves);
    return launch_status();
  }
  /* pass 0: a small LDS slice at full occupancy; chunks whose streams overflow it are flagged and compressed again by
   * the next pass; the last pass holds the worst case. The passes behind the first: the workgroups the card holds at their
   * LDS size, looping (see the kernel) */
  auto resident = [](size_t wgs, size_t lds_bytes) -> unsigned {
    const size_t fit = kCus * (kLdsPerCu / lds_bytes);
    return (unsigned)(wgs < fit ? wgs : fit);
  };
  const uint32_t last = per_wave > kMidBudget ? 2u : 1u;
  hipLaunchKernelGGL(cascaded_compress_kernel<false>, dim3((unsigned)((batch_size + 3) / 4)), dim3(256), 4 * kCompSmallBudget, stream,
                     device_uncompressed_ptrs, device_uncompressed_bytes, batch_size, device_compressed_ptrs,
                     device_compressed_bytes, p, todo, 0u, last, kCompSmallBudget, 4u);
  if (last == 2) {
    hipLaunchKernelGGL(cascaded_compress_kernel<true>, dim3(resident((batch_size + 3) / 4, 4 * kMidBudget)), dim3(256), 4 * kMidBudget,
                       stream, device_uncompressed_ptrs, device_uncompressed_bytes, batch_size, device_compressed_ptrs,
                       device_compressed_bytes, p, todo, 1u, last, kMidBudget, 4u);
  }
  hipLaunchKernelGGL(cascaded_compress_kernel<true>, dim3(resident(grid, (size_t)per_wave * waves)), dim3(64 * waves), per_wave * waves,
                     stream, device_uncompressed_ptrs, device_uncompressed_bytes, batch_size, device_compressed_ptrs,
                     device_compressed_bytes, p, todo, last, last, per_wave, waves);
  return launch_status();
}

nvcompStatus_t nvcompBatchedCascadedDecompressGetTempSize(
    size_t num_chunks, size_t /*max_uncompressed_chunk_bytes*/, size_t* temp_bytes)
{
  if (temp_bytes == nullptr) {
    return nvcompErrorInvalidValue;
  }
  *temp_bytes = 4 * num_chunks; /* one "needs the large-LDS pass" word per chunk */
  return nvcompSuccess;
}

nvcompStatus_t nvcompBatchedCascadedDecompressAsync(
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
  nvlog::call(3, "nvcompBatchedCascadedDecompressAsync(batch_size=%zu, statuses=%s, actual_sizes=%s, stream=%p)", batch_size,
              device_statuses ? "yes" : "null", device_actual_uncompressed_bytes ? "yes" : "null", (void*)stream);
  if (batch_size == 0) {
    return nvcompSuccess;
  }
  if (device_compressed_ptrs == nullptr || device_compressed_bytes == nullptr || device_uncompressed_bytes == nullptr
      || device_uncompressed_ptrs == nullptr || device_temp_ptr == nullptr || temp_bytes < 4 * batch_size) {
    return nvcompErrorInvalidValue;
  }
  uint32_t* todo = (uint32_t*)device_temp_ptr;
  clear_stale_error();
  /* passes 1 and 2: the workgroups the card holds at 64 KiB of LDS each (two per CU), looping (see the kernel) */
  const size_t kLaterGrid = kCus * (kLdsPerCu / kBigBudget);
  if (batch_size <= kDecTeamMaxBatch) {
    /* a workgroup per chunk, and the last pass folded into the one before it (decode_one_chunk: fold); batches the card
     * holds at 64 KiB of LDS a workgroup start there: ONE launch */
    const dim3 grid((unsigned)batch_size), later((unsigned)(batch_size < kLaterGrid ? batch_size : kLaterGrid));
    const unsigned extra = 16; /* the team's verdict words, two pairs */
    (void)extra;               /* (the host emulation's launch macro has no LDS size) */
    if (batch_size <= kLaterGrid) {
      hipLaunchKernelGGL((cascaded_decompress_kernel<true, false, true>), grid, dim3(256), 4 * kFastBudget + extra,
                         stream, device_compressed_ptrs, device_compressed_bytes, device_uncompressed_bytes,
                         device_actual_uncompressed_bytes, batch_size, device_uncompressed_ptrs, device_statuses, todo, 1u,
                         kFastBudget, 4u);
      return launch_status();
    }
    hipLaunchKernelGGL((cascaded_decompress_kernel<true, false>), grid, dim3(256), 4 * kDecSmallBudget + extra,
                       stream, device_compressed_ptrs, device_compressed_bytes, device_uncompressed_bytes,
                       device_actual_uncompressed_bytes, batch_size, device_uncompressed_ptrs, device_statuses, todo, 0u,
                       kDecSmallBudget, 4u);
    hipLaunchKernelGGL((cascaded_decompress_kernel<true, true, true>), later, dim3(256), 4 * kFastBudget + extra,
                       stream, device_compressed_ptrs, device_compressed_bytes, device_uncompressed_bytes,
                       device_actual_uncompressed_bytes, batch_size, device_uncompressed_ptrs, device_statuses, todo, 1u,
                       kFastBudget, 4u);
    return launch_status();
  } else {
    const size_t wgs = (batch_size + 3) / 4;
    const dim3 grid((unsigned)wgs), later((unsigned)(wgs < kLaterGrid ? wgs : kLaterGrid));
    /* (four waves a workgroup: one and two measured the same, profiles/r06_ab_cascaded_launches.jsonl r6cascw) */
    hipLaunchKernelGGL((cascaded_decompress_kernel<false, false>), grid, dim3(256), 4 * kDecSmallBudget,
                       stream, device_compressed_ptrs, device_compressed_bytes, device_uncompressed_bytes,
                       device_actual_uncompressed_bytes, batch_size, device_uncompressed_ptrs, device_statuses, todo, 0u,
                       kDecSmallBudget, 4u);
    hipLaunchKernelGGL((cascaded_decompress_kernel<false, true>), later, dim3(256), 4 * kFastBudget,
                       stream, device_compressed_ptrs, device_compressed_bytes, device_uncompressed_bytes,
                       device_actual_uncompressed_bytes, batch_size, device_uncompressed_ptrs, device_statuses, todo, 1u,
                       kFastBudget, 4u);
  }
  hipLaunchKernelGGL((cascaded_decompress_kernel<false, true>), dim3((unsigned)(batch_size < kLaterGrid ? batch_size : kLaterGrid)),
                     dim3(64), kBigBudget, stream, device_compressed_ptrs, device_compressed_bytes, device_uncompressed_bytes,
                     device_actual_uncompressed_bytes, batch_size, device_uncompressed_ptrs, device_statuses, todo, 2u,
                     kBigBudget, 1u);
  return launch_status();
}

nvcompStatus_t nvcompBatchedCascadedGetDecompressSizeAsync(
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
  hipLaunchKernelGGL(cascaded_size_kernel, dim3((unsigned)((batch_size + 255) / 256)), dim3(256), 0, stream,
                     device_compressed_ptrs, device_compressed_bytes, device_uncompressed_bytes, batch_size);
  return launch_status();
}

nvcompStatus_t nvcompBatchedCascadedCompressGetTempSizeEx(
    size_t batch_size,
    size_t max_uncompressed_chunk_bytes,
    nvcompBatchedCascadedOpts_t format_opts,
    size_t* temp_bytes,
    const size_t /*max_total_uncompressed_bytes*/)
{
  /* the scratch does not depend on the batch's total size */
  return nvcompBatchedCascadedCompressGetTempSize(batch_size, max_uncompressed_chunk_bytes, format_opts, temp_bytes);
}

nvcompStatus_t nvcompBatchedCascadedDecompressGetTempSizeEx(
    size_t num_chunks, size_t max_uncompressed_chunk_bytes, size_t* temp_bytes, size_t /*max_total_uncompressed_bytes*/)
{
  return nvcompBatchedCascadedDecompressGetTempSize(num_chunks, max_uncompressed_chunk_bytes, temp_bytes);
}

} // extern "C"

#ifdef NVCOMP_CASC_PROF
/* Profiling builds only: read (and clear) the per-phase cycle sums of the Cascaded decoder. */
extern "C" int nvcompAmdCascProfRead(unsigned long long* host_slots, int n)
{
  static unsigned long long v[casc::kProfSlots * 64];
  if (hipMemcpyFromSymbol(v, HIP_SYMBOL(casc::g_prof), sizeof(v)) != hipSuccess) {
    return -1;
  }
  for (int i = 0; i < n && i < (int)casc::kProfSlots; ++i) {
    host_slots[i] = 0;
    for (int k = 0; k < 64; ++k) {
      host_slots[i] += v[i * 64 + k];
    }
  }
  static const unsigned long long z[casc::kProfSlots * 64] = {};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(casc::g_prof), z, sizeof(z));
  return (int)casc::kProfSlots;
}
#endif
