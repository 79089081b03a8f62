/*
 * common/lz_window.hip.h -- LDS-staged, sequence-parallel LZ77 batch executor
 * shared by the LZ4 and Snappy decoders (one wavefront per chunk).
 *
 * Measured on MI355X (profiles/archive/r01_v1_direct_pmc.json): decoding straight to
 * HBM leaves a wave waiting on memory 62 % of its cycles -- every literal/match
 * step is a dependent global load->store round trip -- and reads 10x the
 * algorithmic bytes (partial-line writes and far match sources miss the L2).
 * So everything a sequence can depend on is kept in the CU's LDS:
 *
 *   input ring   (kInRing bytes)  the compressed stream, loaded from HBM in
 *                1 KiB blocks of 16-byte lane loads, ahead of the token chase;
 *   output window (kOutWin bytes) a linear window over the most recent output.
 *                Literals and matches are assembled here; a match whose source
 *                is still inside the window never touches HBM. After every
 *                batch the new bytes are flushed to HBM with 16-byte aligned,
 *                fully coalesced stores, and when the window fills up its last
 *                kKeep bytes slide to the front.
 *
 * Only matches that reach further back than the window ("far") read HBM, all of
 * a batch's far reads being issued together before the LDS work starts.
 */
#pragma once

#include "common/lz_common.hip.h"

namespace lzw {

/* Tunables (overridable with -D for the A/B builds of scripts/build_variants.sh). */
/* The output window holds ONE BATCH: the bytes a batch produces (at most kBatchMax) plus the tail of the previous one
 * that still waits for its 16-byte block to fill up -- and 32 bytes of history, no more. Measured on MI355X (profiles/archive/r03_ab_*.jsonl): 84 % of
 * the matches of the headline workload reach further back than any window that fits the LDS budget (61 % further than
 * 4 KiB), so history only turned a twentieth of the far matches into near ones, and paid for it with 1 KiB of LDS per
 * wave and a slide of 768 bytes every other batch: without it +1.6 % (LZ4 mix), +2.4 % (Snappy), +7 % (text, 1 GiB
 * batches). Rounds 1-2 kept 768 bytes of a 2 KiB window (profiles/archive/r01_window_variants.json). A batch may produce up to
 * 2 KiB: 5 712 B of LDS per wave still fit seven waves per SIMD (163 840 / 28), 64 sequences of text never get there,
 * and data with long matches pays the per-batch costs half as often (sorted-key column 749 -> 887 GB/s, int32 column
 * 684 -> 778, profiles/archive/r03_ab_h.jsonl). */
#ifndef NVCOMP_LZW_BATCHMAX
#define NVCOMP_LZW_BATCHMAX 2048
#endif
#ifndef NVCOMP_LZW_KEEP
#define NVCOMP_LZW_KEEP 32 /* the last 32 bytes stay: what the shortest offsets reach (Snappy's one-byte-offset copies, the
                             * patterns of runs that go on across a batch boundary): Snappy mix +3 %, LZ4 mix +1 % */
#endif
#ifndef NVCOMP_LZW_OUTWIN
#define NVCOMP_LZW_OUTWIN (NVCOMP_LZW_BATCHMAX + 64 + NVCOMP_LZW_KEEP)
#endif
#ifndef NVCOMP_LZW_INRING
#define NVCOMP_LZW_INRING 2048
#endif
#ifndef NVCOMP_LZW_FAR_L4_FROM_F0
#define NVCOMP_LZW_FAR_L4_FROM_F0 0 /* A/B: the last dword of a far match of up to 16 bytes out of its first load */
#endif
/* (Round 4 measured the far-match loads non-temporal -- inline-asm loads waited for by hand --: 663 -> 412 GB/s,
 * docs/HISTORY.md 4; the path is gone.) */
#ifndef NVCOMP_LZW_FLUSH_ALIGN
#define NVCOMP_LZW_FLUSH_ALIGN 16 /* a batch's flush ends on this address boundary (16 | 32 | 64 | 128); the rest waits in the window */
#endif
#ifndef NVCOMP_LZW_WAVES_PER_SIMD
#define NVCOMP_LZW_WAVES_PER_SIMD 7
#endif

constexpr uint32_t kOutWin = NVCOMP_LZW_OUTWIN;     /* bytes of output window per wave */
constexpr uint32_t kBatchMax = NVCOMP_LZW_BATCHMAX; /* most output bytes one batch may produce */
constexpr uint32_t kKeep = NVCOMP_LZW_KEEP;         /* history kept when the window slides */
constexpr uint32_t kInRing = NVCOMP_LZW_INRING;     /* bytes of compressed-stream ring per wave */
constexpr uint32_t kInBlock = kInRing / 2; /* ring refill granule: 16 bytes per lane, 64 lanes (1 KiB) or 32 (deflate's 512-byte blocks) */
static_assert(kInBlock == 1024 || kInBlock == 512, "in_load_block: one 16-byte load by the first kInBlock / 16 lanes");
constexpr uint32_t kFlushAlign = NVCOMP_LZW_FLUSH_ALIGN;
constexpr uint32_t kOutLds = kOutWin + 32;
constexpr uint32_t kInLds = kInRing + 16; /* first 16 bytes mirrored after the end */
/* Profiling builds only (wrong output by construction): bit mask of execute stages to leave out.
 * 1 = HBM flush stores, 2 = far-match HBM loads, 4 = literal copies, 8 = in-window match rounds, 16 = far data into LDS */
#ifndef NVCOMP_LZW_ABLATE_EXEC
#define NVCOMP_LZW_ABLATE_EXEC 0
#endif
#ifndef NVCOMP_LZW_CHASE_ENOUGH
#define NVCOMP_LZW_CHASE_ENOUGH 64 /* tokens in hand from which the chase does not open another window (A/B: 40, 48) */
#endif
constexpr uint32_t kChaseEnough = NVCOMP_LZW_CHASE_ENOUGH;
constexpr uint32_t kChaseWin = 256;                 /* stream positions one chase window covers */
constexpr uint32_t kChaseLevels = 5;                /* jump tables for 1, 2, 4, 8, 16 tokens ahead; 32 = two steps of the
                                                     * last one from the start position (chase_tokens): no sixth round */
constexpr uint32_t kChaseLds = kChaseLevels * kChaseWin;
constexpr uint32_t kLdsPerWave = kOutLds + kInLds + kChaseLds;

constexpr uint32_t kLitShort = 32;   /* lane-parallel literal runs: up to 8 dwords */
constexpr uint32_t kMatchShort = 32; /* lane-parallel matches:      up to 8 dwords */

/* ---- phase clock (profiling builds only: -DNVCOMP_LZW_PROF) ------------------ */
#ifdef NVCOMP_LZW_PROF
constexpr uint32_t kProfSlots = 20;
__device__ unsigned long long g_prof[kProfSlots];
__device__ __forceinline__ unsigned long long* prof_slots()
{
  __shared__ unsigned long long slots[16][kProfSlots + 1];
  return slots[threadIdx.x >> 6];
}
__device__ __forceinline__ void prof_begin()
{
  if (wave::lane_id() == 0) {
    unsigned long long* p = prof_slots();
    for (uint32_t i = 0; i < kProfSlots; ++i) {
      p[i] = 0;
    }
    p[kProfSlots] = __builtin_readcyclecounter();
  }
}
__device__ __forceinline__ void prof_mark(uint32_t slot)
{
  if (wave::lane_id() == 0) {
    unsigned long long* p = prof_slots();
    const unsigned long long t = __builtin_readcyclecounter();
    p[slot] += t - p[kProfSlots];
    p[kProfSlots] = t;
  }
}
__device__ __forceinline__ void prof_end()
{
  if (wave::lane_id() == 0) {
    unsigned long long* p = prof_slots();
    for (uint32_t i = 0; i < kProfSlots; ++i) {
      atomicAdd(&g_prof[i], p[i]);
    }
  }
}
#define LZW_T(slot) lzw::prof_mark(slot)
#else
#define LZW_T(slot) ((void)0)
#endif

/* ---- compressed-stream ring ------------------------------------------------ */

struct InRing
{
  static constexpr uint32_t kMask = kInRing - 1; /* ring index of virtual position v = v & kMask */
  const uint8_t* base; /* chunk pointer rounded down to 16 bytes (uniform) */
  uint8_t* ring;       /* LDS, kInLds bytes, 16-byte aligned */
  uint32_t vbeg;       /* virtual position of the first chunk byte: chunk & 15 */
  uint32_t vend;       /* vbeg + chunk length */
  uint32_t lo, hi;     /* resident virtual range [lo, hi): multiples of kInBlock */
};

__device__ __forceinline__ void in_init(InRing& r, const uint8_t* in, uint32_t in_len, uint8_t* lds)
{
  const uint32_t a = (uint32_t)((uintptr_t)in & 15u);
  r.base = in - a;
  r.ring = lds;
  r.vbeg = a;
  r.vend = a + in_len;
  r.lo = 0;
  r.hi = 0;
}

/* Load the block of kInBlock bytes starting at virtual position vb (multiple of kInBlock)
 * into the ring. Bytes outside the chunk are never fetched (read as zero). */
__device__ __forceinline__ void in_load_block(InRing& r, uint32_t vb)
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  if (kInBlock < 1024 && 16u * lane >= kInBlock) {
    return;
  }
  const uint32_t v = vb + 16u * lane;
  wave::u32x4 x = {0, 0, 0, 0};
  if (v >= r.vbeg && v + 16 <= r.vend) {
    x = wave::gload_u32x4_aligned(r.base + v);
  } else if (v + 16 > r.vbeg && v < r.vend) {
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (uint32_t j = 0; j < 16; ++j) {
      if (v + j >= r.vbeg && v + j < r.vend) {
        w[j >> 2] |= wave::gload_u8(r.base + v + j) << (8 * (j & 3));
      }
    }
    x.x = w[0];
    x.y = w[1];
    x.z = w[2];
    x.w = w[3];
  }
  const uint32_t idx = v & (kInRing - 1);
  *(wave::u32x4*)(r.ring + idx) = x;
  if (idx == 0) {
    *(wave::u32x4*)(r.ring + kInRing) = x; /* mirror: 4-byte reads may run past the end */
  }
}

/* Make [q, want_hi) resident as far as the ring allows, keeping everything from
 * q's block on. q must be >= the oldest position still needed. */
__device__ __forceinline__ void in_ensure(InRing& r, uint32_t q, uint32_t want_hi)
{
  const uint32_t qb = q & ~(kInBlock - 1);
  if (qb >= r.hi || qb < r.lo) { /* nothing useful resident: restart at q's block */
    r.lo = qb;
    r.hi = qb;
  } else {
    r.lo = qb;
  }
  bool loaded = false;
  while (r.hi < want_hi && r.hi < r.vend && r.hi - r.lo < kInRing) {
    in_load_block(r, r.hi);
    r.hi += kInBlock;
    loaded = true;
  }
  if (loaded) {
    wave::sync();
  }
}

template <class R>
__device__ __forceinline__ bool in_resident(const R& r, uint32_t v_lo, uint32_t v_hi)
{
  return v_lo >= r.lo && v_hi <= r.hi;
}

/* Byte at virtual position v (per lane): LDS when resident, HBM otherwise. The stream type R is lzw::InRing (a wave's
 * ring over the chunk) or lzt::Stream (common/lz_team.hip.h: the whole chunk staged in LDS, kMask = all ones). */
template <class R>
__device__ __forceinline__ uint32_t in_byte(const R& r, uint32_t v)
{
  if (v >= r.lo && v < r.hi) {
    return r.ring[v & R::kMask];
  }
  return wave::gload_u8(r.base + v);
}

/* Same for a wave-uniform position; the result is uniform. */
template <class R>
__device__ __forceinline__ uint32_t in_byte_uniform(const R& r, uint32_t v)
{
  return wave::uniform(in_byte(r, v));
}

/* ---- token chase by pointer doubling ------------------------------------------
 *
 * The format-specific part is a functor `delta(r, p)`: the distance from a (speculative)
 * token at virtual position p to the token after it, or kUnknownDelta when that cannot be
 * told from the resident bytes. For a window of 256 positions the wave builds, in LDS,
 * byte tables J_i[p] = distance from p to the 2^i-th token after p (255 = leaves the
 * window / unknown), i = 0..5, by five doubling rounds J_i[p] = J_{i-1}[p] + J_{i-1}[p +
 * J_{i-1}[p]]. Lane n then finds the n-th token after any start position by following the
 * set bits of n through the tables: 64 token positions for 6 dependent LDS reads, instead of
 * 64 dependent scalar steps. The tables do not depend on the start, so a window is built
 * once however many batches it feeds. */

constexpr uint32_t kUnknownDelta = 1u << 28; /* chunk sizes are < 2^28 */
constexpr uint32_t kNxUnknown = 0xffffu;     /* the same in the 16-bit form the chase keeps (real deltas are < 2^15) */

struct Chase
{
  uint32_t wb;    /* virtual position of window slot 0 */
  uint32_t nx01, nx23; /* lane l: the deltas of positions wb + 4 l + {0, 1} and {2, 3}, 16 bits each (kNxUnknown = unknown) */
  uint32_t q;     /* virtual position of the next token */
  uint8_t* tab;   /* LDS, kChaseLds bytes */
};

__device__ __forceinline__ void chase_init(Chase& c, uint32_t q, uint8_t* lds)
{
  c.q = q;
  c.nx01 = 0;
  c.nx23 = 0;
  c.wb = q - kChaseWin; /* forces a build */
  c.tab = lds;
}

/* Lane l of a build owns the 4 consecutive window positions 4 l .. 4 l + 3: their token
 * bytes come from two dword reads of the ring and their table entries are written as one dword.
 * A window is "interior" when every byte a delta may look at is resident and before the end of
 * the chunk; the format's `fast` delta then needs no bounds tests at all. */
template <class R, class Delta>
__device__ __forceinline__ void chase_build(Chase& c, const R& r, Delta delta, uint32_t positions = kChaseWin)
{
  /* `positions` (<= kChaseWin, uniform): only the first that many window positions count as inside the window -- the
   * workgroup-per-chunk decoder (common/lz_team.hip.h) cuts its windows where 64 lanes are sure to hold every token */
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  c.wb = c.q;
  const uint32_t room = r.vend - c.wb; /* c.q < vend */
  const uint32_t limit = room < positions ? room : positions;
  const uint32_t reach = c.wb + kChaseWin + Delta::kReach;
  const bool interior = c.wb >= r.lo && reach <= r.hi && reach <= r.vend;
  const uint32_t base = c.wb + 4 * lane;
  uint32_t nx[4];
  if (interior) {
    /* 8 stream bytes from `base` on, from three aligned dwords (a misaligned ds_read_b32 costs 16x) */
    const uint32_t a0 = base & ~3u;
    const uint32_t d0 = *(const uint32_t*)(r.ring + (a0 & R::kMask));
    const uint32_t d1 = *(const uint32_t*)(r.ring + ((a0 + 4) & R::kMask));
    const uint32_t d2 = *(const uint32_t*)(r.ring + ((a0 + 8) & R::kMask));
    const uint32_t w0 = wave::align_bytes(d1, d0, base & 3u);
    const uint32_t w1 = wave::align_bytes(d2, d1, base & 3u);
    const uint64_t w = ((uint64_t)w1 << 32) | w0;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
      nx[k] = delta.fast(r, base + k, w >> (8 * k));
    }
    /* what the straight-line form gave up on gets the general one (a branch for the wave: rare on text, every window
     * of a sorted column, whose matches take a second length byte) */
    if (Delta::kSecondChance && wave::ballot((nx[0] | nx[1] | nx[2] | nx[3]) >= kUnknownDelta)) {
#pragma unroll
      for (uint32_t k = 0; k < 4; ++k) {
        if (nx[k] >= kUnknownDelta) {
          nx[k] = delta.second(r, base + k, w >> (8 * k));
        }
      }
    }
  } else {
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
      nx[k] = delta(r, base + k);
    }
  }
  uint32_t a[4];
#pragma unroll
  for (uint32_t k = 0; k < 4; ++k) {
    a[k] = 4 * lane + k + nx[k] < limit ? nx[k] : 255u; /* the successor must be a token inside the window */
    nx[k] = nx[k] >= kNxUnknown ? kNxUnknown : nx[k];
  }
  /* kept for the token through which the chain leaves the window (chase_tokens): two registers, not four */
  c.nx01 = nx[0] | (nx[1] << 16);
  c.nx23 = nx[2] | (nx[3] << 16);
  /* the four distances of a lane travel as two dwords of two 16-bit lanes (positions 0|1 and 2|3): a doubling round
   * is two packed adds + two packed saturations instead of four of each, and one byte permute packs the table word */
  uint32_t a01 = a[0] | (a[1] << 16), a23 = a[2] | (a[3] << 16);
  *(uint32_t*)(c.tab + 4 * lane) = wave::perm_bytes(a23, a01, 0x06040200u);
  wave::sync();
#pragma unroll
  for (uint32_t i = 1; i < kChaseLevels; ++i) {
    const uint8_t* prev = c.tab + (i - 1) * kChaseWin + 4 * lane;
    /* a == 255 reads past its table (into the next one, still inside kChaseLds): the sum saturates anyway */
    const uint32_t g0 = prev[0 + (a01 & 0xffffu)], g1 = prev[1 + (a01 >> 16)];
    const uint32_t g2 = prev[2 + (a23 & 0xffffu)], g3 = prev[3 + (a23 >> 16)];
    a01 = wave::pk_add_sat255(a01, g0 | (g1 << 16)); /* valid sums are <= 254: position + sum < 256 */
    a23 = wave::pk_add_sat255(a23, g2 | (g3 << 16));
    *(uint32_t*)(c.tab + i * kChaseWin + 4 * lane) = wave::perm_bytes(a23, a01, 0x06040200u);
    wave::sync();
  }
}

/* Append token positions to seqpos lanes [k, 64); returns the new count. `slow(r, p)`
 * gives the successor of the token at p when its delta is kUnknownDelta.
 * (A window's ~54 tokens and a batch's 64 never line up: every batch enumerates 2.2 times. Round 6 kept what an
 * enumeration found and the batch had no room for in a register and handed it to the next batch with a shuffle -- 36 vector
 * instructions fewer per batch, and slower: mix 648 -> 646 GB/s, Snappy 481 -> 473, text 596 -> 558 (gpurun r6n); one more
 * live register and two more branches in this loop cost more than the table walk they save. Not kept.) */
template <class R, class Delta, class Slow>
__device__ __forceinline__ uint32_t chase_tokens(
    Chase& c, const R& r, uint32_t& seqpos, uint32_t k, Delta delta, Slow slow)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  while (k < 64 && c.q < r.vend) {
    if (c.q - c.wb >= kChaseWin) {
      if (k >= kChaseEnough) {
        break; /* a batch of what one window held is cheaper than a second build + enumeration for a few more tokens */
      }
      chase_build(c, r, delta);
      LZW_T(1);
    }
    /* Lane k + n: the n-th token from c.q, if it lies in this window. Pure lane arithmetic, no predicate logic (the
     * scalar unit is as busy as the vector unit in this kernel: every && of two lane conditions is a scalar
     * instruction): a lane follows the levels named by the bits of n; a level it does not take adds 0; 255 (the
     * successor leaves the window) poisons the lane through `worst`; the position wraps inside the table. */
    const uint32_t idx = (lane - k) & 63u;
    const uint32_t pos0 = c.q - c.wb;
    /* ranks 32 and up start at the 32nd token: two steps of the 16-token table from the start position (every lane
     * reads the same two bytes); the jump tables themselves stop at 16, one doubling round less per window */
    const uint32_t top = (kChaseLevels - 1) * kChaseWin;
    const uint32_t t1 = c.tab[top + pos0];
    const uint32_t t2 = c.tab[top + ((pos0 + t1) & (kChaseWin - 1))];
    const bool upper = (idx & 32u) != 0;
    uint32_t pos = upper ? (pos0 + t1 + t2) & (kChaseWin - 1) : pos0;
    uint32_t worst = upper ? (t1 > t2 ? t1 : t2) : 0u;
#pragma unroll
    for (uint32_t i = 0; i < kChaseLevels; ++i) {
      const uint32_t a = c.tab[i * kChaseWin + pos];
      const uint32_t adv = a & (uint32_t)(-(int32_t)((idx >> i) & 1u));
      worst = adv > worst ? adv : worst;
      pos = (pos + adv) & (kChaseWin - 1);
    }
    /* every n appears in exactly one lane, and the valid n are a prefix: their number is the ballot's population */
    const uint32_t count = wave::popc64(wave::ballot(worst != 255u));
    const uint32_t room = 64 - k;
    const uint32_t take = count < room ? count : room;
    if (idx < take) {
      seqpos = c.wb + pos;
    }
    if (take < count) {
      c.q = c.wb + wave::read_lane(pos, (k + take) & 63u);
    } else {
      /* the window's chain is used up: leave through the last token's own delta */
      const uint32_t last = wave::read_lane(pos, (k + count - 1) & 63u);
      const uint32_t pair = wave::read_lane((last & 2u) ? c.nx23 : c.nx01, last >> 2);
      const uint32_t d = (last & 1u) ? pair >> 16 : pair & 0xffffu;
      c.q = d == kNxUnknown ? slow(r, c.wb + last) : c.wb + last + d;
    }
    k += take;
    LZW_T(2);
  }
  return k;
}

/* ---- output window --------------------------------------------------------- */

struct OutWindow
{
  uint8_t* win;      /* LDS, kOutLds bytes, 16-byte aligned */
  uint8_t* out;      /* chunk output pointer in HBM (uniform) */
  uint32_t align;    /* out & 15: window index = position - wbase + align */
  uint32_t falign;   /* out & (kFlushAlign - 1): position + falign is address-congruent modulo kFlushAlign */
  uint32_t wbase;    /* output position of window index `align` (multiple of 16) */
  uint32_t valid_lo; /* positions >= valid_lo (and < op) are present in the window */
  uint32_t flushed;  /* positions < flushed are in HBM; valid_lo <= flushed <= op, op - flushed < 16 between batches */
};

__device__ __forceinline__ void out_init(OutWindow& w, uint8_t* out, uint8_t* lds)
{
  w.win = lds;
  w.out = out;
  w.align = (uint32_t)((uintptr_t)out & 15u);
  w.falign = (uint32_t)((uintptr_t)out & (kFlushAlign - 1));
  w.wbase = 0;
  w.valid_lo = 0;
  w.flushed = 0;
}

__device__ __forceinline__ uint8_t* out_at(const OutWindow& w, uint32_t pos)
{
  return w.win + (pos - w.wbase + w.align);
}

/* Slide the window so that the batch's `need` bytes (at most kBatchMax) fit after position op. (Until round 6 it made room for
 * kBatchMax whatever the batch produced, i.e. slid in front of every batch: a batch of text is ~700 bytes, the slide two
 * syncs and a copy -- NVCOMP_LZW_ROOM_EXACT = 0 is that behaviour. What stays behind without a slide is history: a match into
 * it is an LDS copy instead of a load from memory.) */
#ifndef NVCOMP_LZW_LAZY_FLUSH
#define NVCOMP_LZW_LAZY_FLUSH 0 /* A/B: 1 = a batch's blocks are written to HBM when the window slides, not at its end. Measured (gpurun r6ay): LZ4 mix
                                 * 683 -> 686, Snappy mix 501 -> 484, float32 column 412 -> 400: off */
#endif
#ifndef NVCOMP_LZW_KEEP_MAX
#define NVCOMP_LZW_KEEP_MAX 0 /* A/B: history kept by a slide when the batch is small (0: kKeep always) */
#endif
#ifndef NVCOMP_LZW_ROOM_EXACT
#define NVCOMP_LZW_ROOM_EXACT 1
#endif
__device__ __forceinline__ void out_make_room(OutWindow& w, uint32_t op, uint32_t need)
{
  if (op - w.wbase + w.align + (NVCOMP_LZW_ROOM_EXACT ? need : kBatchMax) <= kOutWin) {
    return;
  }
  /* history kept: kKeep bytes, or -- NVCOMP_LZW_KEEP_MAX -- as much as the batch leaves room for, up to that many (one
   * pass of the copy loop below moves up to 1 KiB: more history costs no more instructions) */
  uint32_t keep = kKeep;
  if (NVCOMP_LZW_KEEP_MAX > NVCOMP_LZW_KEEP) {
    const uint32_t fits = kOutWin - 32u - need; /* need <= kBatchMax: at least kKeep + 32 */
    keep = fits < (uint32_t)NVCOMP_LZW_KEEP_MAX ? fits : (uint32_t)NVCOMP_LZW_KEEP_MAX;
    keep = keep > kKeep ? keep : kKeep;
  }
  uint32_t keep_from = op > keep ? op - keep : 0;
  if (keep_from > w.flushed) {
    keep_from = w.flushed; /* the tail that waits for its 16-byte block to fill up always stays (kKeep < 16: no history at all) */
  }
  if (keep_from < w.valid_lo) {
    keep_from = w.valid_lo;
  }
  /* wbase stays a multiple of 16 so that window index and HBM address agree mod 16 */
  const uint32_t new_base = keep_from & ~15u;
  const uint32_t shift = new_base - w.wbase; /* multiple of 16 */
  if (shift == 0) {
    return;
  }
  const uint32_t len = op - new_base + w.align; /* window bytes still needed, from index 0 */
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  for (uint32_t base = 0; base < len; base += 1024) {
    const uint32_t i = base + lane * 16;
    wave::u32x4 t = {0, 0, 0, 0};
    if (i < len) {
      t = *(const wave::u32x4*)(w.win + shift + i);
    }
    wave::sync(); /* every lane has read before any lane overwrites */
    if (i < len) {
      *(wave::u32x4*)(w.win + i) = t;
    }
    wave::sync();
  }
  w.wbase = new_base;
  if (w.valid_lo < new_base) {
    w.valid_lo = new_base;
  }
}

/* Flush window bytes of output positions [from, to) to HBM: byte stores up to
 * the first 16-byte boundary, aligned 16-byte lane stores, byte stores for the tail. */
__device__ __forceinline__ void out_flush_range(const OutWindow& w, uint32_t from, uint32_t to)
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  const uint32_t a_from = from + w.align; /* absolute-address-congruent coordinates */
  const uint32_t a_to = to + w.align;
  uint32_t body_lo = (a_from + 15u) & ~15u;
  uint32_t body_hi = a_to & ~15u;
  if (body_lo > body_hi) { /* everything inside one 16-byte block */
    body_lo = a_to;
    body_hi = a_to;
  }
  uint8_t* gout = w.out - w.align;            /* so that gout + a == out + pos */
  const uint8_t* lwin = w.win - w.wbase;      /* so that lwin + a == win + (pos - wbase + align) */
  if (lane < body_lo - a_from) {
    wave::gstore_u8(gout + a_from + lane, lwin[a_from + lane]);
  }
  for (uint32_t a = body_lo + lane * 16; a < body_hi; a += 1024) {
    wave::gstore_u32x4_aligned(gout + a, *(const wave::u32x4*)(lwin + a));
  }
  if (lane < a_to - body_hi) {
    wave::gstore_u8(gout + body_hi + lane, lwin[body_hi + lane]);
  }
}

/* Per-batch flush: only whole 16-byte blocks go out; the last op % 16 bytes wait in the window for the
 * next batch (sub-dword global stores cost a memory request each). After the first flush `flushed` sits on
 * a 16-byte boundary of the output address, so a batch issues aligned 16-byte stores only. */
__device__ __forceinline__ void out_flush(OutWindow& w, uint32_t op_end)
{
  /* address-congruent coordinate of the end of the last whole block (of kFlushAlign bytes: a multiple of 16) */
  const uint32_t f_to = (op_end + w.falign) & ~(kFlushAlign - 1);
  if (f_to > w.flushed + w.falign) {
    if (!(NVCOMP_LZW_ABLATE_EXEC & 1)) out_flush_range(w, w.flushed, f_to - w.falign);
    w.flushed = f_to - w.falign;
  }
}

/* After a sequence was streamed HBM -> HBM behind the window's back (the callers' `big` path): restart the window at op. */
__device__ __forceinline__ void restart_window(OutWindow& w, uint32_t op)
{
  w.wbase = op & ~15u;
  w.valid_lo = op;
  w.flushed = op;
}

/* Everything up to op_end, tail bytes included (end of the chunk, or before HBM-to-HBM copies). */
__device__ __forceinline__ void out_flush_all(OutWindow& w, uint32_t op_end)
{
  if (op_end > w.flushed) {
    out_flush_range(w, w.flushed, op_end);
    w.flushed = op_end;
  }
}

/* The first `take` of `count` sequences in hand are done: the others move down to lane 0, and the lanes behind them
 * become EMPTY sequences again (execute_window_batch relies on that instead of masking by the count). */
__device__ __forceinline__ void drop_front(lz::Seq& s, uint32_t take, uint32_t count)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  const uint32_t from = (lane + take) & 63u;
  const bool keep = lane < count - take;
  const uint32_t a = wave::shuffle(s.lit_src, from), b = wave::shuffle(s.lit_len, from);
  const uint32_t c = wave::shuffle(s.match_off, from), d = wave::shuffle(s.match_len, from);
  s.lit_src = keep ? a : 0u;
  s.lit_len = keep ? b : 0u;
  s.match_off = keep ? c : 0u;
  s.match_len = keep ? d : 0u;
}

/* ---- two waves per chunk: what the producer and the consumer share (lz4_decode_window.hip.h: pair) ---- */
namespace pair {

constexpr uint32_t kSlotBytes = 16 + 4 * 64 * 4; /* n, flags, pad | lit_src[64] | lit_len[64] | match_off[64] | match_len[64] */
constexpr uint32_t kFlagLast = 1, kFlagBad = 2;
constexpr uint32_t kCtrlBytes = 16; /* state[2], abort, pad */
/* window | consumer ring | producer ring | chase tables | two slots | control */
constexpr uint32_t kLdsPerChunk = lzw::kOutLds + 2 * lzw::kInLds + lzw::kChaseLds + 2 * kSlotBytes + kCtrlBytes;

struct Shared
{
  uint8_t* slots;  /* two of kSlotBytes each: slot(k) -- not an array of two pointers: indexed by a run-time k, that array
                    * lived in scratch memory (40 bytes per lane, a scratch load per hand-over) */
  uint32_t* state; /* [2]: 0 = empty, 1 = full */
  uint32_t* abort; /* the consumer gave up: the producer stops waiting */
  __device__ __forceinline__ uint8_t* slot(uint32_t k) const { return slots + k * kSlotBytes; }
};

__device__ __forceinline__ Shared shared_at(uint8_t* lds)
{
  uint8_t* q = lds + lzw::kOutLds + 2 * lzw::kInLds + lzw::kChaseLds;
  Shared sh;
  sh.slots = q;
  sh.state = (uint32_t*)(q + 2 * kSlotBytes);
  sh.abort = sh.state + 2;
  return sh;
}

/* lane 0's view of a flag word, the same for the whole wave */
__device__ __forceinline__ uint32_t poll(const uint32_t* p)
{
  return wave::read_lane(wave::lds_load_acquire(p), 0);
}

} // namespace pair

/* ---- LDS copies ------------------------------------------------------------ */

__device__ __forceinline__ uint32_t ld32(const uint8_t* p)
{
  return lz::ld_u32(p);
}

/* Number of 4-byte steps (2, 4 or 8) that cover the longest participating run:
 * two ballots instead of a wave-wide max reduction. */
__device__ __forceinline__ uint32_t steps_for(bool participates, uint32_t len)
{
  if (wave::ballot(participates && len > 16)) {
    return 8;
  }
  return wave::ballot(participates && len > 8) ? 4u : 2u;
}

/* Note on alignment: a ds_read_b32 / ds_write_b32 with any misaligned lane serialises the wave instruction
 * on gfx950 (64 cycles instead of 4, scripts/microbench/lds_unaligned.hip). A variant of the copies below
 * that reads aligned dwords, realigns with v_alignbyte_b32 and writes aligned dwords + head/tail bytes
 * halved the LDS busy time but added 48 % VALU instructions and was 13 % slower (profiles/
 * r01_exec_ablation.json): the decoder is bound by instruction issue, not by the LDS pipe, so the
 * misaligned dword moves stay. The token chase, which reads the ring with uniform phase, uses aligned reads. */

/* Per-lane copy of len (4..32) bytes in `steps` dword moves whose offsets are
 * clamped to len-4: the last moves simply repeat the final dword, so there is no
 * tail handling and no per-step predication. Byte-serial semantics hold for
 * overlapping LDS ranges as long as the distance is >= 4. */
template <uint32_t STEPS>
__device__ __forceinline__ void copy_dwords_clamped(uint8_t* dst, const uint8_t* src, uint32_t len)
{
  const uint32_t last = len - 4;
#pragma unroll
  for (uint32_t i = 0; i < STEPS; ++i) {
    const uint32_t o = 4 * i < last ? 4 * i : last;
    lz::st_u32(dst + o, lz::ld_u32(src + o));
  }
}

/* Same, for sources that must all be fetched before anything is written
 * (HBM gathers: one round trip for the whole batch). */
template <uint32_t STEPS>
__device__ __forceinline__ void load_dwords_clamped(uint32_t (&buf)[8], const uint8_t* src, uint32_t len)
{
  const uint32_t last = len - 4;
#pragma unroll
  for (uint32_t i = 0; i < STEPS; ++i) {
    const uint32_t o = 4 * i < last ? 4 * i : last;
    buf[i] = wave::gload_u32(src + o);
  }
}

template <uint32_t STEPS>
__device__ __forceinline__ void store_dwords_clamped(uint8_t* dst, const uint32_t (&buf)[8], uint32_t len)
{
  const uint32_t last = len - 4;
#pragma unroll
  for (uint32_t i = 0; i < STEPS; ++i) {
    const uint32_t o = 4 * i < last ? 4 * i : last;
    lz::st_u32(dst + o, buf[i]);
  }
}

/* Far-match data (registers) into the window with aligned accesses only: dst sits h = 1..4 bytes into the
 * aligned frame that starts at dst - h (h = 4: dst itself is aligned). d[] = the match's sequential dwords,
 * l4 = its last 4 bytes. Up to 3 head bytes, whole aligned dwords, up to 3 tail bytes. */
template <uint32_t STEPS>
__device__ __forceinline__ void far_store_aligned(uint8_t* dst, const uint32_t (&d)[8], uint32_t l4, uint32_t len, bool on)
{
  const uint32_t a = (uint32_t)((uintptr_t)dst & 3u);
  const uint32_t h = a ? a : 4u;
  uint8_t* frame = dst - h;
  const uint32_t end = on ? h + len : 0u; /* frame bytes [h, end) */
  const uint32_t sh = 4u - h;
  const uint32_t v0 = wave::align_bytes(d[0], 0u, sh);
#pragma unroll
  for (uint32_t b = 1; b < 4; ++b) {
    if (b >= h && b < end) {
      frame[b] = (uint8_t)(v0 >> (8 * b));
    }
  }
#pragma unroll
  for (uint32_t j = 1; j <= STEPS; ++j) {
    const uint32_t v = wave::align_bytes(j < STEPS ? d[j < 8 ? j : 7] : 0u, d[j - 1], sh);
    if (4 * j + 4 <= end) {
      *(uint32_t*)(frame + 4 * j) = v;
    }
  }
  const uint32_t tc = end & 3u;
  uint8_t* tp = frame + (end & ~3u);
  const uint32_t tv = l4 >> (8 * ((4u - tc) & 3u));
#pragma unroll
  for (uint32_t b = 0; b < 3; ++b) {
    if (b < tc) {
      tp[b] = (uint8_t)(tv >> (8 * b));
    }
  }
}

/* sequential (unclamped) dwords of a far match plus its last 4 bytes; reads up to 3 bytes past the match */
template <uint32_t STEPS>
__device__ __forceinline__ void far_load_seq(uint32_t (&buf)[8], uint32_t& l4, const uint8_t* src, uint32_t len)
{
#pragma unroll
  for (uint32_t i = 0; i < STEPS; ++i) {
    if (4 * i < len) {
      buf[i] = wave::gload_u32(src + 4 * i);
    }
  }
  l4 = wave::gload_u32(src + len - 4);
}

/* x / off == umulhi(x, kMagic.v[off]) for x < 2^16 and 2 <= off < 256 (the runs' periods): a table instead of the 32-bit
 * division 0xffffffff / off + 1, which was ~40 instructions per long match. */
struct MagicTable
{
  uint32_t v[256];
  constexpr MagicTable() : v()
  {
    for (uint32_t i = 2; i < 256; ++i) {
      v[i] = 0xffffffffu / i + 1;
    }
  }
};
__constant__ static const MagicTable kMagic = MagicTable();

/* Periodic fill: d[i] = d[i - off] for i in [0, len) with off < 256, i.e. the `off` bytes before d repeated. The
 * pattern is read ONCE -- lane l keeps its bytes 4l .. 4l+3 -- and every output dword is assembled from four
 * cross-lane fetches (ds_bpermute) at byte index (position mod off); the dwords go out with ALIGNED stores, 256 bytes
 * per step, a few head / tail bytes around them. No step reads what an earlier step wrote: no round trips through
 * LDS, and when off divides 256 (1, 2, 4, 8, ... : runs and typed columns) the dword of a lane never changes, a step is
 * one store. The pattern-doubling copy this replaces for short periods moved 4 .. 64 bytes per dependent round trip. */
__device__ __forceinline__ void lds_periodic_fill(uint8_t* d, uint32_t off, uint32_t len)
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  /* the pattern, a dword per lane: ONE (misaligned) read by the off / 4 lanes that hold a piece of it -- a misaligned
   * LDS access costs per active lane, and the periods of runs are short; the last lane's dword may reach past d - 1:
   * bytes at or behind `off` are never selected */
  uint32_t pat = 0;
  if (4 * lane < off) {
    pat = ld32(d - off + 4 * lane);
  }
  const uint32_t head = (4u - ((uint32_t)(uintptr_t)d & 3u)) & 3u; /* bytes up to the first aligned dword */
  const uint32_t h = head < len ? head : len;
  const uint32_t magic = kMagic.v[off]; /* x / off == umulhi(x, magic) for x < 2^16 (off 1: x mod 1 = 0 below, magic unused) */
  if (lane < h) {
    const uint32_t sidx = off == 1 ? 0u : lane - __umulhi(lane, magic) * off;
    d[lane] = (uint8_t)(wave::shuffle(pat, sidx >> 2) >> (8 * (sidx & 3u)));
  } else if (h != 0) {
    (void)wave::shuffle(pat, 0); /* the cross-lane fetch is executed by the whole wave */
  }
  const uint32_t body = (len - h) >> 2; /* aligned dwords */
  const uint32_t step = off == 1 ? 0u : 256u - __umulhi(256u, magic) * off; /* 256 mod off: how far the phase moves per 256-byte step */
  uint32_t r = h + 4 * lane;
  r = off == 1 ? 0u : r - __umulhi(r, magic) * off;
  uint32_t word = 0;
  bool fresh = true;
  for (uint32_t base = 0; base < body; base += 64) {
    if (fresh) {
      uint32_t sidx = r;
      word = 0;
#pragma unroll
      for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t b = (wave::shuffle(pat, sidx >> 2) >> (8 * (sidx & 3u))) & 0xffu;
        word |= b << (8 * k);
        sidx = sidx + 1 == off ? 0 : sidx + 1;
      }
      fresh = step != 0;
      r += step;
      r = r >= off ? r - off : r;
    }
    if (base + lane < body) {
      *(uint32_t*)(d + h + 4 * (base + lane)) = word;
    }
  }
  const uint32_t tail_at = h + 4 * body;
  const uint32_t tail = len - tail_at; /* 0 .. 3 */
  {
    const uint32_t i = tail_at + lane;
    const uint32_t sidx = off == 1 ? 0u : i - __umulhi(i, magic) * off;
    const uint32_t b = wave::shuffle(pat, (lane < tail ? sidx : 0u) >> 2) >> (8 * (sidx & 3u));
    if (lane < tail) {
      d[i] = (uint8_t)b;
    }
  }
  wave::sync();
}

/* The same for the periods of runs and typed columns -- off in {1, 2, 4, 8, 16}, len >= 32: every ALIGNED 16-byte
 * block of the run holds the same 16 bytes. The first 32 bytes are written byte by byte (32 lanes, index & (off - 1)),
 * the aligned block inside them is read back by all lanes at once (one address: a broadcast), and the run is a
 * sequence of ds_write_b128 -- 1 KiB per instruction, no cross-lane fetches, no division. A sorted-key column compressed
 * by liblz4 (the reference's published shape) is almost only such matches: 400 bytes at offset 8. */
__device__ __forceinline__ void lds_pow2_fill(uint8_t* d, uint32_t off, uint32_t len)
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  if (lane < 32) {
    d[lane] = d[(int32_t)(lane & (off - 1)) - (int32_t)off];
  }
  wave::sync();
  uint8_t* a = d + ((16u - ((uint32_t)(uintptr_t)d & 15u)) & 15u); /* first aligned block: inside d[0, 32) */
  const wave::u32x4 q = *(const wave::u32x4*)a;
  const uint32_t nblk = (uint32_t)(d + len - a) >> 4; /* whole blocks from a on */
  for (uint32_t b = lane; b < nblk; b += 64) {
    *(wave::u32x4*)(a + 16 * b) = q;
  }
  uint8_t* t = a + 16 * nblk;
  if (t + lane < d + len) { /* the last partial block: its bytes are the block's first ones */
    t[lane] = a[lane];
  }
  wave::sync();
}

/* Match copy inside the window with byte-serial semantics: d[i] = d[i - off].
 * Periods below 256 are a periodic fill; longer ones move 256 bytes per step (a step never reads what it writes). */
__device__ __forceinline__ void lds_match_copy(uint8_t* d, uint32_t off, uint32_t len)
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  if (off <= 16 && (off & (off - 1)) == 0 && len >= 32) {
    lds_pow2_fill(d, off, len);
    return;
  }
  if (off < 256) {
    lds_periodic_fill(d, off, len);
    return;
  }
  uint32_t done = 0;
  while (done < len) {
    const uint32_t rem = len - done;
    if (rem >= 256) {
      uint8_t* t = d + done + lane * 4;
      lz::st_u32(t, ld32(t - off));
      done += 256;
    } else {
      for (uint32_t i = lane; i < rem; i += 64) {
        uint8_t* t = d + done + i;
        *t = *(t - off);
      }
      done = len;
    }
    wave::sync();
  }
}

/* dst (LDS) <- src (HBM), non-overlapping, whole wave, any alignment. */
__device__ __forceinline__ void copy_to_lds(uint8_t* dst, const uint8_t* src, uint32_t len)
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  uint32_t base = 0;
  for (; base + 256 <= len; base += 256) {
    lz::st_u32(dst + base + lane * 4, wave::gload_u32(src + base + lane * 4));
  }
  for (uint32_t i = base + lane; i < len; i += 64) {
    dst[i] = (uint8_t)wave::gload_u8(src + i);
  }
}

/* ---- long runs: straight to HBM ----------------------------------------------------------------------------------
 *
 * A sequence with a long literal run or a long match does not go through the window: the window's bytes are flushed,
 * the sequence is written to the output buffer directly with 16-byte lane accesses (1 KiB per wave instruction), and the
 * window restarts behind it (it holds no history). Data of this kind is what CAN approach the HBM roofline
 * (incompressible chunks are one literal run, runs and sorted columns are a few long matches), and through the window
 * it paid an LDS round trip, a per-batch cut at kBatchMax bytes and, for matches, a dependent load -> store chain per
 * step. The reference's only published number is of this shape (doc/Benchmarks.md:88-95).
 */
#ifndef NVCOMP_LZW_STREAM_LIT
#define NVCOMP_LZW_STREAM_LIT 4096 /* literal runs from this length on are streamed */
#endif
#ifndef NVCOMP_LZW_STREAM_MATCH
#define NVCOMP_LZW_STREAM_MATCH 4096 /* matches from this length on are streamed */
#endif
constexpr uint32_t kStreamLit = NVCOMP_LZW_STREAM_LIT;
constexpr uint32_t kStreamMatch = NVCOMP_LZW_STREAM_MATCH;

/* dst[0, n) = src[0, n): different buffers or src at least 8 KiB in front of dst; any alignment. The stores are 16-byte
 * aligned and non-temporal, four loads of 1 KiB are in flight before the first store (one load -> store round trip per KiB
 * made a 64 KiB literal run latency-bound at 2 TB/s of copy traffic), and the next four are issued BEFORE the stores of
 * the four in hand. scripts/probes/copy_bench.hip (16 384 runs of 64 KiB, the decoders' launch shape): load-then-store
 * 2 380 GB/s -- the very rate the decoder had on incompressible chunks --, with the loads ahead 2 500, with non-temporal
 * stores as well 2 615; hipMemcpyAsync device-to-device 2 499. */
__device__ __forceinline__ void stream_copy(uint8_t* dst, const uint8_t* src, uint32_t n)
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  const uint32_t head = (16u - (uint32_t)((uintptr_t)dst & 15u)) & 15u;
  const uint32_t h = head < n ? head : n;
  if (lane < h) {
    wave::gstore_u8(dst + lane, wave::gload_u8(src + lane));
  }
  const uint32_t body_end = h + ((n - h) & ~15u); /* the 16-byte blocks end here */
  uint32_t base = h;
  if (base + 4096 <= body_end) {
    uint32_t at = base + 16 * lane;
    wave::u32x4 a = wave::gload_u32x4(src + at), b = wave::gload_u32x4(src + at + 1024);
    wave::u32x4 c = wave::gload_u32x4(src + at + 2048), d = wave::gload_u32x4(src + at + 3072);
    for (base += 4096; base + 4096 <= body_end; base += 4096) {
      const uint32_t nx = base + 16 * lane;
      const wave::u32x4 a2 = wave::gload_u32x4(src + nx), b2 = wave::gload_u32x4(src + nx + 1024);
      const wave::u32x4 c2 = wave::gload_u32x4(src + nx + 2048), d2 = wave::gload_u32x4(src + nx + 3072);
      wave::gstore_u32x4_aligned_nt(dst + at, a);
      wave::gstore_u32x4_aligned_nt(dst + at + 1024, b);
      wave::gstore_u32x4_aligned_nt(dst + at + 2048, c);
      wave::gstore_u32x4_aligned_nt(dst + at + 3072, d);
      a = a2, b = b2, c = c2, d = d2;
      at = nx;
    }
    wave::gstore_u32x4_aligned_nt(dst + at, a);
    wave::gstore_u32x4_aligned_nt(dst + at + 1024, b);
    wave::gstore_u32x4_aligned_nt(dst + at + 2048, c);
    wave::gstore_u32x4_aligned_nt(dst + at + 3072, d);
  }
  for (; base < body_end; base += 1024) {
    const uint32_t at = base + 16 * lane;
    if (at < body_end) {
      wave::gstore_u32x4_aligned(dst + at, wave::gload_u32x4(src + at));
    }
  }
  const uint32_t tail = body_end + lane;
  if (tail < n) {
    wave::gstore_u8(dst + tail, wave::gload_u8(src + tail));
  }
}

/* d[i] = d[i - off] for i in [0, len), off < 256, the off bytes in front of d already in HBM: the pattern is read once
 * (a dword per lane) and every 16-byte store is assembled in registers from cross-lane fetches at (position mod off);
 * when off divides 1 KiB (1, 2, 4, 8, ...: runs, typed columns) a lane's 16 bytes never change and the run is a sequence
 * of plain stores -- no load at all behind the first. */
__device__ __forceinline__ void stream_periodic(uint8_t* d, uint32_t off, uint32_t len)
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  uint32_t pat = 0;
  if (4 * lane < off) {
    pat = wave::gload_u32(d - off + 4 * lane); /* bytes at or behind `off` are never selected */
  }
  const uint32_t magic = kMagic.v[off]; /* x / off == umulhi(x, magic) for x < 2^16 */
  const uint32_t head = (16u - (uint32_t)((uintptr_t)d & 15u)) & 15u;
  const uint32_t h = head < len ? head : len;
  {
    const uint32_t sidx = off == 1 ? 0u : lane - __umulhi(lane, magic) * off; /* lane < 64 <= 2^16 */
    const uint32_t b = wave::shuffle(pat, (lane < h ? sidx : 0u) >> 2) >> (8 * (sidx & 3u));
    if (lane < h) {
      wave::gstore_u8(d + lane, b);
    }
  }
  const uint32_t blocks = (len - h) >> 4; /* aligned 16-byte blocks */
  const uint32_t step = off == 1 ? 0u : 1024u - __umulhi(1024u, magic) * off; /* 1024 mod off: the phase shift per KiB */
  uint32_t r = h + 16 * lane;
  r = off == 1 ? 0u : r - __umulhi(r, magic) * off;
  wave::u32x4 q = {0, 0, 0, 0};
  bool fresh = true;
  for (uint32_t base = 0; base < blocks; base += 64) {
    if (fresh) {
      uint32_t sidx = r;
      uint32_t w[4];
#pragma unroll
      for (uint32_t k = 0; k < 4; ++k) {
        uint32_t word = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
          const uint32_t b = (wave::shuffle(pat, sidx >> 2) >> (8 * (sidx & 3u))) & 0xffu;
          word |= b << (8 * j);
          sidx = sidx + 1 == off ? 0 : sidx + 1;
        }
        w[k] = word;
      }
      q.x = w[0], q.y = w[1], q.z = w[2], q.w = w[3];
      fresh = step != 0;
      r += step;
      r = r >= off ? r - off : r;
    }
    if (base + lane < blocks) {
      wave::gstore_u32x4_aligned(d + h + 16 * (base + lane), q);
    }
  }
  const uint32_t tail_at = h + 16 * blocks;
  {
    const uint32_t i = tail_at + lane;
    const uint32_t x = i % off; /* once per run: i may be as large as the chunk */
    const uint32_t b = wave::shuffle(pat, (i < len ? x : 0u) >> 2) >> (8 * (x & 3u));
    if (i < len) {
      wave::gstore_u8(d + i, b);
    }
  }
}

/* d[i] = d[i - off], i in [0, len), byte-serial semantics, everything in front of d already in HBM (one wave's vector
 * memory operations are served in issue order: a later load sees an earlier store of the same wave). */
__device__ __forceinline__ void stream_match(uint8_t* d, uint32_t off, uint32_t len)
{
  if (off < 256) {
    stream_periodic(d, off, len);
    return;
  }
  /* The effective offset E stays a multiple of off and doubles as the run grows (E <= done + off: the source never
   * starts in front of d - off), so a step of up to E bytes reads nothing the same step writes: 4 KiB with four loads
   * in flight once E allows it, 1 KiB or 256 bytes before, single bytes for what is left. */
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  uint32_t done = 0;
  uint32_t E = off;
  while (done < len) {
    while (E < 4096 && 2 * E <= done + off) {
      E *= 2;
    }
    const uint32_t rem = len - done;
    if (E >= 4096 && rem >= 4096) {
      uint8_t* t = d + done + 16 * lane;
      const wave::u32x4 a = wave::gload_u32x4(t - E), b = wave::gload_u32x4(t - E + 1024);
      const wave::u32x4 c = wave::gload_u32x4(t - E + 2048), e = wave::gload_u32x4(t - E + 3072);
      wave::gstore_u32x4(t, a);
      wave::gstore_u32x4(t + 1024, b);
      wave::gstore_u32x4(t + 2048, c);
      wave::gstore_u32x4(t + 3072, e);
      done += 4096;
    } else if (E >= 1024 && rem >= 1024) {
      uint8_t* t = d + done + 16 * lane;
      wave::gstore_u32x4(t, wave::gload_u32x4(t - E));
      done += 1024;
    } else if (rem >= 256) { /* E >= off >= 256 */
      uint8_t* t = d + done + 4 * lane;
      wave::gstore_u32(t, wave::gload_u32(t - E));
      done += 256;
    } else {
      for (uint32_t i = lane; i < rem; i += 64) { /* rem < 256 <= E: no byte of this step reads another */
        uint8_t* t = d + done + i;
        wave::gstore_u8(t, wave::gload_u8(t - E));
      }
      done = len;
    }
    wave::sync();
  }
}

/* A sequence that is streamed instead of executed in the window (execute_window_batch reported `big` for the first
 * sequence in hand): literals [lsrc, lsrc + llen) of the stream, then the match (moff, mlen). Validates like the batch
 * executor; returns false on error. The window is flushed first and restarts empty behind the sequence. */
template <bool CHECKED>
__device__ __forceinline__ bool stream_sequence(
    const InRing& ir, OutWindow& ow, uint32_t out_cap, uint32_t& op, uint32_t lsrc, uint32_t llen, uint32_t moff, uint32_t mlen,
    uint32_t& err)
{
  /* tested whether or not the caller asked for statuses (once per streamed sequence: nothing): an unchecked decode of a
   * corrupt stream must not turn a claimed length of megabytes into an HBM-to-HBM copy past the output slot */
  {
    const uint64_t end = (uint64_t)op + llen + mlen;
    if (end > out_cap || (mlen != 0 && (moff == 0 || moff > op + llen))) {
      err |= end > out_cap ? lz::kErrOutput : lz::kErrOffset;
      return false;
    }
  }
  out_flush_all(ow, op); /* the copies below read what the window still held back */
  wave::sync();
  if (llen) {
    stream_copy(ow.out + op, ir.base + lsrc, llen);
    wave::sync();
  }
  if (mlen) {
    stream_match(ow.out + op + llen, moff, mlen);
  }
  op += llen + mlen;
  restart_window(ow, op);
  return true;
}

/* (Round 6 measured a path of its own for "a few literals, then a long match of period 1 .. 16" -- the sequences of sorted
 * key columns --: the whole wave executed such a sequence alone, sources in LDS, the run's aligned block stored straight to HBM,
 * no batch around it. ~150 instructions a sequence against the ~100 the batch executor spends on it amortised: sorted-key
 * column (HC) 2 166 -> 1 720 GB/s, int32 column 1 476 -> 1 262, gpurun r6l. Gone; what these chunks need is several runs in
 * flight at once, not a shorter path for one.) */
/* A match copied by the whole wave into the window at output position hw (everything below hw is final): its source
 * may start in front of what the window holds -- those bytes are in HBM (flushed before the window let go of them). */
__device__ __forceinline__ void coop_match(OutWindow& ow, uint32_t hw, uint32_t foff, uint32_t flen)
{
  const uint32_t fsrc = hw - foff;
  if (fsrc >= ow.valid_lo) {
    lds_match_copy(out_at(ow, hw), foff, flen);
  } else {
    const uint32_t n_hbm = fsrc + flen <= ow.valid_lo ? flen : ow.valid_lo - fsrc;
    copy_to_lds(out_at(ow, hw), ow.out + fsrc, n_hbm);
    wave::sync();
    if (n_hbm < flen) {
      lds_match_copy(out_at(ow, hw + n_hbm), foff, flen - n_hbm);
    }
  }
}

/* ---- runs: sequences that are "a few literals, then a match of period 1, 2, 4, 8 or 16" ---------------------------------
 *
 * Sorted key columns, typed columns, runs -- the reference's only published LZ4 number is of this shape
 * (doc/Benchmarks.md:88-95) -- decode to a few hundred such sequences a chunk, 200-400 bytes each. The batch executor
 * below takes five of them per batch (2 KiB of window) and copies every match with the whole wave, one after the other:
 * ~65 vector and ~80 scalar instructions a sequence, 0.28 of the memory roofline with neither the memory system nor the
 * vector units busy. Here up to 64 of them are executed at once and never pass through the window:
 *
 *   pattern   With a period that divides 16, every ALIGNED 16-byte block of a run holds the same 16 bytes Q (byte i = the
 *             byte of every output address congruent to i). Q of sequence k is Q of sequence k - 1 with the literals of k
 *             written over it at their address residues: a "last writer" scan across the lanes
 *             (wave::scan_last_writer), seeded with the 16 bytes in front of the batch from the window.
 *   joints    The blocks that are not all run -- the end of run k - 1, the literals of k, the head of run k: at most two
 *             blocks a sequence with up to 16 literals -- are assembled in a 32-byte slot per lane (the window's LDS, which
 *             nothing else needs meanwhile) and stored by their lane.
 *   bodies    The whole blocks of a run are stores of Q, 1 KiB per instruction, a short scalar loop per run.
 *
 * Sequences without literals behind a match of the same period continue its run (liblz4's fast compressor cuts a run in
 * two, Snappy's copies end at 64 bytes): they are merged into it. The window restarts behind the batch with the last 16
 * bytes as its history. Returns the number of sequences executed; 0 = not this kind of batch, nothing touched: the
 * caller goes on with execute_window_batch (which also reports what is wrong with a corrupt stream).
 */
#ifndef NVCOMP_LZW_RUNS
#define NVCOMP_LZW_RUNS 1 /* A/B: 0 = every batch through execute_window_batch (rounds 1-5) */
#endif
#ifndef NVCOMP_LZW_RUN_NT
#define NVCOMP_LZW_RUN_NT 0 /* A/B: the sweep's stores non-temporal */
#endif
#ifndef NVCOMP_LZW_RUN_MIN
#define NVCOMP_LZW_RUN_MIN 4 /* fewer leading sequences of the shape (and fewer than 256 bytes): not worth the fixed cost of this path */
#endif
constexpr uint32_t kRunMin = NVCOMP_LZW_RUN_MIN;
#ifndef NVCOMP_LZW_RUN_MIN_BYTES
#define NVCOMP_LZW_RUN_MIN_BYTES 2048 /* a batch of runs smaller than this (a batch of the window): not worth it -- the path's
                                        * fixed cost is ~450 instructions, the window's batch of 2 KiB costs ~1 000 all in (measured:
                                        * Snappy's int32 column, whose 64 elements a refill are ~9 runs or 2-4 KB, lost 10 % here) */
#endif
constexpr uint32_t kRunMinBytes = NVCOMP_LZW_RUN_MIN_BYTES;

/* When the decoders try execute_run_batch (the loops that hold it: lz4w::decode_chunk<., ., true>). A batch of the window
 * that was cut by its size with fewer than 24 sequences -- long sequences -- turns the attempts on for the batches behind
 * it. An attempt that finds the first sequences not of the shape at all (liblz4's fast compressor on the sorted-key column:
 * every other sequence is a 6-byte match far back) spaces the next ones out, 2, 4, 8, 16 batches of the window between them;
 * an attempt declined for a short run among the first sequences or a small batch does not (the int32 column: one run in
 * twenty is shorter than 16 bytes; 1 500 GB/s with those counted, 1 830 without: gpurun r6v, r6w). */
#ifndef NVCOMP_LZW_RUN_GATE
#define NVCOMP_LZW_RUN_GATE 1 /* A/B: 0 = every batch is tried */
#endif
/* One uniform word (as a struct of three the compiler kept it in scratch memory): bit 0 = the last batch of the window took
 * few, long sequences (set at the start: the loops that hold the path get the chunks that shrank 8 x, their first batch is
 * tried); bits 4-6 = attempts in a row that found other shapes; bits 8-15 = batches of the window to let pass before the
 * next attempt. */
typedef uint32_t RunGate;
constexpr RunGate kRunGateInit = 1u;
__device__ __forceinline__ bool run_gate_open(RunGate g)
{
  return NVCOMP_LZW_RUNS && (!NVCOMP_LZW_RUN_GATE || (g & 0xff01u) == 1u);
}
__device__ __forceinline__ RunGate run_gate_tried(RunGate g, uint32_t take, bool misfit)
{
  if (take) {
    return g & ~0x70u;
  }
  if (!misfit) {
    return g;
  }
  const uint32_t m = (g >> 4) & 7u, m1 = m < 4 ? m + 1 : 4u;
  return (g & 1u) | (m1 << 4) | ((1u << m1) << 8);
}
__device__ __forceinline__ RunGate run_gate_window_took(RunGate g, uint32_t take, uint32_t count)
{
  const uint32_t skip = g >> 8;
  return (g & 0x70u) | (take < 24u && take < count ? 1u : 0u) | ((skip ? skip - 1u : 0u) << 8);
}

constexpr uint32_t kRunMax = 60; /* sequences a batch: 32 bytes of joint a lane + the table of the sweep = the window's LDS */
constexpr bool kRunFits = kRunMax * 32 + 64 * 4 <= kOutLds; /* its slots and table live in the window's LDS (the callers assert it) */

/* mask of the low n bytes of a dword, n clamped to 0 .. 4 */
__device__ __forceinline__ uint32_t low_bytes_mask(int32_t n)
{
  const uint32_t c = n < 0 ? 0u : n > 4 ? 4u : (uint32_t)n;
  return c == 4 ? ~0u : (1u << (8u * c)) - 1u;
}

/* x as a ring of 16 bytes, rotated so that byte j moves to byte (j + sft) & 15 */
__device__ __forceinline__ void rotate16(uint32_t (&x)[4], uint32_t sft)
{
  const uint32_t bs = sft & 3u, ws = sft >> 2;
  uint32_t t[4];
#pragma unroll
  for (uint32_t d = 0; d < 4; ++d) {
    t[d] = bs ? wave::align_bytes(x[d], x[(d + 3) & 3], 4u - bs) : x[d];
  }
#pragma unroll
  for (uint32_t d = 0; d < 4; ++d) {
    x[d] = ws == 0 ? t[d] : ws == 1 ? t[(d + 3) & 3] : ws == 2 ? t[(d + 2) & 3] : t[(d + 1) & 3];
  }
}

/* the last `off` bytes of the 16-byte value x (off = 1, 2, 4, 8, 16: uniform), repeated over all 16 */
__device__ __forceinline__ void repeat_tail16(uint32_t (&x)[4], uint32_t off)
{
  if (off == 8) {
    x[0] = x[2], x[1] = x[3];
  } else if (off <= 4) {
    uint32_t v = x[3];
    if (off == 2) {
      v = (v & 0xffff0000u) | (v >> 16);
    } else if (off == 1) {
      v >>= 24;
      v |= v << 8;
      v |= v << 16;
    }
    x[0] = v, x[1] = v, x[2] = v, x[3] = v;
  }
}

template <bool CHECKED>
__device__ __forceinline__ uint32_t execute_run_batch(
    const InRing& ir, OutWindow& ow, uint32_t out_cap, uint32_t& op, uint32_t n, const lz::Seq& s, bool& misfit)
{
  misfit = true; /* declined because the first sequences are not of this shape at all (RunGate spaces the attempts out) */
  /* uniform tests on the first sequence and the window: what a batch of any other kind of data pays for this path */
  uint32_t off = wave::read_lane(s.match_off, 0);
  if ((off - 1u >= 16u || (off & (off - 1u)) != 0) && n > 1) {
    /* the second sequence's instead: Snappy's literals are an element of their own in front of the copy; liblz4's fast
     * compressor starts a run of equal keys with a short match from far back (below: speculated) */
    off = wave::read_lane(s.match_off, 1);
  }
  if (off - 1u >= 16u || (off & (off - 1u)) != 0) {
    LZ_STAT("run_decl_period", 1);
    return 0;
  }
  /* the joints are whole aligned blocks: everything in front of the window's tail block must be in HBM (not so at the
   * start of a chunk whose buffer is not 16-byte aligned: the first batch goes through the window) */
  if (((ow.flushed + ow.align) & 15u) != 0 || op - ow.flushed >= 16u) {
    LZ_STAT("run_decl_window", 1);
    return 0;
  }
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  /* the leading sequences of this shape: the period of the first one, at most 16 literals (resident in the ring); a
   * sequence may be literals only (Snappy's literal elements; the run is then what follows it) or nothing at all (the
   * followers of a merged copy train) */
  /* ... and a match whose distance is a MULTIPLE of the period is taken for what it mostly is on such data -- the same
   * bytes the run would have produced there (equal keys: the new key's unchanged bytes copied from an earlier key) -- and
   * checked against the patterns further down */
  const bool weak = lane < n && (s.match_len == 0 || ((s.match_off & (off - 1u)) == 0 && s.match_off != 0 && s.match_len < kStreamMatch))
                    && (s.lit_len == 0 || (s.lit_len <= 16u && s.lit_src - ir.lo + s.lit_len <= ir.hi - ir.lo));
  const uint64_t not_weak = ~wave::ballot(weak) | (1ull << kRunMax);
  const uint32_t r0 = wave::ctz64(not_weak);
  if (r0 < kRunMin) {
    LZ_STAT("run_decl_lead", 1);
    return 0;
  }
  misfit = false; /* what declines from here on is this batch's bad luck: a short run among the first, a small batch */
  const bool in0 = lane < r0;
  const uint32_t len = in0 ? s.lit_len + s.match_len : 0u;
  const uint32_t incl = wave::scan_add_inclusive(len);
  /* a sequence without literals goes on with the run in front of it: heads are the others; a head's run ends where the
   * next head's literals start */
  const bool head = in0 && (lane == 0 || s.lit_len != 0);
  uint64_t heads = wave::ballot(head);
  const uint64_t hx = heads | (1ull << r0);
  const uint64_t above = lane < 63 ? hx >> (lane + 1) : 0ull;
  const uint32_t next_head = above ? lane + 1 + wave::ctz64(above) : r0;
  const uint32_t L = op + incl - len;      /* where the sequence's literals go */
  const uint32_t M = L + s.lit_len;        /* where its match starts */
  uint32_t E = 0; /* where the sequence's (merged) run ends */
  /* How many of the r1 leading sequences a batch can be: its last run must be 16 bytes long -- the window restarts with its
   * pattern as the last 16 bytes of the output. (Runs in between may be of any length: the joints of sequences that meet in
   * one block share it, below. The first versions ended a batch at a run shorter than 16 bytes, then at one without a block
   * boundary in it: int32 column 1 750-1 810, then 2 370-2 570 GB/s.) */
  auto settle = [&](uint32_t r1) -> uint32_t {
    const uint32_t nh = next_head < r1 ? next_head : r1;
    E = op + wave::shuffle(incl, (nh - 1u) & 63u);
    uint32_t r = r1;
    for (uint32_t i = 0; i < 4 && r >= kRunMin; ++i) {
      const uint32_t last_head = 63u - (uint32_t)__builtin_clzll(heads & ((1ull << r) - 1ull)); /* lane 0 is a head */
      if (wave::read_lane(E - M, last_head) >= 16u) {
        return r;
      }
      r = last_head;
    }
    return 0;
  };
  uint32_t R = settle(r0);
  if (R < kRunMin) {
    LZ_STAT("run_decl_short", 1);
    return 0;
  }
  heads &= (1ull << R) - 1ull;
  uint32_t total = wave::read_lane(incl, R - 1);
  {
    /* the batch's end is the only output test; the first match must have its period in front of it (every other one
     * starts behind it), in the window if the literals do not cover it */
    const uint32_t m0 = op + wave::read_lane(s.lit_len, 0);
    if (op + total > out_cap || m0 < off || (m0 - op < off && m0 - off < ow.valid_lo)) {
      LZ_STAT("run_decl_first", 1);
    return 0;
    }
    /* short runs and small batches are as well off in the window */
    if (total < 64u * wave::popc64(heads) || total < kRunMinBytes) {
      LZ_STAT("run_decl_small", 1);
    return 0;
    }
  }
  LZW_T(4);
  LZ_STAT("run_batches", 1);
  LZ_STAT("run_seqs", R);
  LZ_STAT("run_heads", wave::popc64(heads));
  LZ_STAT("run_bytes", total);

  /* the 16 bytes in front of the batch by address residue: the window's tail block up to op, the block before behind it */
  uint32_t qi[4];
  {
    const uint32_t cop = op + ow.align; /* "coordinates": position + align, congruent to the address modulo 16; window index = coordinate - wbase */
    const uint32_t fl = cop & ~15u, t = cop & 15u;
    const wave::u32x4 w1 = *(const wave::u32x4*)(ow.win + (fl - ow.wbase));
    wave::u32x4 w0 = {0, 0, 0, 0};
    if (fl - ow.wbase >= 16u) {
      w0 = *(const wave::u32x4*)(ow.win + (fl - 16u - ow.wbase));
    }
    const uint32_t a1[4] = {w1.x, w1.y, w1.z, w1.w}, a0[4] = {w0.x, w0.y, w0.z, w0.w};
#pragma unroll
    for (uint32_t d = 0; d < 4; ++d) {
      const uint32_t m = low_bytes_mask((int32_t)t - (int32_t)(4 * d));
      qi[d] = (a1[d] & m) | (a0[d] & ~m);
    }
  }
  /* a lane's contribution: its last 16 literal bytes (those it has: the others masked out), their last `off` repeated,
   * rotated to the address residues of the bytes in front of its match */
  uint32_t q[4], qm[4];
  {
    const uint32_t e = s.lit_src + s.lit_len;
#pragma unroll
    for (uint32_t d = 0; d < 4; ++d) {
      q[d] = ld32(ir.ring + ((e - 16u + 4u * d) & (kInRing - 1)));
      qm[d] = ~low_bytes_mask(16 - (int32_t)(4 * d) - (int32_t)s.lit_len);
    }
    repeat_tail16(q, off);
    repeat_tail16(qm, off);
    const uint32_t sft = (M + ow.align) & 15u;
    rotate16(q, sft);
    rotate16(qm, sft);
    if (lane == 0) {
      /* the first sequence inherits from the output in front of the batch: its last `off` bytes, repeated like a run */
      uint32_t h[4] = {qi[0], qi[1], qi[2], qi[3]};
      const uint32_t t = (op + ow.align) & 15u;
      rotate16(h, (16u - t) & 15u); /* byte j = position op - 16 + j */
      repeat_tail16(h, off);
      rotate16(h, t);
#pragma unroll
      for (uint32_t d = 0; d < 4; ++d) {
        q[d] = (q[d] & qm[d]) | (h[d] & ~qm[d]);
        qm[d] = ~0u;
      }
    }
#pragma unroll
    for (uint32_t d = 0; d < 4; ++d) {
      wave::scan_last_writer(q[d], qm[d]);
    }
  }
  /* ---- the speculated matches (distance a multiple of the period, not the period): each one's source bytes must be what
   * the run's pattern gives at its destination. The source lies in a run of this batch (then its bytes are that run's
   * pattern: compare the two patterns at the residues the match covers) or, up to 16 bytes, in the output in front of the
   * batch (one 16-byte load). A match that fails, or whose source is anywhere else, ends the batch in front of it: every
   * pattern below it is right whatever comes behind (the scan runs upwards), and nothing has been written yet. ---- */
  {
    const bool spec = lane < R && s.match_len != 0 && s.match_off != off;
    if (wave::ballot(spec)) {
      const uint32_t sp = M - s.match_off; /* wraps for an offset beyond the output: fails every test below */
      const uint32_t m = s.match_len;
      const uint32_t sft = (M + ow.align) & 15u;
      /* (a) in this batch: the sequence t whose output holds sp (the L are ascending; lanes behind the batch hold its end) */
      uint32_t t = 0;
#pragma unroll
      for (uint32_t step = 32; step != 0; step >>= 1) {
        const uint32_t c = t + step;
        t = wave::shuffle(L, c) <= sp ? c : t;
      }
      const uint32_t mt = wave::shuffle(M, t), et = wave::shuffle(E, t);
      bool ok = spec && sp >= op && sp + off >= mt && sp + m <= et;
      {
        uint32_t cover[4];
#pragma unroll
        for (uint32_t d = 0; d < 4; ++d) {
          cover[d] = low_bytes_mask((int32_t)m - (int32_t)(4 * d)); /* the first min(m, 16) bytes ... */
        }
        rotate16(cover, sft); /* ... at the residues of the match's destination */
        uint32_t diff = 0;
#pragma unroll
        for (uint32_t d = 0; d < 4; ++d) {
          diff |= (wave::shuffle(q[d], t) ^ q[d]) & cover[d];
        }
        ok = ok && diff == 0;
      }
      /* (b) in HBM in front of the batch, 16 bytes at most, the load inside the chunk's buffer */
      const bool hbm = spec && !ok && m <= 16u && sp <= ow.flushed && m <= ow.flushed - sp && out_cap >= 16u && sp <= out_cap - 16u;
      if (wave::ballot(hbm)) {
        uint32_t want[4] = {q[0], q[1], q[2], q[3]};
        rotate16(want, (16u - sft) & 15u); /* byte i = the pattern's byte at the destination's position M + i */
        wave::u32x4 got = {0, 0, 0, 0};
        if (hbm) {
          got = wave::gload_u32x4(ow.out + sp);
        }
        const uint32_t g[4] = {got.x, got.y, got.z, got.w};
        uint32_t diff = 0;
#pragma unroll
        for (uint32_t d = 0; d < 4; ++d) {
          diff |= (g[d] ^ want[d]) & low_bytes_mask((int32_t)m - (int32_t)(4 * d));
        }
        ok = ok || (hbm && diff == 0);
      }
      const uint64_t failed = wave::ballot(spec && !ok);
      LZ_STAT("run_spec", wave::popc64(wave::ballot(spec)));
      LZ_STAT("run_spec_failed", wave::popc64(failed));
      if (failed) {
        /* the batch ends in front of the first one; its last run may have become short, the batch small */
        R = settle(wave::ctz64(failed));
        if (R < kRunMin) {
          misfit = true;
          LZ_STAT("run_decl_spec", 1);
          return 0;
        }
        heads &= (1ull << R) - 1ull;
        total = wave::read_lane(incl, R - 1);
        if (total < 64u * wave::popc64(heads) || total < kRunMinBytes) {
          misfit = true;
          LZ_STAT("run_decl_spec", 1);
          return 0;
        }
      }
    }
  }
  LZW_T(16); /* runs: the patterns */
  /* ---- joints: the blocks that are not all one run, assembled on a canvas in LDS (the window's, which nothing else needs
   * meanwhile). A head's joint are the blocks from its first literal to the first whole block of its run, two at most;
   * when its run is too short to reach the next block boundary the next head's joint starts in the same block, and the
   * two (or more) share that block of the canvas. Every byte of the canvas has one writer -- the head whose literals or
   * whose run it belongs to (a run's bytes: its pattern, masked) -- so the writes are ORs into a zeroed canvas and plain
   * stores of the literals. ---- */
  uint8_t* const gbase = ow.out - ow.align; /* coordinate 0 */
  const bool jlane = wave::lane_in(heads);
  {
    const uint32_t cl = L + ow.align, cm = M + ow.align; /* coordinates of the literals and of the run */
    const uint32_t a = cl & 15u;                         /* where the literals start in the joint's first block */
    const uint32_t b0 = cl >> 4, b1 = (cm + 15u) >> 4;   /* the joint's blocks [b0, b1) */
    const uint32_t nj = jlane ? b1 - b0 : 0u;
    /* a head whose first block is the last block of the joint below shares it (the b1 are ascending: the running maximum is
     * the nearest one below) */
    const uint32_t below_b1 = wave::prev_lane(wave::scan_max_inclusive(nj ? b1 : 0u));
    const uint32_t shared = nj != 0 && below_b1 == b0 + 1u ? 1u : 0u;
    /* The common batch (the sorted-key column: every run hundreds of bytes) has no shared block and no run that ends inside
     * its own joint: every head then owns the two blocks at 2 x its lane, writes them whole -- the pattern below up to its
     * literals, its own pattern behind -- and its literals over them: a sixth of the instructions of the general case. */
    const bool plain = wave::ballot(jlane && (shared != 0 || ((E + ow.align) >> 4) < b1)) == 0;
    const uint32_t cv = plain ? 2u * lane : wave::scan_add_inclusive(nj - shared) - nj; /* the joint's first block on the canvas */
    uint32_t* const canvas = (uint32_t*)ow.win;
    wave::sync(); /* the window's blocks have been read */
    if (plain) {
      wave::u32x4 w0, w1;
      uint32_t x[4];
#pragma unroll
      for (uint32_t d = 0; d < 4; ++d) {
        const uint32_t below = wave::prev_lane(q[d]);
        const uint32_t m = low_bytes_mask((int32_t)a - (int32_t)(4 * d));
        x[d] = ((lane == 0 ? qi[d] : below) & m) | (q[d] & ~m);
      }
      w0.x = x[0], w0.y = x[1], w0.z = x[2], w0.w = x[3];
      w1.x = q[0], w1.y = q[1], w1.z = q[2], w1.w = q[3];
      if (lane < kRunMax) {
        *(wave::u32x4*)(ow.win + 32u * lane) = w0;
        *(wave::u32x4*)(ow.win + 32u * lane + 16) = w1;
      }
    } else {
      const wave::u32x4 zero = {0, 0, 0, 0};
      if (lane < kRunMax) {
        *(wave::u32x4*)(ow.win + 32u * lane) = zero;
        *(wave::u32x4*)(ow.win + 32u * lane + 16) = zero;
      }
    }
    wave::sync();
    const uint32_t next_cv = plain ? 0u : wave::shuffle(cv, next_head & 63u); /* where the next head's joint starts */
    if (jlane && !plain) {
      /* my run's bytes inside my own joint: from the run's start to the joint's end or the run's, whichever is first */
      const uint32_t run_lo = a + s.lit_len;          /* relative to my first block: 0 .. 31 */
      const uint32_t run_hi = E + ow.align - 16u * b0; /* where my run ends, relative to my first block */
      const uint32_t hi = run_hi < 16u * nj ? run_hi : 16u * nj;
#pragma unroll
      for (uint32_t d = 0; d < 8; ++d) {
        const uint32_t m = low_bytes_mask((int32_t)hi - (int32_t)(4 * d)) & ~low_bytes_mask((int32_t)run_lo - (int32_t)(4 * d));
        if (m != 0 && 4 * d < 16u * nj) {
          wave::lds_or(canvas + 4u * cv + d, q[d & 3] & m);
        }
      }
      /* ... and in front of the next head's literals, when those start in a block of their own */
      const uint32_t ce = E + ow.align;
      if (next_head < R && (ce >> 4) >= b1 && (ce & 15u) != 0) {
#pragma unroll
        for (uint32_t d = 0; d < 4; ++d) {
          const uint32_t m = low_bytes_mask((int32_t)(ce & 15u) - (int32_t)(4 * d));
          if (m != 0) {
            wave::lds_or(canvas + 4u * next_cv + d, q[d] & m);
          }
        }
      }
      /* the first joint's first block starts with the output in front of the batch (the window's tail block) */
      if (lane == 0 && nj != 0) {
#pragma unroll
        for (uint32_t d = 0; d < 4; ++d) {
          const uint32_t m = low_bytes_mask((int32_t)a - (int32_t)(4 * d));
          if (m != 0) {
            wave::lds_or(canvas + 4u * cv + d, qi[d] & m);
          }
        }
      }
    }
    if (jlane) {
      /* the literals */
      uint8_t* at = ow.win + 16u * cv + a;
      if (s.lit_len >= 4) {
        const uint32_t last = s.lit_len - 4;
        uint32_t data[4];
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) {
          const uint32_t o = 4 * i < last ? 4 * i : last;
          data[i] = ld32(ir.ring + ((s.lit_src + o) & (kInRing - 1)));
        }
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) {
          const uint32_t o = 4 * i < last ? 4 * i : last;
          lz::st_u32(at + o, data[i]);
        }
      } else if (s.lit_len != 0) {
        at[0] = ir.ring[s.lit_src & (kInRing - 1)];
        if (s.lit_len > 1) {
          at[1] = ir.ring[(s.lit_src + 1) & (kInRing - 1)];
        }
        if (s.lit_len > 2) {
          at[2] = ir.ring[(s.lit_src + 2) & (kInRing - 1)];
        }
      }
    }
    /* the table of the sweep: one word a head, in address order -- its first block's number in the batch (14 bits), how many
     * of its blocks are joint blocks (2), where they are on the canvas (7), its lane (6: whose pattern the blocks behind
     * the joint hold); unused entries compare above every block. Heads that start in one block have equal block numbers:
     * the search below takes the last of them, whose canvas block is the shared one and whose run goes on behind it. */
    uint32_t* tab = (uint32_t*)(ow.win + 32u * kRunMax);
    tab[lane] = ~0u;
    wave::sync();
    if (jlane) {
      const uint32_t blk = b0 - ((op + ow.align) >> 4);
      tab[wave::prefix_popc(heads)] = (blk << 15) | (nj << 13) | (cv << 6) | lane;
    }
    wave::sync();
  }
  LZW_T(17); /* runs: the joints */
  /* ---- the sweep: every whole block of the batch, in address order, 1 KiB a store instruction. A lane finds the head its
   * block belongs to (binary search of the table) and takes the block from the canvas or that head's pattern from its
   * lane. (Stores run by run -- a run's ~25 blocks an instruction, the joints scattered -- left the memory pipe the bound:
   * 2 550 GB/s on the sorted-key column, two thirds of the time in that loop: gpurun r6q, r6r.) ---- */
  {
    const uint32_t* tab = (const uint32_t*)(ow.win + 32u * kRunMax);
    const uint32_t c0 = (op + ow.align) & ~15u;
    const uint32_t nblk = (((op + total + ow.align) & ~15u) - c0) >> 4;
    for (uint32_t base = 0; base < nblk; base += 64) {
      const uint32_t bi = base + lane;
      uint32_t lo = 0;
#pragma unroll
      for (uint32_t step = 32; step != 0; step >>= 1) {
        const uint32_t c = lo + step;
        lo = (tab[c] >> 15) <= bi ? c : lo;
      }
      const uint32_t key = tab[lo];
      const uint32_t k = key & 63u, jb = bi - (key >> 15);
      const bool joint = jb < ((key >> 13) & 3u);
      wave::u32x4 x;
      x.x = wave::shuffle(q[0], k), x.y = wave::shuffle(q[1], k), x.z = wave::shuffle(q[2], k), x.w = wave::shuffle(q[3], k);
      if (joint) {
        x = *(const wave::u32x4*)(ow.win + 16u * (((key >> 6) & 127u) + jb));
      }
      if (bi < nblk) {
#if NVCOMP_LZW_RUN_NT
        wave::gstore_u32x4_aligned_nt(gbase + c0 + 16u * bi, x);
#else
        wave::gstore_u32x4_aligned(gbase + c0 + 16u * bi, x);
#endif
      }
    }
  }
  LZW_T(18); /* runs: the bodies */
  /* ---- the window restarts behind the batch: the last run's pattern is its history, its last partial block the tail
   * that waits for the next flush ---- */
  {
    wave::u32x4 ql;
    ql.x = wave::read_lane(q[0], R - 1), ql.y = wave::read_lane(q[1], R - 1);
    ql.z = wave::read_lane(q[2], R - 1), ql.w = wave::read_lane(q[3], R - 1);
    wave::sync();
    op += total;
    const uint32_t nbase = (op - 16u) & ~15u;
    const uint32_t fl = (op + ow.align) & ~15u;
    const uint32_t i0 = fl - 16u - nbase; /* 0 or 16 */
    if (lane == 0) {
      *(wave::u32x4*)(ow.win + i0) = ql;
      *(wave::u32x4*)(ow.win + i0 + 16) = ql;
    }
    wave::sync(); /* and later far reads of this wave see the stores above */
    ow.wbase = nbase;
    ow.valid_lo = op - 16u;
    ow.flushed = fl - ow.align;
  }
  LZW_T(19); /* runs: the window restart */
  return R;
}

/*
 * Execute the first sequences of a parsed batch inside the window. Lane k owns
 * sequence k (k < n, n >= 1); lanes >= n hold EMPTY sequences (lit_len = match_len = 0:
 * the callers keep it so), lit positions are virtual positions in the input ring's
 * coordinate system. Consumes as many leading sequences as fit kBatchMax output
 * bytes (at least one unless the first one alone is larger: then `big` is set
 * and nothing is consumed). Returns the number of sequences consumed and adds
 * the bytes produced to op.
 *
 * Written for the scalar unit as much as for the vector unit (they are equally busy in this kernel): every
 * && / || of two lane conditions is a scalar instruction on 64-bit masks, so ranges are tested with one unsigned
 * compare ((x - lo) <= hi - lo), conditions that grow with the lane are tested once for the wave, and nothing is
 * masked that the callers' invariant already zeroes.
 */
struct NoHook
{
  __device__ __forceinline__ void operator()() const {}
};

/* `after_far`: called once per batch behind the point where the far matches' loads have been waited for (the decoders
 * with a token index settle their prefetched positions there: common/lz_index.hip.h). */
/* LAZY_FLUSH: the batch's whole blocks are not written to HBM at its end but when the window has to slide (and at the
 * chunk's end, and in front of a streamed sequence: out_flush_all) -- every third or fourth batch of text, 2 KiB at a time.
 * Nothing needs them earlier: a match into bytes the window still holds is copied in LDS, and a far match reads below
 * `flushed`. (The run executor wants the window's tail block to be the only unflushed one: its loops stay eager.) */
template <bool CHECKED, bool RING_LITERALS = false, class AfterFar = NoHook, bool LAZY_FLUSH = false>
__device__ __forceinline__ uint32_t execute_window_batch(
    InRing& ir, OutWindow& ow, uint32_t out_cap, uint32_t& op, uint32_t n, const lz::Seq& s, uint32_t& err, bool& big,
    AfterFar after_far = AfterFar())
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  const uint32_t len = s.lit_len + s.match_len;
  const uint32_t incl = wave::scan_add_inclusive(len);
  big = false;

  /* leading sequences whose cumulative output fits one batch (a length is < 2^31: the first sum above kBatchMax has
   * not wrapped, whatever the sums behind it do) */
  /* ... and a sequence with a long literal run or a long match is not for the window at all (stream_sequence) */
  const bool streamed = !RING_LITERALS && (s.lit_len >= kStreamLit || s.match_len >= kStreamMatch);
  const uint64_t over = wave::ballot(incl > kBatchMax || streamed);
  uint32_t take = over ? wave::ctz64(over) : 64u;
  take = take < n ? take : n;
  if (take == 0) {
    big = true;
    return 0;
  }
  const bool mine = lane < take;
  const uint32_t total = wave::read_lane(incl, take - 1);
  const uint32_t lit_dst = op + incl - len;
  const uint32_t match_dst = lit_dst + s.lit_len;
  const uint32_t my_lit = mine ? s.lit_len : 0;
  const uint32_t my_match = mine ? s.match_len : 0;

  if (CHECKED) {
    /* the sums grow with the lane: the batch's end is the only output test (op <= out_cap <= 2^26, total <= kBatchMax);
     * offset 0 wraps to the largest value, so one compare covers "0 or beyond the produced output" */
    const bool bad_out = op + total > out_cap;
    const uint64_t any_off = wave::ballot(my_match != 0 && s.match_off - 1 >= match_dst);
    if (bad_out || any_off) {
      err |= (bad_out ? lz::kErrOutput : 0u) | (any_off ? lz::kErrOffset : 0u);
      return 0;
    }
  }

  LZ_STAT("batches", 1);
  LZ_STAT("seqs", take);
  LZ_STAT("bytes", total);
  LZW_T(4);
  if (LAZY_FLUSH && op - ow.wbase + ow.align + total > kOutWin) {
    out_flush(ow, op); /* the window slides: what it lets go of must be in HBM */
    wave::sync();
  }
  out_make_room(ow, op, total);
  LZW_T(5);

  /* ---- far matches: sources the window no longer holds, read from HBM ---- */
  const uint32_t match_src = match_dst - s.match_off;
  /* DEFLATE (RING_LITERALS) has matches of THREE bytes -- a sixth of all matches of a zlib stream, and as whole-wave
   * copies, one after the other, they were 18 % of that decoder's time (phase clock, profiles/archive/r03_deflate_phases.json) */
  constexpr uint32_t kMinShort = RING_LITERALS ? 3 : 4;
  const bool short_match = my_match - kMinShort <= kMatchShort - kMinShort; /* kMinShort .. kMatchShort */
  const bool is_near = my_match != 0 && match_src >= ow.valid_lo;
  /* the data comes with one or two 16-byte loads, which must stay inside the chunk's buffer (a source in the last 32
   * bytes of the buffer, or a wrapped one of a corrupt unchecked stream, takes the cooperative path below) */
  const bool far_lane = short_match && match_src + my_match <= ow.flushed && out_cap >= 32 && match_src <= out_cap - 32;
  uint32_t far_l4 = 0;
  uint32_t far_data[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const uint32_t far_steps = steps_for(far_lane, my_match);
  if (far_lane && !(NVCOMP_LZW_ABLATE_EXEC & 2)) {
#ifdef NVCOMP_LZW_FAR_ABLATE /* profiling builds only (wrong output): far reads folded onto the chunk's first KiB */
    const uint8_t* src = ow.out + (match_src & 1023u);
#else
    const uint8_t* src = ow.out + match_src;
#endif
    /* A scattered load costs the CU's address unit a slot per lane whatever its width (profiles/archive/r02_decode_phases.json:
     * issuing 3-9 dword loads per batch was 11 % of the wave's time): 16 bytes per load, two loads at most, plus the
     * match's last dword. */
    const wave::u32x4 f0 = wave::gload_u32x4(src);
    far_data[0] = f0.x, far_data[1] = f0.y, far_data[2] = f0.z, far_data[3] = f0.w;
    if (my_match > 16) {
      const wave::u32x4 f1 = wave::gload_u32x4(src + 16);
      far_data[4] = f1.x, far_data[5] = f1.y, far_data[6] = f1.z, far_data[7] = f1.w;
    }
    /* the match's last four bytes; of a three-byte match: a byte that is never used, then its three */
#if NVCOMP_LZW_FAR_L4_FROM_F0
    /* ... of a match of up to 16 bytes they are among the 16 bytes in hand: no third load for it */
    if (my_match > 16) {
      far_l4 = wave::gload_u32(src + my_match - 4);
    } else {
      const uint32_t d = my_match >= 4 ? my_match - 4 : 0u, k = d >> 2;
      const uint32_t lo = k == 0 ? f0.x : k == 1 ? f0.y : k == 2 ? f0.z : f0.w;
      const uint32_t hi = k == 0 ? f0.y : k == 1 ? f0.z : f0.w;
      far_l4 = RING_LITERALS && my_match < 4 ? f0.x << 8 : wave::align_bytes(hi, lo, d & 3u);
    }
#else
    far_l4 = RING_LITERALS && my_match < 4 ? f0.x << 8 : wave::gload_u32(src + my_match - 4);
#endif
  }

  LZW_T(11); /* far classification + load issue */
  /* ---- literals ---- */
  if (!(NVCOMP_LZW_ABLATE_EXEC & 4)) {
    uint8_t* dst = out_at(ow, lit_dst);
    /* a run of 1 .. kLitShort bytes that the ring holds is copied by its own lane (a position in front of the resident
     * range wraps to a huge value: one compare) */
    const bool lit_own = my_lit - 1 < kLitShort && s.lit_src - ir.lo + my_lit <= ir.hi - ir.lo;
    const bool lit_lane = lit_own && my_lit >= 4;
    /* the ring wraps at kInRing; its 16-byte mirror covers a dword that starts before the end,
     * and a run crossing the end is split by the modulo per step */
    const uint32_t lit_steps = steps_for(lit_lane, my_lit);
    if (lit_lane) {
      const uint32_t last = my_lit - 4;
      uint32_t data[8];
#pragma unroll
      for (uint32_t i = 0; i < 8; ++i) {
        if (i < lit_steps) {
          const uint32_t o = 4 * i < last ? 4 * i : last;
          data[i] = ld32(ir.ring + ((s.lit_src + o) & (kInRing - 1)));
        }
      }
#pragma unroll
      for (uint32_t i = 0; i < 8; ++i) {
        if (i < lit_steps) {
          const uint32_t o = 4 * i < last ? 4 * i : last;
          lz::st_u32(dst + o, data[i]);
        }
      }
    }
    LZW_T(6); /* literal runs of 4..32 bytes */
    /* 1..3 bytes: byte-wise (as ONE misaligned dword store, whose excess bytes the match behind it would overwrite, they
     * measured 5 % slower on the headline: a misaligned LDS store costs a cycle or two per ACTIVE lane, and 30 % of the
     * sequences have such a run, 10 % one of four bytes or more) */
    const bool lit_tiny = lit_own && my_lit < 4;
    if (wave::ballot(lit_tiny)) {
      if (lit_tiny) {
        dst[0] = ir.ring[s.lit_src & (kInRing - 1)];
        if (my_lit > 1) {
          dst[1] = ir.ring[(s.lit_src + 1) & (kInRing - 1)];
        }
        if (my_lit > 2) {
          dst[2] = ir.ring[(s.lit_src + 2) & (kInRing - 1)];
        }
      }
    }
    LZW_T(12); /* literal runs of 1..3 bytes */
    uint64_t pending = wave::ballot(my_lit != 0 && !lit_own);
    LZ_STAT("lit_lanes", wave::popc64(wave::ballot(lit_own)));
    LZ_STAT("lit_coop", wave::popc64(pending));
    while (pending) {
      const uint32_t j = wave::ctz64(pending);
      pending &= pending - 1;
      const uint32_t jsrc = wave::read_lane(s.lit_src, j);
      const uint32_t jlen = wave::read_lane(my_lit, j);
      const uint32_t jdst = wave::read_lane(lit_dst, j);
      if (!RING_LITERALS) {
        copy_to_lds(out_at(ow, jdst), ir.base + jsrc, jlen);
      } else {
        /* a ring without a stream behind it (deflate/deflate_decode.hip.h: the literals were decoded, not copied out
         * of the chunk): the run is resident by construction */
        uint8_t* d = out_at(ow, jdst);
        for (uint32_t i = lane; i < jlen; i += 64) {
          d[i] = ir.ring[(jsrc + i) & (kInRing - 1)];
        }
      }
    }
  }

  LZW_T(13); /* long literal runs, whole wave */
  /* ---- far match data into the window ---- */
  if (wave::ballot(far_lane) && !(NVCOMP_LZW_ABLATE_EXEC & 16)) {
    uint8_t* dst = out_at(ow, far_lane ? match_dst : ow.wbase);
    if (far_steps == 2) {
      far_store_aligned<2>(dst, far_data, far_l4, my_match, far_lane);
    } else if (far_steps == 4) {
      far_store_aligned<4>(dst, far_data, far_l4, my_match, far_lane);
    } else {
      far_store_aligned<8>(dst, far_data, far_l4, my_match, far_lane);
    }
  }
  wave::sync();
  after_far();
  LZW_T(7);

  /* ---- remaining matches, oldest first: multi-round resolution in LDS ---- */
  {
    const bool near_lane = is_near && short_match && s.match_off >= 4;
    uint64_t pending = (NVCOMP_LZW_ABLATE_EXEC & 8) ? 0ull : wave::ballot(my_match != 0 && !far_lane);
    const uint64_t near_mask = wave::ballot(near_lane);
    LZ_STAT("match_far_lanes", wave::popc64(wave::ballot(far_lane)));
    LZ_STAT("match_near_lanes", wave::popc64(near_mask));
    LZ_STAT("match_coop", wave::popc64(pending & ~near_mask));
    LZ_STAT("match_coop_below_4", wave::popc64(wave::ballot(my_match != 0 && my_match < 4)));
    LZ_STAT("match_coop_long", wave::popc64(wave::ballot(my_match > kMatchShort)));
    LZ_STAT("match_coop_short_period", wave::popc64(wave::ballot(is_near && short_match && s.match_off < 4)));
    while (pending) {
      const uint32_t f = wave::ctz64(pending);
      const uint32_t hw = wave::read_lane(match_dst, f); /* every byte below hw is final */
      if (!((near_mask >> f) & 1)) {
        /* the oldest pending match is long, has a period < 4, or reaches behind the window */
        const uint32_t foff = wave::read_lane(s.match_off, f);
        const uint32_t flen = wave::read_lane(my_match, f);
        coop_match(ow, hw, foff, flen);
        pending &= ~(1ull << f);
        LZW_T(14); /* matches copied by the whole wave */
        continue;
      }
      const bool ready = near_lane && wave::lane_in(pending) && (lane == f || match_src + my_match <= hw);
      const uint32_t steps = steps_for(ready, my_match);
      LZ_STAT("mrr_rounds", 1);
      LZ_STAT("mrr_iters", steps);
      if (ready) {
        const uint8_t* src = out_at(ow, match_src);
        uint8_t* dst = out_at(ow, match_dst);
        if (RING_LITERALS && my_match < 4) { /* three bytes, offset >= 4 */
          const uint8_t b0 = src[0], b1 = src[1], b2 = src[2];
          dst[0] = b0, dst[1] = b1, dst[2] = b2;
        } else if (steps == 2) {
          copy_dwords_clamped<2>(dst, src, my_match);
        } else if (steps == 4) {
          copy_dwords_clamped<4>(dst, src, my_match);
        } else {
          copy_dwords_clamped<8>(dst, src, my_match);
        }
      }
      wave::sync();
      pending &= ~wave::ballot(ready);
    }
  }

  LZW_T(8);
  if (!LAZY_FLUSH) {
    out_flush(ow, op + total);
  }
  LZW_T(9);
  wave::sync(); /* later far reads of this wave must see the flushed bytes */
  op += total;
  return take;
}

} // namespace lzw
