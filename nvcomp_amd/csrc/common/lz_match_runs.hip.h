/*
 * common/lz_match_runs.hip.h -- the LZ compressors' path for runs: sorted key columns, typed columns, zeros.
 *
 * A chunk of such data is, byte for byte, its own copy from p bytes back (p = the element's width: 1, 2, 4 or 8) except
 * at the few places where a value changes. The match finders (common/lz_match_wide.hip.h) find exactly these matches,
 * position by position, 256 positions a step with a hash table probe each: 195 GB/s on an int32 column, 355 on a sorted
 * 8-byte key column (round 5), where the decoder reads them at 3 TB/s. Here the wave looks at 1 KiB of input per step:
 * every lane compares its 16 bytes with the 16 bytes p back (two loads, a funnel shift, a zero-byte mask), and only
 * the MISMATCHES -- two or three a KiB -- cost scalar work: a mismatch ends the stretch of equal bytes in front of it,
 * and a stretch of four bytes or more is a match of offset p (the same sequences liblz4's HC parser finds on these
 * columns; the decoder's run executor, common/lz_window.hip.h, takes them 60 at a time).
 *
 * Used for a chunk whose first KiB is such a copy to 15 parts in 16; given up (the caller's match finder starts over)
 * when the output grows beyond a quarter of the input consumed. Format-independent: sequences go out through the
 * format's Emitter (lz4/lz4_encode.hip.h, snappy/snappy_encode.hip.h), whose end-of-block rules the caller passes in.
 */
#pragma once

#include "common/lz_common.hip.h"

namespace lzm {
namespace runs {

constexpr uint32_t kNotRuns = ~0u; /* encode_chunk: this chunk is for the match finder */
constexpr uint32_t kMinChunk = 4096;
constexpr uint32_t kMinMatch = 4;

#ifndef NVCOMP_LZM_RUNS
#define NVCOMP_LZM_RUNS 1 /* A/B: 0 = every chunk through the match finder (rounds 1-5) */
#endif

/* 16 bytes at src + at, bytes at or behind n read as zero (never fetched) */
__device__ __forceinline__ wave::u32x4 load16_guarded(const uint8_t* __restrict__ src, uint32_t n, uint32_t at)
{
  wave::u32x4 v = {0, 0, 0, 0};
  if (at + 16 <= n) {
    v = wave::gload_u32x4(src + at);
  } else if (at < n) {
    uint32_t w[4] = {0, 0, 0, 0};
    for (uint32_t j = 0; at + j < n; ++j) {
      w[j >> 2] |= wave::gload_u8(src + at + j) << (8 * (j & 3));
    }
    v.x = w[0], v.y = w[1], v.z = w[2], v.w = w[3];
  }
  return v;
}

/* bit i set: byte i of the 16 bytes `cur` differs from the byte p back (`prv` = the 16 bytes in front of cur); p = 1, 2, 4, 8 */
__device__ __forceinline__ uint32_t mismatch16(wave::u32x4 prv, wave::u32x4 cur, uint32_t p)
{
  const uint32_t w[8] = {prv.x, prv.y, prv.z, prv.w, cur.x, cur.y, cur.z, cur.w};
  uint32_t bits = 0;
#pragma unroll
  for (uint32_t d = 0; d < 4; ++d) {
    uint32_t back;
    if (p == 8) {
      back = w[d + 2];
    } else if (p == 4) {
      back = w[d + 3];
    } else if (p == 2) {
      back = wave::align_bytes(w[d + 4], w[d + 3], 2);
    } else {
      back = wave::align_bytes(w[d + 4], w[d + 3], 3);
    }
    const uint32_t x = w[d + 4] ^ back;
    /* 0x80 in every byte of x that is not zero (exact: no carries between the bytes) */
    const uint32_t t = ((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x;
    /* the four flags as a nibble: the partial products of the multiplication land on distinct bits */
    const uint32_t nib = (((t >> 7) & 0x01010101u) * 0x10204080u) >> 28;
    bits |= nib << (4 * d);
  }
  return bits;
}

constexpr uint32_t kListCap = 1024;               /* mismatch positions collected before they are turned into sequences */
constexpr uint32_t kListBytes = 4 * (kListCap + 1); /* LDS the caller lends (its hash table's) */
constexpr uint32_t kFlushAt = 192;
constexpr uint32_t kLaneLiterals = 64;            /* literal runs up to here are written by the sequence's own lane */

/* Compress src[0, n) as runs of period 1, 2, 4 or 8 with the calling wave. last_start / match_end: the format's
 * end-of-block rules (the last match starts at or before last_start and ends at or before match_end). `list`: kListBytes of
 * LDS. Returns the compressed size, or kNotRuns when the chunk is not of this kind (dst may have been written to).
 *
 * The positions where a byte differs from its p-th predecessor are collected, a KiB of input a step, into a list in LDS;
 * every ~200 of them the list becomes sequences, 64 at a time, a lane a mismatch: the stretch of equal bytes in front of
 * it is a match if it is four bytes long (and the format's end rules allow it), its literals start where the last match
 * below ended (a running maximum across the lanes), its place in the output is a prefix sum of the sizes, and the lane
 * writes it whole (Emitter::emit_lane). (The first version walked the mismatches with the scalar unit and emitted every
 * sequence with the whole wave: 1 120 GB/s on the sorted-key column, 700 on the int32 column -- the walk and the
 * emissions, one after the other, were the chunk's time.) */
/* (NOT inlined: inside the match finders' kernels -- 128 registers, some spilled already -- its registers made the
 * allocator spill 17 instead of 6 in the match finder's loop: the mix 185 -> 168 GB/s, gpurun r6ao. A function of its own has
 * an allocation of its own.) */
template <class Emitter>
__device__ __attribute__((noinline)) uint32_t encode_chunk(
    const uint8_t* __restrict__ src, uint32_t n, uint8_t* dst, uint32_t* list, uint32_t last_start, uint32_t match_end, bool any_match)
{
  static_assert(!Emitter::kStream, "byte-aligned formats only");
  if (!NVCOMP_LZM_RUNS || !any_match || n < kMinChunk) {
    return kNotRuns;
  }
  const uint32_t lane = (uint32_t)wave::lane_id();
  /* ---- which period, if any: the KiB behind the first 16 bytes against itself 1, 2, 4 and 8 bytes back ---- */
  uint32_t p = 0;
  {
    const wave::u32x4 prv = wave::gload_u32x4(src + 16 * lane), cur = wave::gload_u32x4(src + 16 + 16 * lane);
    uint32_t best = 64; /* mismatches allowed in 1 024 bytes: one part in 16 */
#pragma unroll
    for (uint32_t q = 1; q <= 8; q <<= 1) {
      const uint32_t bad = wave::reduce_add((uint32_t)__builtin_popcount(mismatch16(prv, cur, q)));
      if (bad < best) { /* (on a tie the shorter period stays: a KiB of one int32 value is its own copy from 8 back too) */
        best = bad;
        p = q;
      }
    }
  }
  if (p == 0) {
    return kNotRuns;
  }
  uint32_t op = 0;
  uint32_t lit_start = 0; /* everything in front of it is written */
  uint32_t pending = 0;   /* mismatch positions in list[1 ..]; list[0] = the last one in front of them (+ 1 = where the open stretch starts) */
  if (lane == 0) {
    list[0] = ~0u; /* the open stretch starts at 0 (the first p positions are mismatches: it ends at once) */
  }
  /* the list's first `count` positions become sequences */
  auto flush = [&](uint32_t count) -> bool {
    for (uint32_t g = 0; g < count; g += 64) {
      const uint32_t i = g + lane;
      const bool valid = i < count;
      const uint32_t q = valid ? list[1 + i] : 0u;
      const uint32_t run_start = valid ? list[i] + 1u : 0u;
      const uint32_t end = q < match_end ? q : match_end;
      const bool emit = valid && run_start <= last_start && end >= run_start + kMinMatch;
      /* the literals start where the last match below ended (the ends ascend with the lane) */
      const uint32_t ends = wave::scan_max_inclusive(emit ? end : 0u);
      const uint32_t below = wave::prev_lane(ends);
      const uint32_t ls = below > lit_start ? below : lit_start;
      const uint32_t lit_len = run_start - ls, mlen = end - run_start;
      const uint32_t size = emit ? Emitter::seq_size(lit_len, mlen, p) : 0u;
      const uint32_t incl = wave::scan_add_inclusive(size);
      const bool by_wave = emit && lit_len > kLaneLiterals; /* a long literal run: the whole wave copies it */
      if (emit && !by_wave) {
        Emitter::emit_lane(dst + op + incl - size, src + ls, lit_len, p, mlen);
      }
      for (uint64_t big = wave::ballot(by_wave); big; big &= big - 1) {
        const uint32_t j = wave::ctz64(big);
        Emitter::match(dst + op + wave::read_lane(incl - size, j), src + wave::read_lane(ls, j), wave::read_lane(lit_len, j), p,
                       wave::read_lane(mlen, j));
      }
      op += wave::read_lane(incl, 63);
      const uint32_t top = wave::read_lane(ends, 63);
      lit_start = top > lit_start ? top : lit_start;
    }
    return true;
  };
  /* a lane's 16 bytes of the step at `at` and the 16 in front of them (which end inside the chunk: at < n) */
  auto fetch = [&](uint32_t at, wave::u32x4& cur, wave::u32x4& prv) {
    cur = load16_guarded(src, n, at);
    prv = wave::u32x4{0, 0, 0, 0};
    if (at >= 16 && at < n) {
      prv = wave::gload_u32x4(src + at - 16);
    }
  };
  wave::u32x4 ncur, nprv; /* the NEXT step's bytes are requested before this step's mismatches are looked at */
  fetch(16 * lane, ncur, nprv);
  for (uint32_t base = 0; base < n; base += 1024) {
    const uint32_t at = base + 16 * lane;
    const wave::u32x4 cur = ncur, prv = nprv;
    if (base + 1024 < n) {
      fetch(at + 1024, ncur, nprv);
    }
    uint32_t mis = mismatch16(prv, cur, p);
    /* the positions without a predecessor end a stretch; those behind the chunk are not positions */
    if (at < p) {
      mis |= (1u << (p - at)) - 1u;
    }
    if (at + 16 > n) {
      mis &= at >= n ? 0u : ~(0xffffu << (n - at));
    }
    const uint32_t cnt = (uint32_t)__builtin_popcount(mis);
    const uint32_t incl = wave::scan_add_inclusive(cnt);
    const uint32_t total = wave::read_lane(incl, 63);
    if (pending + total > kListCap) {
      return kNotRuns; /* more than a thousand mismatches in two or three KiB: not runs */
    }
    uint32_t slot = 1 + pending + incl - cnt;
    for (uint32_t m = mis; wave::ballot(m != 0); m &= m - 1) {
      if (m != 0) {
        list[slot++] = at + (uint32_t)__builtin_ctz(m);
      }
    }
    pending += total;
    wave::sync_wave();
    if (pending >= kFlushAt) {
      if (!flush(pending)) {
        return kNotRuns;
      }
      wave::sync_wave();
      if (lane == 0) {
        list[0] = list[pending];
      }
      pending = 0;
      wave::sync_wave();
      /* not runs after all (the first KiB was not the chunk): the match finder takes it from the start */
      if (op > (base + 1024) / 4 + 256 || base + 1024 - lit_start > 4096) {
        return kNotRuns;
      }
    }
  }
  /* the chunk's end closes the open stretch like a mismatch */
  if (lane == 0) {
    list[1 + pending] = n;
  }
  wave::sync_wave();
  if (!flush(pending + 1)) {
    return kNotRuns;
  }
  if (n - lit_start > 4096 && op > n / 8) {
    return kNotRuns;
  }
  op += Emitter::tail(dst + op, src + lit_start, n - lit_start);
  return op;
}

} // namespace runs
} // namespace lzm
