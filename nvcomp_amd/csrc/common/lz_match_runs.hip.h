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

/* Compress src[0, n) as runs of period 1, 2, 4 or 8 with the calling wave. last_start / match_end: the format's
 * end-of-block rules (the last match starts at or before last_start and ends at or before match_end). Returns the
 * compressed size, or kNotRuns when the chunk is not of this kind (dst may have been written to). */
template <class Emitter>
__device__ __forceinline__ uint32_t encode_chunk(
    const uint8_t* __restrict__ src, uint32_t n, uint8_t* dst, uint32_t last_start, uint32_t match_end, bool any_match)
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
  /* ---- the chunk, 1 KiB a step ---- */
  uint32_t op = 0;
  uint32_t lit_start = 0; /* everything in front of it is written */
  uint32_t run_start = 0; /* the stretch of bytes equal to their p-th predecessor that is open: [run_start, here) (the first p
                           * positions count as mismatches: it opens at p) */
  for (uint32_t base = 0; base < n; base += 1024) {
    const uint32_t at = base + 16 * lane;
    const wave::u32x4 cur = load16_guarded(src, n, at);
    wave::u32x4 prv = {0, 0, 0, 0};
    if (at >= 16 && at < n) {
      prv = wave::gload_u32x4(src + at - 16); /* ends in front of `at`: inside the chunk */
    }
    uint32_t mis = mismatch16(prv, cur, p);
    /* the positions without a predecessor, and those behind the chunk, end a stretch */
    if (at < p) {
      mis |= (1u << (p - at)) - 1u;
    }
    if (at + 16 > n) {
      mis |= at >= n ? 0xffffu : (0xffffu << (n - at)) & 0xffffu;
    }
    for (uint64_t lanes = wave::ballot(mis != 0); lanes; lanes &= lanes - 1) {
      const uint32_t j = wave::ctz64(lanes);
      uint32_t m = wave::read_lane(mis, j);
      const uint32_t jat = base + 16 * j;
      if (jat >= n) {
        break; /* behind the chunk: the end below closes the open stretch */
      }
      while (m) {
        const uint32_t q = jat + (uint32_t)__builtin_ctz(m); /* byte q differs from byte q - p */
        m &= m - 1;
        if (q >= n) {
          break;
        }
        const uint32_t end = q < match_end ? q : match_end;
        if (run_start <= last_start && end >= run_start + kMinMatch) {
          op += Emitter::match(dst + op, src + lit_start, run_start - lit_start, p, end - run_start);
          lit_start = end;
        }
        run_start = q + 1;
      }
    }
    /* not runs after all (the first KiB was not the chunk): the match finder takes it from the start */
    if (op > (base + 1024) / 4 + 256) {
      return kNotRuns;
    }
  }
  if (run_start <= last_start && match_end >= run_start + kMinMatch) { /* the stretch that is open at the chunk's end */
    op += Emitter::match(dst + op, src + lit_start, run_start - lit_start, p, match_end - run_start);
    lit_start = match_end;
  }
  op += Emitter::tail(dst + op, src + lit_start, n - lit_start);
  return op;
}

} // namespace runs
} // namespace lzm
