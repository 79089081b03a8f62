/*
 * common/lz_match.hip.h -- the wave-parallel greedy match finder shared by the
 * LZ4 and Snappy compressors (one wavefront per chunk).
 *
 * Each step looks at 64 consecutive input positions, one per lane:
 *   1. hash the 4-byte word, probe a per-wave LDS hash table (2-byte entries holding
 *      position mod 65536), verify the candidate; repeats inside the window at the
 *      distances typed columns produce (1, 2, 4, 8 bytes) are found by comparing with
 *      the neighbouring lanes' words;
 *   2. every lane with a verified candidate measures its own match, up to kLaneCap bytes;
 *   3. a scalar walk picks the non-overlapping matches greedily in position order
 *      (a match that hit the cap becomes the last of the step and is extended by the
 *      whole wave, 64 bytes per ballot);
 *   4. a DPP prefix sum over the sequence sizes gives every selected lane its output
 *      offset and the lanes emit their sequences together (literal run + match); only
 *      a sequence with a long literal run or a long match is written cooperatively;
 *   5. the positions the step consumed are inserted into the hash table.
 * Greedy, single probe: the ratio class of the CPU "fast" compressors. Match distances
 * are limited to 65535 (both formats' 2-byte offset forms).
 *
 * Emitter concept (all static):
 *   uint32_t seq_size(lit_len, match_len, offset)                 bytes one sequence takes
 *   bool     is_small(lit_len, match_len)                         can a single lane write it?
 *   void     emit_small_header(dst, lit_len, offset, match_len)   per lane; literals are copied by the caller
 *   uint32_t lit_offset(lit_len)                                  where the literal bytes start inside the sequence
 *   uint32_t match(dst, lit, lit_len, offset, match_len)          whole wave, any size; returns bytes written
 *   uint32_t tail(dst, lit, lit_len)                              whole wave: the final literal-only part
 *   kReach                                                        largest match distance of the format
 * A format whose sequences do not start at byte boundaries (DEFLATE: deflate/deflate_encode.hip.h) sets kStream and
 * takes the sequences through a sink of its own instead -- `Sink* sink`, the last argument of encode_chunk():
 *   void window(sink, src, sel, lit_from, lit_len, match_len, offset, before8, before_ok)   per lane, one step's sequences
 *   void one(sink, lit, lit_len, offset, match_len)                                          whole wave, one sequence
 */
#pragma once

#include "common/lz_common.hip.h"

namespace lzm {

#ifndef NVCOMP_LZM_HASH_BITS
#define NVCOMP_LZM_HASH_BITS 12
#endif
/* One-byte tags beside the positions (A/B build): a probe would look at the candidate's bytes in memory only when eight
 * more hash bits agree. Measured (profiles/archive/r02_lz_compress.json): no gain -- the compressor is bound by the LENGTH of its
 * dependent chain per window, not by the candidate fetches -- and the extra 4 KiB of LDS per wave cost occupancy. Off. */
/* The position side of a window (the 40 bytes around each of its 64 positions) comes out of a per-wave LDS image of the
 * input, filled 512 bytes at a time by ONE coalesced load per lane, instead of four byte-strided lane loads per window:
 * those cost the L1 (TCP) one tag lookup per lane and instruction -- 10.3 G lookups per GiB, one per cycle and CU for
 * the whole kernel (profiles/archive/r02_lz_compress.json, counters). 0 = the round-2 register path (A/B build). */
#ifndef NVCOMP_LZM_STAGE
#define NVCOMP_LZM_STAGE 1
#endif
/* ---- the output staging (NVCOMP_LZM_STAGE_OUT, A/B build) ----
 * The sequences of a window composed in LDS and written with ONE coalesced store of consecutive dwords (up to three bytes
 * wait for the next window) instead of five or six scattered store instructions. Built and measured in round 3
 * (profiles/archive/r03_compress_ab.jsonl): 115.9 against 117.9 GB/s on the mix, 111.3 against 113.7 for Snappy -- the scattered
 * stores of ten lanes are not what the address unit is busy with (the candidate loads of 64 lanes are), and the 256 bytes
 * come out of the hash table. Off. */
#ifndef NVCOMP_LZM_STAGE_OUT
#define NVCOMP_LZM_STAGE_OUT 0
#endif
#ifndef NVCOMP_LZM_TAGS
#define NVCOMP_LZM_TAGS 0
#endif
/* Literal runs of more than 8 bytes come out of the input image instead of memory (emit_small below). */
#ifndef NVCOMP_LZM_LIT_IMAGE
#define NVCOMP_LZM_LIT_IMAGE 1
#endif
/* One step of lazy evaluation in the selection walk. Measured on the mix with the emulator (8 chunks per class, weighted
 * by the mix's shares; liblz4 LZ4_compress_default on the same chunks: 1.8455): greedy 1.7841, lazy 1.8392 at 3 456
 * table entries; 1.7998 -> 1.8601 at 4 096; 1.8408 -> 1.9165 at 8 192. Two or four steps add 0.0007; asking for a
 * margin of one byte loses 0.005. */
#ifndef NVCOMP_LZM_LAZY
#define NVCOMP_LZM_LAZY 1
#endif
/* Positions inside a selected match are not inserted into the table (A/B: 1.7841 -> 1.7976 greedy, 1.8392 -> 1.8430
 * lazy at 3 456 entries; worse than inserting everything once the table has 8 192). */
#ifndef NVCOMP_LZM_INSERT_UNCOVERED
#define NVCOMP_LZM_INSERT_UNCOVERED 0
#endif
#ifndef NVCOMP_LZM_WAVES_PER_SIMD
#define NVCOMP_LZM_WAVES_PER_SIMD 5 /* what the 8 KiB hash table per wave allows (4-wave workgroups, 160 KB LDS per CU) */
#endif
constexpr uint32_t kHashBits = NVCOMP_LZM_HASH_BITS;
/* Entries of the per-wave table (any multiple of 128). With the input image beside it a wave has 8 KiB of LDS at
 * 5 waves/SIMD (4-wave workgroups, 160 KB per CU): 3456 two-byte entries + 1072 bytes of image (3328 when the A/B output
 * staging takes 256 bytes). */
#ifdef NVCOMP_LZM_HASH_ENTRIES
constexpr uint32_t kHashSize = NVCOMP_LZM_HASH_ENTRIES;
#elif NVCOMP_LZM_STAGE
constexpr uint32_t kHashSize = NVCOMP_LZM_STAGE_OUT ? 3328 : 3456;
#else
constexpr uint32_t kHashSize = 1u << kHashBits;
#endif
static_assert(kHashSize % 128 == 0 && kHashSize <= 65536, "the table is cleared 64 dwords at a time");
/* The table a wave owns, in uint16 units: kHashSize positions (mod 65536), followed by kHashSize one-byte tags in the
 * NVCOMP_LZM_TAGS build. */
constexpr uint32_t kTableU16 = kHashSize + (NVCOMP_LZM_TAGS ? kHashSize / 2 : 0);
constexpr uint32_t kMinMatch = 4;
/* The candidate side comes with TWO 16-byte loads per lane (8 bytes before the candidate and 24 from it on) instead of
 * three (32 from it on): a lane's load is a lookup of its own in the CU's address unit whatever its width, and the
 * candidate loads were 40 % of a window's lookups in a kernel that is bound by exactly that unit. The per-lane
 * measurement then stops at 24 bytes; longer matches are finished by the whole wave, as before. */
#ifndef NVCOMP_LZM_CAND32
#define NVCOMP_LZM_CAND32 1
#endif
constexpr uint32_t kLaneCap = NVCOMP_LZM_CAND32 ? 24 : 32; /* per-lane match measurement, compared in registers */
/* Hit lanes in a window from which its first match is probed cooperatively. Runs and periodic columns hit in (nearly)
 * every lane; text hits in about half of them, and at 32 the probe ran on 40 % of its windows to find nothing
 * (60 against 32: +4 % on the mix, +8 % on the int32 column, same ratio). */
#ifndef NVCOMP_LZM_DENSE_HITS
#define NVCOMP_LZM_DENSE_HITS 60
#endif
constexpr uint32_t kDenseHits = NVCOMP_LZM_DENSE_HITS;
constexpr uint32_t kBackMax = 8;    /* bytes a lane's match may grow backwards over its literal run */

/* ---- phase clock (profiling builds only: -DNVCOMP_LZM_PROF, scripts/build_variants.sh) ---- */
#ifdef NVCOMP_LZM_PROF
constexpr uint32_t kProfSlots = 12;
__device__ unsigned long long g_prof[kProfSlots];
struct ProfClock
{
  unsigned long long acc[kProfSlots];
  unsigned long long last;
  __device__ __forceinline__ void begin()
  {
    for (uint32_t i = 0; i < kProfSlots; ++i) {
      acc[i] = 0;
    }
    last = __builtin_readcyclecounter();
  }
  __device__ __forceinline__ void mark(uint32_t slot)
  {
    const unsigned long long t = __builtin_readcyclecounter();
    acc[slot] += t - last;
    last = t;
  }
  __device__ __forceinline__ void end()
  {
    if (wave::lane_id() == 0) {
      for (uint32_t i = 0; i < kProfSlots; ++i) {
        atomicAdd(&g_prof[i], acc[i]);
      }
    }
  }
};
#define LZM_PROF_DECL lzm::ProfClock prof_clock; prof_clock.begin()
#define LZM_T(slot) prof_clock.mark(slot)
#define LZM_PROF_END prof_clock.end()
#else
#define LZM_PROF_DECL ((void)0)
#define LZM_T(slot) ((void)0)
#define LZM_PROF_END ((void)0)
#endif

__device__ __forceinline__ uint32_t hash4(uint32_t v)
{
  const uint32_t h = v * 2654435761u;
  if constexpr ((kHashSize & (kHashSize - 1)) == 0) {
    return h >> (32 - kHashBits);
  } else {
    return __umulhi(h, kHashSize); /* multiply-shift range reduction onto [0, kHashSize) */
  }
}

/* Eight hash bits that do not take part in the index. */
__device__ __forceinline__ uint32_t tag4(uint32_t v)
{
  const uint32_t h = v * 2654435761u;
  if constexpr ((kHashSize & (kHashSize - 1)) == 0) {
    return (h >> (24 - kHashBits)) & 0xffu;
  } else {
    return (h >> 6) & 0xffu;
  }
}

/* Wave-cooperative extension of a match known to be at least `have` bytes long: 64 bytes per step, compared as 16
 * dwords by 16 lanes (a lane's load is a lookup of its own in the CU's address unit, whatever its width: bytes by all 64
 * lanes cost four times as many). The last one to three bytes in front of match_end are compared singly. */
__device__ __forceinline__ uint32_t extend_match(
    const uint8_t* __restrict__ src, uint32_t mpos, uint32_t mcand, uint32_t have, uint32_t match_end)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  uint32_t mlen = have;
  for (;;) {
    const uint32_t p = mpos + mlen + 4 * lane;
    const bool mine = lane < 16;
    const bool whole = mine && p + 4 <= match_end;
    uint32_t x = 0;
    if (whole) {
      x = wave::gload_u32(src + p) ^ wave::gload_u32(src + mcand + mlen + 4 * lane);
    }
    const uint64_t stop = wave::ballot(mine && (x != 0 || !whole));
    if (stop != 0) {
      const uint32_t f = wave::ctz64(stop);
      const uint32_t xf = wave::read_lane(x, f);
      mlen += 4 * f;
      if (xf != 0) {
        return mlen + ((uint32_t)__builtin_ctz(xf) >> 3);
      }
      /* fewer than four bytes are left in front of match_end */
      const uint32_t q = mpos + mlen + lane;
      const bool same = q < match_end && wave::gload_u8(src + q) == wave::gload_u8(src + mcand + mlen + lane);
      const uint64_t diff = ~wave::ballot(same);
      return mlen + wave::ctz64(diff);
    }
    mlen += 64;
  }
}

/* The stream around a position, in registers: bytes [pos - 8, pos) and [pos, pos + 32). */
struct Around
{
  uint32_t pre[2];
  uint32_t fwd[8];
};

/* Both loads are unaligned 16-byte lane loads of consecutive addresses one byte apart: a wave touches two or three
 * cache lines. `with_pre` = pos >= 8. The caller guarantees pos + 32 <= the end of the chunk. */
__device__ __forceinline__ void load_around(Around& r, const uint8_t* __restrict__ src, uint32_t pos, bool with_pre)
{
  const wave::u32x4 a = wave::gload_u32x4(src + pos);
  const wave::u32x4 b = wave::gload_u32x4(src + pos + 16);
  r.fwd[0] = a.x, r.fwd[1] = a.y, r.fwd[2] = a.z, r.fwd[3] = a.w;
  r.fwd[4] = b.x, r.fwd[5] = b.y, r.fwd[6] = b.z, r.fwd[7] = b.w;
  r.pre[0] = 0, r.pre[1] = 0;
  if (with_pre) {
    r.pre[0] = wave::gload_u32(src + pos - 8);
    r.pre[1] = wave::gload_u32(src + pos - 4);
  }
}

/* ---- the input image (NVCOMP_LZM_STAGE) ----
 * Two 512-byte blocks of the chunk, block b (bytes [512 b, 512 b + 512)) in slot b & 1, so that byte p lives at image
 * offset p & 1023; the first kStageMirror bytes of slot 0 are repeated behind slot 1, which lets a lane read its 44
 * bytes from ascending addresses whichever way round the two blocks lie. A window needs at most two consecutive
 * blocks. Lane l carries bytes [8 l, 8 l + 8) of a block. */
constexpr uint32_t kStageBlock = 512;
constexpr uint32_t kStageMirror = 48;
constexpr uint32_t kStageBytes = NVCOMP_LZM_STAGE ? 2 * kStageBlock + kStageMirror : 8;
constexpr uint32_t kNoBlock = ~0u;
constexpr uint32_t kOutStage = NVCOMP_LZM_STAGE_OUT ? 256 : 0;
constexpr uint32_t kImageBytes = kStageBytes + kOutStage; /* what a wave's `image` argument points at */

__device__ __forceinline__ uint64_t stage_fetch(const uint8_t* __restrict__ src, uint32_t n, uint32_t blk)
{
  const uint32_t at = blk * kStageBlock + 8 * (uint32_t)wave::lane_id();
  return at + 8 <= n ? wave::gload_u64(src + at) : 0ull; /* a piece that crosses the end is never looked at */
}

__device__ __forceinline__ void stage_commit(uint8_t* image, uint32_t blk, uint64_t piece)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  const uint32_t slot = blk & 1u;
  *(uint64_t*)(image + slot * kStageBlock + 8 * lane) = piece;
  if (slot == 0 && lane < kStageMirror / 8) {
    *(uint64_t*)(image + 2 * kStageBlock + 8 * lane) = piece;
  }
}

/* load_around() out of the image: eleven aligned dwords, realigned by the position's low two bits. */
__device__ __forceinline__ void stage_around(Around& r, const uint8_t* image, uint32_t pos)
{
  const uint32_t from = (pos - 8u) & (2 * kStageBlock - 1);
  const uint32_t* q = (const uint32_t*)(image + (from & ~3u));
  const uint32_t sh = pos & 3u;
  uint32_t w[11];
#pragma unroll
  for (uint32_t i = 0; i < 11; ++i) {
    w[i] = q[i];
  }
  r.pre[0] = wave::align_bytes(w[1], w[0], sh);
  r.pre[1] = wave::align_bytes(w[2], w[1], sh);
#pragma unroll
  for (uint32_t i = 0; i < 8; ++i) {
    r.fwd[i] = wave::align_bytes(w[i + 3], w[i + 2], sh);
  }
  if (pos < 8) { /* the first eight positions of a chunk have nothing before them */
    r.pre[0] = 0, r.pre[1] = 0;
  }
}

/* What one lane knows about its position after the probe. */
struct Probe
{
  uint32_t word; /* the 4 bytes at the position */
  uint32_t cand; /* candidate position (valid when found) */
  uint32_t mlen; /* match length, capped at kLaneCap (0 when not found) */
  uint32_t back; /* equal bytes right before position and candidate, at most kBackMax */
  bool found;
};

/* Candidate from the hash table entry `low` (position mod 65536): the nearest position below pos with these low
 * bits; `ok` = it exists and is within the formats' 65535-byte reach. */
__device__ __forceinline__ uint32_t table_candidate(uint32_t pos, uint32_t low, bool& ok, uint32_t reach = 65535u)
{
  uint32_t cand = (pos & ~0xffffu) | low;
  if (cand >= pos) {
    cand -= 0x10000u;
  }
  ok = cand < pos && pos - cand <= reach; /* cand wraps to a huge value when there is none */
  return cand;
}

/* A repeat of the word at distance 1, 2, 4 or 8 inside the window (runs, typed columns): found by comparing with
 * the neighbouring lanes' words, no memory access. Returns the distance or 0. */
__device__ __forceinline__ uint32_t neighbour_repeat(uint32_t word, bool eligible, uint32_t stride)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  uint32_t dist = 0;
  for (uint32_t d = 1; d <= 8; d *= 2) {
    const uint32_t other = wave::shuffle(word, (lane - d) & 63u);
    if (eligible && dist == 0 && lane >= d && other == word) {
      dist = d * stride;
    }
  }
  return dist;
}

/*
 * Probe of a window whose 64 positions can all read 32 bytes ahead (every window but the last one or two of a
 * chunk). The compressor is bound by dependent memory round trips, so a step has exactly two:
 * the position side (`me`, loaded one step AHEAD by the caller) and ONE round of candidate-side loads -- word,
 * the 28 bytes behind it and the 8 bytes before it together -- after which the match length up to kLaneCap and
 * the backward growth are plain register compares. A neighbour repeat is preferred to the table's candidate: it
 * is known to match without looking.
 */
__device__ __forceinline__ Probe probe_fast(
    const uint8_t* __restrict__ src, const uint16_t* table, const Around& me, uint32_t pos, bool eligible, uint32_t match_end,
    uint32_t stride, uint32_t reach = 65535u)
{
  Probe p;
  p.word = me.fwd[0];
  p.mlen = 0;
  p.back = 0;
  bool ok;
  {
    const uint32_t slot = hash4(p.word);
    p.cand = table_candidate(pos, table[slot], ok, reach);
#if NVCOMP_LZM_TAGS
    ok = ok && ((const uint8_t*)(table + kHashSize))[slot] == tag4(p.word);
#endif
  }
  const uint32_t near = neighbour_repeat(p.word, eligible, stride);
  if (near) {
    p.cand = pos - near;
    ok = true;
  }
  ok = ok && eligible;
  Around c;
#pragma unroll
  for (uint32_t i = 0; i < 8; ++i) {
    c.fwd[i] = ~me.fwd[i]; /* lanes without a candidate compare unequal */
  }
  c.pre[0] = 0, c.pre[1] = 0;
  if (ok) {
#if NVCOMP_LZM_STAGE
    /* 40 bytes from 8 before the candidate in three loads (a candidate in the first 8 bytes of the chunk: 40 from it) */
    const bool cback = p.cand >= kBackMax;
    const uint8_t* cp = src + p.cand - (cback ? kBackMax : 0u);
    const wave::u32x4 a = wave::gload_u32x4(cp);
    const wave::u32x4 b = wave::gload_u32x4(cp + 16);
#if NVCOMP_LZM_CAND32
    const uint64_t d = ~(((uint64_t)me.fwd[7] << 32) | me.fwd[6]); /* not fetched: bytes 24..31 compare unequal */
#else
    const uint64_t d = wave::gload_u64(cp + 32);
#endif
    c.pre[0] = cback ? a.x : 0u;
    c.pre[1] = cback ? a.y : 0u;
    c.fwd[0] = cback ? a.z : a.x;
    c.fwd[1] = cback ? a.w : a.y;
    c.fwd[2] = cback ? b.x : a.z;
    c.fwd[3] = cback ? b.y : a.w;
    c.fwd[4] = cback ? b.z : b.x;
    c.fwd[5] = cback ? b.w : b.y;
    c.fwd[6] = cback ? (uint32_t)d : b.z;
    c.fwd[7] = cback ? (uint32_t)(d >> 32) : b.w;
#else
    load_around(c, src, p.cand, p.cand >= kBackMax);
#endif
  }
  uint32_t x[8];
#pragma unroll
  for (uint32_t i = 0; i < 8; ++i) {
    x[i] = me.fwd[i] ^ c.fwd[i];
  }
  p.found = ok && x[0] == 0;
  /* first differing byte among bytes 4..31 */
  uint32_t idx = 8, xv = 0;
#pragma unroll
  for (uint32_t i = 7; i >= 1; --i) {
    idx = x[i] ? i : idx;
    xv = x[i] ? x[i] : xv;
  }
  uint32_t mlen = 4 * idx + (xv ? (uint32_t)__builtin_ctz(xv) >> 3 : 0u);
  mlen = mlen < kLaneCap ? mlen : kLaneCap; /* a candidate in the first 8 bytes of the chunk brought 32 */
  const uint32_t room = match_end - pos; /* >= 4 for an eligible position */
  mlen = mlen < room ? mlen : room;
  p.mlen = p.found ? mlen : 0;
  if (p.found && p.cand >= kBackMax) { /* pos > cand >= 8: both sides hold their 8 bytes */
    const uint64_t xb = ((uint64_t)(me.pre[1] ^ c.pre[1]) << 32) | (me.pre[0] ^ c.pre[0]);
    p.back = xb ? (uint32_t)__builtin_clzll(xb) >> 3 : kBackMax;
  }
  return p;
}

/* The same probe with nothing assumed about how far a position may read: the last windows of a chunk. */
__device__ __forceinline__ Probe probe_safe(
    const uint8_t* __restrict__ src, const uint16_t* table, uint32_t pos, bool eligible, uint32_t match_end, uint32_t stride,
    uint32_t reach = 65535u)
{
  Probe p;
  p.word = 0;
  p.cand = 0;
  p.mlen = 0;
  p.back = 0;
  p.found = false;
  bool ok = false;
  if (eligible) {
    p.word = lz::ld_u32(src + pos);
    const uint32_t slot = hash4(p.word);
    p.cand = table_candidate(pos, table[slot], ok, reach);
#if NVCOMP_LZM_TAGS
    ok = ok && ((const uint8_t*)(table + kHashSize))[slot] == tag4(p.word);
#endif
  }
  const uint32_t near = neighbour_repeat(p.word, eligible, stride);
  if (near) {
    p.cand = pos - near;
    p.found = true;
  } else if (ok) {
    p.found = lz::ld_u32(src + p.cand) == p.word;
  }
  if (p.found) {
    const uint32_t room = match_end - pos;
    const uint32_t cap = room < kLaneCap ? room : kLaneCap;
    uint32_t mlen = kMinMatch;
    bool open = true;
    while (open && mlen + 4 <= cap) {
      const uint32_t x = lz::ld_u32(src + pos + mlen) ^ lz::ld_u32(src + p.cand + mlen);
      if (x != 0) {
        mlen += (uint32_t)__builtin_ctz(x) >> 3;
        open = false;
      } else {
        mlen += 4;
      }
    }
    while (open && mlen < cap && src[pos + mlen] == src[p.cand + mlen]) { /* at most 3 tail bytes */
      ++mlen;
    }
    p.mlen = mlen;
    while (p.back < kBackMax && p.back < p.cand && src[pos - 1 - p.back] == src[p.cand - 1 - p.back]) {
      ++p.back;
    }
  }
  return p;
}

/*
 * STRIDE = bytes between the positions two neighbouring lanes look at: 1 for untyped data; 2, 4 or 8 when the caller
 * declared the chunk an array of 2-, 4- or 8-byte elements (nvcompBatchedLZ4Opts_t.data_type, reference
 * CHANGELOG.md:168-169 "an optimization to the LZ4 compressor based on specification of input data as char, short, or
 * int"): matches then start at element boundaries only and lie a whole number of elements back, a window covers
 * 64 x STRIDE bytes per step, and the neighbour compares look 1, 2, 4 and 8 ELEMENTS back.
 */
template <class Emitter, uint32_t STRIDE = 1, class Sink = uint32_t>
__device__ __forceinline__ uint32_t encode_chunk(
    const uint8_t* __restrict__ src, uint32_t n, uint8_t* dst, uint16_t* table, uint8_t* image, uint32_t last_start,
    uint32_t match_end, bool any_match, Sink* sink = nullptr)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  for (uint32_t i = lane; i < kTableU16 / 2; i += 64) {
    ((uint32_t*)table)[i] = 0;
  }
  wave::sync();

  uint32_t op = 0;
  uint32_t anchor = 0;
  /* the output staging: dst[op - fill, op) is still in stage_out[0, fill) (fill < 4 between windows) */
  uint8_t* stage_out = image + kStageBytes;
  uint32_t fill = 0;
  auto flush_staged = [&]() {
    if (lane < fill) {
      wave::gstore_u8(dst + op - fill + lane, stage_out[lane]);
    }
    fill = 0;
  };
  LZM_PROF_DECL;
  if (any_match) {
    uint32_t ip = 0;
    constexpr uint32_t kWin = 64 * STRIDE; /* bytes a window covers */
    uint32_t skip = 0;        /* leading BYTES of the window that the previous step's last match already covers */
#if NVCOMP_LZM_STAGE
    uint32_t have0 = kNoBlock, have1 = kNoBlock; /* the blocks in the image's two slots */
    uint32_t coming = kNoBlock;                  /* the block requested ahead, in `piece` until its slot is free */
    uint64_t piece = 0;
#else
    Around ahead;             /* position side of window `ahead_ip`, requested one step early */
    uint32_t ahead_ip = ~0u;
#pragma unroll
    for (uint32_t i = 0; i < 8; ++i) {
      ahead.fwd[i] = 0;
    }
    ahead.pre[0] = 0, ahead.pre[1] = 0;
#endif
    while (ip <= last_start) {
      const uint32_t pos = ip + lane * STRIDE;
      const bool eligible = pos <= last_start;
      /* wave-uniform: every lane may read its 32 bytes (and the 8-byte pieces holding them lie inside the chunk) */
      const bool fast = ip + 63 * STRIDE + 32 + (NVCOMP_LZM_STAGE ? 7 : 0) <= n;
      LZM_T(0); /* loop top */
      Probe pr;
      Around me; /* fast windows: the bytes around this lane's position stay in registers until its literals are written */
      me.pre[0] = 0, me.pre[1] = 0;
      if (fast) {
#if NVCOMP_LZM_STAGE
        const uint32_t first = (ip >= 8 ? ip - 8 : 0u) / kStageBlock;
        const uint32_t last = (ip + 63 * STRIDE + 31) / kStageBlock; /* first or first + 1 */
        if (coming != kNoBlock) {
          const uint32_t there = coming & 1u ? have1 : have0;
          if (there != first && there != last) { /* nothing this window reads is overwritten */
            stage_commit(image, coming, piece);
            have0 = coming & 1u ? have0 : coming;
            have1 = coming & 1u ? coming : have1;
            coming = kNoBlock;
          }
        }
        if ((first & 1u ? have1 : have0) != first) { /* start of the chunk, or the window behind a long match */
          stage_commit(image, first, stage_fetch(src, n, first));
          have0 = first & 1u ? have0 : first;
          have1 = first & 1u ? first : have1;
        }
        if ((last & 1u ? have1 : have0) != last) {
          stage_commit(image, last, stage_fetch(src, n, last));
          have0 = last & 1u ? have0 : last;
          have1 = last & 1u ? last : have1;
        }
        wave::sync();
        stage_around(me, image, pos);
        if (coming == kNoBlock) { /* the block behind this window travels while the window is worked on */
          const uint32_t next = last + 1;
          if (next * kStageBlock < n && (next & 1u ? have1 : have0) != next) {
            piece = stage_fetch(src, n, next);
            coming = next;
          }
        }
#else
        if (ahead_ip == ip) {
          me = ahead;
        } else {
          load_around(me, src, pos, pos >= kBackMax);
        }
        if (ip + kWin + 63 * STRIDE + 32 <= n) { /* the next window's position side travels while this one is worked on */
          ahead_ip = ip + kWin;
          load_around(ahead, src, pos + kWin, true);
        }
#endif
        pr = probe_fast(src, table, me, pos, eligible, match_end, STRIDE, Emitter::kReach);
      } else {
        pr = probe_safe(src, table, pos, eligible, match_end, STRIDE, Emitter::kReach);
      }
      const uint32_t word = pr.word;
      const uint32_t cand = pr.cand;
      uint32_t mlen = pr.mlen;
      LZM_T(1); /* probe: position side, table, candidate loads, compares */
      /* this window's positions go into the table now: the next probe (whose loads may already be under way)
       * sees them, and nothing below reads the table */
      wave::sync();
      auto insert_lanes = [&](bool mine) {
        if (mine) {
          const uint32_t slot = hash4(word);
          table[slot] = (uint16_t)pos;
#if NVCOMP_LZM_TAGS
          ((uint8_t*)(table + kHashSize))[slot] = (uint8_t)tag4(word);
#endif
        }
        wave::sync();
      };
#if !NVCOMP_LZM_INSERT_UNCOVERED
      insert_lanes(eligible);
#endif
      const uint64_t hits = wave::ballot(pr.found && lane * STRIDE >= skip);
      LZM_T(2); /* table insert */
      if (hits == 0) {
#if NVCOMP_LZM_INSERT_UNCOVERED
        insert_lanes(eligible && lane * STRIDE >= skip);
#endif
        /* no match starts in this window: its positions become literals */
        skip = skip > kWin ? skip - kWin : 0;
        ip += kWin;
        continue;
      }

      /* ---- long first match (runs, periodic columns): one cooperative probe decides. Only windows where most
       * positions hit are worth the extra memory round trip (on text a window has 10-20 hits and the per-lane
       * measurement finds the rare long match anyway) ---- */
      if (wave::popc64(hits) >= kDenseHits) {
        const uint32_t f0 = wave::ctz64(hits);
        uint32_t mpos = ip + f0 * STRIDE;
        uint32_t mcand = wave::read_lane(cand, f0);
        const uint32_t p = mpos + kMinMatch + lane;
        const bool same = p < match_end && src[p] == src[mcand + kMinMatch + lane];
        const uint64_t diff = ~wave::ballot(same);
        uint32_t len0 = diff ? kMinMatch + wave::ctz64(diff) : extend_match(src, mpos, mcand, kMinMatch + 64, match_end);
        if (len0 >= kLaneCap) {
#if NVCOMP_LZM_INSERT_UNCOVERED
          insert_lanes(eligible && lane <= f0 && lane * STRIDE >= skip);
#endif
          /* grow it backwards over the pending literals (what the CPU compressors call catching up): lane l
           * compares the l-th byte before the match with the l-th byte before its source */
          {
            const uint32_t room = mpos - anchor < mcand ? mpos - anchor : mcand;
            const bool eq = lane < room && src[mpos - 1 - lane] == src[mcand - 1 - lane];
            const uint64_t ne = ~wave::ballot(eq);
            const uint32_t back = ne ? wave::ctz64(ne) : 64u;
            mpos -= back;
            mcand -= back;
            len0 += back;
          }
          if constexpr (Emitter::kStream) {
            Emitter::one(*sink, src + anchor, mpos - anchor, mpos - mcand, len0);
          } else {
            flush_staged();
            op += Emitter::match(dst + op, src + anchor, mpos - anchor, mpos - mcand, len0);
          }
          const uint32_t next = mpos + len0;
          anchor = next;
          /* the next window starts right behind the match (at the next element boundary): every one of its 64
           * positions is searched; the position side requested ahead is not the data it needs and is loaded again */
          ip = (next + STRIDE - 1) / STRIDE * STRIDE;
          skip = 0;
          continue;
        }
      }

      LZM_T(3); /* dense-window check */
      /* ---- greedy selection in position order (scalar walk over the hit mask) ----
       * The walk is the serial part of a window (a sixth of the compressor's time): it carries nothing but the mask --
       * find the next hit, read its length, mask off what the match covers. Where each selected lane's literal run
       * starts (the end of the match selected before it) is worked out afterwards, for all of them at once. */
      uint64_t selected = 0;
#if NVCOMP_LZM_INSERT_UNCOVERED
      uint64_t covered = 0;
#endif
      uint32_t cur = 0;       /* window-relative BYTE position the next match may start at */
      uint64_t rest = hits;
      while (rest) {
        uint32_t f = wave::ctz64(rest);
        uint32_t flen = wave::read_lane(mlen, f);
#if NVCOMP_LZM_LAZY
        /* one step of lazy evaluation: every lane has measured its own match, so "does the next position start a
         * longer one?" is one more lane read. The shorter match becomes a literal. */
        if (f < 63 && ((hits >> (f + 1)) & 1) && flen < kLaneCap) {
          const uint32_t l1 = wave::read_lane(mlen, f + 1);
          if (l1 > flen) {
            f = f + 1;
            flen = l1;
          }
        }
#endif
        selected |= 1ull << f;
        cur = f * STRIDE + flen;
        if (flen >= kLaneCap) { /* the capped match may be much longer: measure it with the whole wave */
          const uint32_t full = extend_match(src, ip + f * STRIDE, wave::read_lane(cand, f), kLaneCap, match_end);
          mlen = wave::write_lane(mlen, full, f);
          cur = f * STRIDE + full;
        }
        {
          const uint32_t next_lane = (cur + STRIDE - 1) / STRIDE; /* first lane at or behind the match's end */
          rest = next_lane < 64 ? (hits & (~0ull << next_lane)) : 0ull;
#if NVCOMP_LZM_INSERT_UNCOVERED
          covered |= (next_lane < 64 ? ~(~0ull << next_lane) : ~0ull) & (f < 63 ? (~0ull << (f + 1)) : 0ull);
#endif
        }
      }
#if NVCOMP_LZM_INSERT_UNCOVERED
      insert_lanes(eligible && lane * STRIDE >= skip && !((covered >> lane) & 1));
#endif
      const uint32_t lit_from = ip + cur; /* behind the last selected match */
      /* a selected lane's run starts where the selected match below it ends (the first one's: at the anchor) */
      uint32_t prev_end;
      {
        const uint64_t below = selected & ((1ull << lane) - 1);
        const uint32_t prev = below ? 63u - (uint32_t)__builtin_clzll(below) : 0u;
        const uint32_t prev_stop = wave::shuffle(pos + mlen, prev);
        prev_end = below ? prev_stop : anchor;
      }

      LZM_T(4); /* selection */
      /* ---- emit the selected sequences ---- */
      const bool sel = (selected >> lane) & 1;
      uint32_t lit_len = sel ? pos - prev_end : 0;
      uint32_t my_len = sel ? mlen : 0;
      /* every selected match grows backwards over its own literal run (never into the previous match) */
      {
        const uint32_t back = pr.back < lit_len ? pr.back : lit_len;
        lit_len -= sel ? back : 0u;
        my_len += sel ? back : 0u;
      }
      const uint32_t offset = pos - cand;
      if constexpr (Emitter::kStream) {
        const uint32_t run = sel ? pos - prev_end : 0; /* before the match grew backwards: the run ends at pos */
        Emitter::window(*sink, src, sel, prev_end, lit_len, my_len, offset, ((uint64_t)me.pre[1] << 32) | me.pre[0],
                        fast && run <= 8 && pos >= kBackMax, run);
      } else {
        const uint32_t size = sel ? Emitter::seq_size(lit_len, my_len, offset) : 0;
        const uint32_t incl = wave::scan_add_inclusive(size);
        const uint32_t total = wave::read_lane(incl, 63);
        const bool small = sel && Emitter::is_small(lit_len, my_len);
        uint64_t big = wave::ballot(sel && !small);
        LZM_T(5); /* sizes + scan */
        /* the small sequences of the window, written by their own lanes at base + (their offset in the window's output) */
        auto emit_small = [&](uint8_t* base) {
          uint8_t* my_dst = base + incl - size;
          if (small) {
            Emitter::emit_small_header(my_dst, lit_len, offset, my_len);
          }
          LZM_T(6); /* headers */
          /* literal runs of the small sequences. Up to 8 bytes (nearly all of them on text) are the bytes right before
           * the lane's position, which a fast window holds in registers: no load at all. Longer runs are read back,
           * four dwords in flight per round (a load-store pair per step costs a memory round trip per step). */
          const uint32_t run = sel ? pos - prev_end : 0; /* before the match grew backwards: the run ends at pos */
          uint8_t* ld = my_dst + Emitter::lit_offset(lit_len);
          const bool from_regs = small && fast && run <= 8 && pos >= kBackMax;
          if (from_regs && lit_len != 0) {
            const uint64_t before = (((uint64_t)me.pre[1] << 32) | me.pre[0]) >> (8 * (8 - run));
            if (lit_len >= 4) {
              lz::st_u32(ld, (uint32_t)before);
              lz::st_u32(ld + lit_len - 4, (uint32_t)(before >> (8 * (lit_len - 4))));
            } else {
              ld[0] = (uint8_t)before;
              if (lit_len > 1) {
                ld[1] = (uint8_t)(before >> 8);
              }
              if (lit_len > 2) {
                ld[2] = (uint8_t)(before >> 16);
              }
            }
          }
          const bool from_mem = small && !from_regs && lit_len != 0;
          const uint8_t* ls = src + prev_end;
#if NVCOMP_LZM_STAGE && NVCOMP_LZM_LIT_IMAGE
          /* A run of more than 8 bytes is read back -- out of the input image when both its ends lie in the two blocks
           * the image holds (nearly always: a run of up to 64 bytes that ends inside this window): six windows in ten
           * have such a run, and read from memory each cost its window a dependent round trip (the phase clock put
           * 17 % of the compressor's time into this copy). */
          const uint32_t blk_a = prev_end / kStageBlock, blk_b = (prev_end + lit_len - 1) / kStageBlock;
          const bool in_image = fast && (blk_a & 1u ? have1 : have0) == blk_a && (blk_b & 1u ? have1 : have0) == blk_b;
#else
          const bool in_image = false;
#endif
          const bool lit4 = from_mem && lit_len >= 4;
          for (uint32_t base4 = 0; wave::ballot(lit4 && lit_len > base4) != 0; base4 += 16) {
            if (lit4 && lit_len > base4) {
              const uint32_t lastoff = lit_len - 4;
              uint32_t v[4];
#pragma unroll
              for (uint32_t i = 0; i < 4; ++i) {
                const uint32_t o = base4 + 4 * i < lastoff ? base4 + 4 * i : lastoff;
                /* a dword that starts in the last three bytes of the image's second block runs into the mirror of the first */
                v[i] = in_image ? lz::ld_u32(image + ((prev_end + o) & (2 * kStageBlock - 1))) : wave::gload_u32(ls + o);
              }
#pragma unroll
              for (uint32_t i = 0; i < 4; ++i) {
                const uint32_t o = base4 + 4 * i < lastoff ? base4 + 4 * i : lastoff;
                lz::st_u32(ld + o, v[i]);
              }
            }
          }
          if (from_mem && lit_len < 4) {
            ld[0] = (uint8_t)wave::gload_u8(ls);
            if (lit_len > 1) {
              ld[1] = (uint8_t)wave::gload_u8(ls + 1);
            }
            if (lit_len > 2) {
              ld[2] = (uint8_t)wave::gload_u8(ls + 2);
            }
          }
          LZM_T(7); /* literals */
        };
        if (NVCOMP_LZM_STAGE_OUT && big == 0 && fill + total <= kOutStage) {
          /* composed in LDS behind the bytes still staged, then out with one coalesced store of whole dwords */
          emit_small(stage_out + fill);
          wave::sync();
          const uint32_t bytes = fill + total;
          const uint32_t nd = bytes >> 2;
          uint8_t* g = dst + op - fill;
          for (uint32_t i = lane; i < nd; i += 64) {
            wave::gstore_u32(g + 4 * i, *(const uint32_t*)(stage_out + 4 * i));
          }
          fill = bytes & 3u;
          if (fill != 0) {
            const uint32_t carry = *(const uint32_t*)(stage_out + 4 * nd); /* every lane reads and writes the same dword */
            wave::sync();
            *(uint32_t*)stage_out = carry;
          }
          wave::sync();
        } else {
          flush_staged();
          emit_small(dst + op);
          /* the few sequences a single lane cannot write: long literal run (first of the step,
           * after match-less windows) or long match (last of the step) */
          while (big) {
            const uint32_t j = wave::ctz64(big);
            big &= big - 1;
            const uint32_t jdst = op + wave::read_lane(incl - size, j);
            const uint32_t jlit = wave::read_lane(prev_end, j);
            Emitter::match(dst + jdst, src + jlit, wave::read_lane(lit_len, j), wave::read_lane(offset, j), wave::read_lane(my_len, j));
          }
        }
        op += total;
      }
      LZM_T(8); /* cooperative sequences */
      anchor = lit_from;
      /* the next window starts 64 positions on whatever the last match covers of it (so that the data requested
       * ahead is the data needed); whole windows inside a long match are skipped */
      if (cur >= 2 * kWin) { /* a long match measured by the whole wave: go on right behind it */
        ip = (ip + cur + STRIDE - 1) / STRIDE * STRIDE;
        skip = 0;
      } else {
        ip += kWin;
        skip = cur > kWin ? cur - kWin : 0;
      }
    }
  }
  if constexpr (Emitter::kStream) {
    Emitter::one(*sink, src + anchor, n - anchor, 0, 0);
  } else {
    flush_staged();
    op += Emitter::tail(dst + op, src + anchor, n - anchor);
  }
  LZM_T(9);
  LZM_PROF_END;
  return op;
}

} // namespace lzm
