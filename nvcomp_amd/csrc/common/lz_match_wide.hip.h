/*
 * common/lz_match_wide.hip.h -- the match finder of the LZ4 / Snappy compressors for untyped data, round 5: one wavefront
 * per chunk as in common/lz_match.hip.h, but a STEP covers 256 positions (four sub-windows of 64, lane l of sub-window k
 * at position ip + 64 k + l) and the expensive parts run on COMPACTED work:
 *
 *   1. probe (all 256 positions): the word at the position out of the LDS image of the input, hash, table entry ->
 *      candidate, table insert -- sub-window after sub-window, so that a probe sees every position in front of its own
 *      sub-window, exactly as the one-window compressor; repeats 1 / 2 / 4 / 8 bytes back by comparing with the
 *      neighbouring lanes' words;
 *   2. word check: ONE dword load per position with a table candidate (four in flight per lane): three candidates in
 *      four are hash collisions, and they stop here;
 *   3. measure: the positions whose candidate holds the same word are compacted into a queue in LDS (ballot + popcount)
 *      and measured 64 at a time, position side out of the image, candidate side with two 16-byte loads: a window of
 *      text has 10-20 such positions, so that the one-window compressor ran its 100-instruction measurement with a quarter
 *      of its lanes; here a step of text needs one or two passes instead of four;
 *   4. select: the scalar walk over the hit masks, greedy with one step of lazy evaluation (lz_match.hip.h);
 *   5. emit: the selected matches -- at most 64 a step, a match covers four positions -- are compacted into lanes, sizes
 *      and offsets come from one DPP scan, every lane writes its own sequence, literal bytes out of the image.
 *
 * Why (profiles/r04_final_pmc.json, docs/HISTORY.md 3.2): the one-window compressor spends 239 vector + 210 scalar
 * instructions and ~470 L1 lookups per 64 positions with 15-25 % of its lanes doing useful work behind the probe; the
 * vector units are 72 % busy, the L1 tag lookup rate is at its limit of one per cycle and CU. Per 256 positions this
 * file issues about half the vector instructions, a third of the scalar ones and a third of the lookups.
 *
 * Emitter concept: as in lz_match.hip.h (formats whose sequences start at byte boundaries only).
 */
#pragma once

#include "common/lz_match.hip.h"

namespace lzm {
namespace wide {

#ifndef NVCOMP_LZMW_HASH_ENTRIES
#define NVCOMP_LZMW_HASH_ENTRIES 3936
#endif
/* After this many steps in a row without a single hit (incompressible stretches) only every fourth position's candidate
 * is looked at -- a match found up to three positions late gets those bytes back by its growth backwards -- until a step
 * has a hit again; 0 = never. Every position still goes into the table. Why: a candidate look is one scattered 12-byte
 * load, and on noise every position has a (colliding) candidate: 1 GiB of noise asks the fabric for 10^9 lines. */
#ifndef NVCOMP_LZMW_FIRST_BATCH_EARLY
#define NVCOMP_LZMW_FIRST_BATCH_EARLY 1
#endif
#ifndef NVCOMP_LZMW_QUIET_STEPS
#define NVCOMP_LZMW_QUIET_STEPS 2
#endif
constexpr uint32_t kEntries = NVCOMP_LZMW_HASH_ENTRIES; /* 2-byte entries: position mod 65536 */
static_assert(kEntries % 2 == 0 && kEntries <= 65536, "the table is cleared a dword at a time");
constexpr uint32_t kSub = 4;             /* sub-windows of 64 positions per step */
constexpr uint32_t kStep = 64 * kSub;    /* positions per step */
constexpr uint32_t kBlock = 256;         /* bytes of an image block: one dword per lane */
constexpr uint32_t kRing = 4 * kBlock;   /* the image holds blocks b-1 ... b+2 of the step in block b, byte p at p & 1023 */
constexpr uint32_t kMirror = 48;         /* the first bytes of slot 0 once more behind slot 3: reads run on ascending addresses */
constexpr uint32_t kImage = kRing + kMirror;
constexpr uint32_t kQueue = 4 * kStep;   /* hit queue: (position in the step) | distance << 8; afterwards the sequence slots */
constexpr uint32_t kResults = kStep;     /* one byte per position: the match length of a hit that was measured through the queue */
constexpr uint32_t kScratch = kQueue + kResults;
constexpr uint32_t kCap = 24;            /* per-lane match measurement (longer matches: the whole wave, lzm::extend_match) */
constexpr uint32_t kBack = 4;            /* bytes a match may grow backwards over its literal run: what the word check brings */
constexpr uint32_t kLdsPerWave = 2 * kEntries + kImage + kScratch;

__device__ __forceinline__ uint32_t hashw(uint32_t v)
{
  const uint32_t h = v * 2654435761u;
  if constexpr ((kEntries & (kEntries - 1)) == 0) {
    return h >> (32 - __builtin_ctz(kEntries));
  } else {
    /* the next power of two's worth of hash bits, folded: the first slots take two values each. (A multiply-shift range
     * reduction is a second quarter-rate multiplication per position.) */
    constexpr uint32_t kBits = 32 - __builtin_clz(kEntries - 1);
    static_assert(2 * kEntries >= (1u << kBits), "one fold");
    const uint32_t x = h >> (32 - kBits);
    const uint32_t folded = x - kEntries; /* wraps to a huge value below kEntries */
    return x < folded ? x : folded;
  }
}

/* The input image: a ring of four 256-byte blocks, blocks [lo, hi) resident. */
struct Image
{
  uint8_t* base;
  uint32_t lo, hi;
  uint32_t coming; /* block requested ahead (its dwords wait in `piece`), or ~0u */
  uint32_t piece;
};

__device__ __forceinline__ uint32_t image_fetch(const uint8_t* __restrict__ src, uint32_t n, uint32_t blk)
{
  const uint32_t at = blk * kBlock + 4 * (uint32_t)wave::lane_id();
  uint32_t v = 0;
  if (at + 4 <= n) {
    v = wave::gload_u32(src + at);
  } else if (at < n) { /* the chunk's last one to three bytes */
    v = wave::gload_u8(src + at);
    if (at + 1 < n) {
      v |= wave::gload_u8(src + at + 1) << 8;
    }
    if (at + 2 < n) {
      v |= wave::gload_u8(src + at + 2) << 16;
    }
  }
  return v;
}

__device__ __forceinline__ void image_commit(uint8_t* base, uint32_t blk, uint32_t piece)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  const uint32_t slot = blk & 3u;
  *(uint32_t*)(base + slot * kBlock + 4 * lane) = piece;
  if (slot == 0 && lane < kMirror / 4) {
    *(uint32_t*)(base + kRing + 4 * lane) = piece;
  }
}

/* Blocks b - 1 (when there is one), b and b + 1 resident; b + 2 requested for the next step. */
__device__ __forceinline__ void image_ensure(Image& im, const uint8_t* __restrict__ src, uint32_t n, uint32_t b)
{
  const uint32_t need_lo = b ? b - 1 : 0u;
  const uint32_t need_hi = b + 2;
  wave::sync_wave(); /* the step before has read what is overwritten now */
  if (need_lo < im.lo || need_lo > im.hi) { /* behind a long match: nothing resident is of use */
    im.lo = need_lo;
    im.hi = need_lo;
  }
  while (im.hi < need_hi) {
    const uint32_t p = im.coming == im.hi ? im.piece : image_fetch(src, n, im.hi);
    image_commit(im.base, im.hi, p);
    ++im.hi;
  }
  im.lo = im.hi - im.lo > 4 ? im.hi - 4 : im.lo;
  im.coming = ~0u;
  wave::sync_wave();
  if ((b + 2) * kBlock < n) {
    im.piece = image_fetch(src, n, b + 2);
    im.coming = b + 2;
  }
}

/* The four bytes at chunk position p (resident). */
__device__ __forceinline__ uint32_t image_u32(const uint8_t* base, uint32_t p)
{
  const uint32_t o = p & (kRing - 1);
  const uint32_t* q = (const uint32_t*)(base + (o & ~3u));
  return wave::align_bytes(q[1], q[0], o & 3u);
}

/* Twelve bytes from chunk position p on (resident): four aligned dwords. */
__device__ __forceinline__ wave::u32x3 image_u96(const uint8_t* base, uint32_t p)
{
  const uint32_t o = p & (kRing - 1);
  const uint32_t* q = (const uint32_t*)(base + (o & ~3u));
  const uint32_t q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
  wave::u32x3 r = {wave::align_bytes(q1, q0, o & 3u), wave::align_bytes(q2, q1, o & 3u), wave::align_bytes(q3, q2, o & 3u)};
  return r;
}

/* The candidate side of a measurement, requested: the position and its candidate agree in their first 8 bytes (the word
 * check brought those, and the growth backwards); what is still open are bytes [c + 8, c + 24): one 16-byte load. A
 * candidate in the last 24 bytes of the chunk (rare) is read dword by dword, as far as the chunk goes, when the measurement
 * is finished. */
struct Pending
{
  wave::u32x4 b;
  uint32_t p, c;
  bool active, whole;
};

__device__ __forceinline__ Pending measure_describe(uint32_t n, uint32_t p, uint32_t c, bool active)
{
  Pending m;
  m.p = p, m.c = c, m.active = active;
  m.whole = active && c + kCap <= n;
  return m;
}

__device__ __forceinline__ Pending measure_request(const uint8_t* __restrict__ src, uint32_t n, uint32_t p, uint32_t c, bool active)
{
  Pending m = measure_describe(n, p, c, active);
  /* a lane that has nothing to measure (or goes the slow way) reads the chunk's first bytes: one address for all of them */
  m.b.x = m.b.y = m.b.z = m.b.w = 0;
  if (n >= 16) {
    m.b = wave::gload_u32x4(src + (m.whole ? c + 8 : 0u));
  }
  return m;
}

/* The first dword in which f and g differ (4: none) and the difference. */
__device__ __forceinline__ void first_difference(
    const uint32_t (&f)[4], uint32_t g0, uint32_t g1, uint32_t g2, uint32_t g3, uint32_t& idx, uint32_t& xv)
{
  const uint32_t x[4] = {f[0] ^ g0, f[1] ^ g1, f[2] ^ g2, f[3] ^ g3};
  idx = 4, xv = 0;
#pragma unroll
  for (int i = 3; i >= 0; --i) {
    idx = x[i] ? (uint32_t)i : idx;
    xv = x[i] ? x[i] : xv;
  }
}

/* What the measuring lane finds for position p and candidate c (c < p, their first 8 bytes are equal or the lane is idle):
 * the match length, 8 ... kCap and at most match_end - p. */
__device__ __forceinline__ uint32_t measure_finish(
    const uint8_t* __restrict__ src, uint32_t n, const uint8_t* img, const Pending& m, uint32_t match_end)
{
  const uint32_t p = m.p, c = m.c;
  /* position side: bytes [p + 8, p + 24) out of the image, five aligned dwords */
  uint32_t f[4];
  {
    const uint32_t sh = p & 3u;
    const uint32_t* r = (const uint32_t*)(img + (((p + 8u) & (kRing - 1)) & ~3u));
    uint32_t v[5];
#pragma unroll
    for (uint32_t i = 0; i < 5; ++i) {
      v[i] = r[i];
    }
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) {
      f[i] = wave::align_bytes(v[i + 1], v[i], sh);
    }
  }
  uint32_t idx, xv;
  first_difference(f, m.b.x, m.b.y, m.b.z, m.b.w, idx, xv);
  if (m.active && !m.whole) {
    /* (the loads AND what is done with them stay inside the branch: a use behind the join would make the common path wait
     * for every load in flight, the next step's candidate loads included) */
    uint32_t g[4];
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) {
      g[i] = ~f[i];
      if (c + 8 + 4 * i + 4 <= n) {
        g[i] = wave::gload_u32(src + c + 8 + 4 * i);
      }
    }
    first_difference(f, g[0], g[1], g[2], g[3], idx, xv);
  }
  uint32_t mlen = 8 + 4 * idx + (xv ? (uint32_t)__builtin_ctz(xv) >> 3 : 0u);
  const uint32_t room = match_end - p;
  mlen = mlen < room ? mlen : room;
  return m.active ? mlen : 0u;
}

/* One sub-window's part of the selection walk: greedy over `eff`, the hit lanes that lazy evaluation does not pass over
 * (decided per lane in front of the walk, so that a step of it is: find the lane, read its length, mask). `cur` =
 * position in the step from which the next match may start (carried from sub-window to sub-window); returns the mask of
 * selected lanes; a match that hit the per-lane cap is measured by the whole wave and its length written back. */
__device__ __forceinline__ uint64_t select_sub(
    const uint8_t* __restrict__ src, uint32_t ip, uint32_t k, uint64_t eff, uint64_t capped, uint32_t& mlen, uint32_t cand,
    uint32_t match_end, uint32_t& cur)
{
  const uint32_t base = 64 * k;
  const uint32_t start = cur <= base ? 0u : cur - base;
  if (start >= 64 || (eff >> start) == 0) {
    return 0;
  }
  if ((eff & capped) == 0) { /* the common case: every length is below 64, the walk is eight scalar instructions a match */
    uint64_t selected;
    uint32_t end;
    wave::select_walk(eff, mlen, start, selected, end);
    cur = base + end;
    return selected;
  }
  uint64_t selected = 0;
  uint64_t rest = eff & (~0ull << start);
  while (rest) {
    const uint32_t f = wave::ctz64(rest);
    uint32_t flen = wave::read_lane(mlen, f);
    if (flen >= kCap) {
      flen = extend_match(src, ip + base + f, wave::read_lane(cand, f), kCap, match_end);
      mlen = wave::write_lane(mlen, flen, f);
    }
    selected |= 1ull << f;
    const uint32_t next_lane = f + flen;
    cur = base + next_lane;
    rest = next_lane < 64 ? (eff & (~0ull << next_lane)) : 0ull;
  }
  return selected;
}

/* What the probe of a step leaves behind: per sub-window the lane's candidate, whether there is one, and the candidate's
 * bytes [c - 4, c + 8) -- on their way: the loads are not waited for here. */
struct Probed
{
  uint32_t cand[kSub];
  wave::u32x3 cbytes[kSub]; /* the candidate's bytes [c - 4, c + 8) */
  uint32_t has; /* bit k: sub-window k has a candidate */
};

template <class Emitter>
__device__ __forceinline__ void probe_step(
    Probed& pr, Image& im, const uint8_t* __restrict__ src, uint32_t n, uint16_t* table, uint32_t ip, uint32_t last_start,
    uint32_t look_limit)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  uint8_t* image = im.base;
  image_ensure(im, src, n, ip / kBlock);
  /* the word at the position, and whether it stands 1, 2, 4 or 8 bytes back as well (runs, typed columns): the twelve bytes
   * from 8 before the position on come out of the image as four aligned dwords, the five words are byte alignments of
   * neighbouring pairs. (Until the middle of round 5 the repeats were found by comparing with the neighbouring lanes'
   * words: seven or eight cross-lane fetches per sub-window through the LDS crossbar.) */
  uint32_t slot[kSub], near[kSub];
#pragma unroll
  for (uint32_t k = 0; k < kSub; ++k) {
    const uint32_t pos = ip + 64 * k + lane;
    const uint32_t o = (pos - 8u) & (kRing - 1);
    const uint32_t* q = (const uint32_t*)(image + (o & ~3u));
    const uint32_t qa = q[0], qb = q[1], qc = q[2], qd = q[3]; /* bytes [A - 8, A + 8), A = the position's dword */
    const uint32_t sh = pos & 3u;
    const uint32_t w = wave::align_bytes(qd, qc, sh);
    const uint32_t w4 = wave::align_bytes(qc, qb, sh), w8 = wave::align_bytes(qb, qa, sh);
    /* (bytes [p - 1, p + 3) and [p - 2, p + 2) are alignments of the two words already at hand, by constants) */
    const uint32_t w1 = wave::align_bytes(w, w4, 3u), w2 = wave::align_bytes(w, w4, 2u);
    uint32_t d = w8 == w ? 8u : 0u;
    d = w4 == w ? 4u : d;
    d = w2 == w ? 2u : d;
    d = w1 == w ? 1u : d;
    near[k] = pos >= 8 ? d : 0u; /* (the chunk's first eight positions have nothing 8 bytes back) */
    slot[k] = hashw(w);
  }
  /* table entry (position mod 65 536) -> candidate = the nearest position below with these low bits, then this sub-window's
   * positions go in: the LDS serves a wave's accesses in issue order, so that the next sub-window's probe sees them without
   * a wait. dist = 0: the entry is the position's own low bits (an entry of 65 536 bytes ago, or the table's zeros at
   * position 0); a candidate below the chunk's start comes out huge and fails the range test below. */
  uint32_t dist[kSub];
#pragma unroll
  for (uint32_t k = 0; k < kSub; ++k) {
    const uint32_t pos = ip + 64 * k + lane;
    dist[k] = (pos - table[slot[k]]) & 0xffffu;
    wave::sync_wave();
    if (pos <= last_start) {
      table[slot[k]] = (uint16_t)pos;
    }
    wave::sync_wave();
  }
  /* the candidate's bytes [c - 4, c + 8) lie inside the chunk: 4 <= c <= n - 8, one unsigned compare of c - 4 */
  const uint32_t span = n >= 12 ? n - 11u : 0u;
  pr.has = 0;
#pragma unroll
  for (uint32_t k = 0; k < kSub; ++k) {
    const uint32_t pos = ip + 64 * k + lane;
    const uint32_t back_by = near[k] ? near[k] : dist[k];
    const bool found = (dist[k] - 1u < Emitter::kReach) | (near[k] != 0);
    pr.cand[k] = pos - back_by;
    const uint32_t from = pr.cand[k] - 4u;
    /* (look_limit: last_start, or less for the lanes that a quiet stretch leaves out) */
    const bool has = found & (pos <= look_limit) & (from < span);
    pr.has |= has ? 1u << k : 0u;
    /* word check: the candidate's bytes [c - 4, c + 8), ONE load per position with a candidate, the four loads of a lane
     * travel together (a lane without a candidate reads the chunk's first bytes: one address for all of them). Three
     * candidates in four are hash collisions and stop here; half of the rest end inside these bytes and are measured by
     * them, growth backwards included */
    pr.cbytes[k] = wave::gload_u32x3(src + (has ? from : 0u));
  }
}

template <class Emitter>
__device__ __forceinline__ uint32_t encode_chunk(
    const uint8_t* __restrict__ src, uint32_t n, uint8_t* dst, uint16_t* table, uint8_t* image, uint8_t* scratch,
    uint32_t last_start, uint32_t match_end, bool any_match)
{
  static_assert(!Emitter::kStream, "byte-aligned formats only");
  const uint32_t lane = (uint32_t)wave::lane_id();
  for (uint32_t i = lane; i < kEntries / 2; i += 64) {
    ((uint32_t*)table)[i] = 0;
  }
  uint32_t* queue = (uint32_t*)scratch;
  uint8_t* results = scratch + kQueue;
  wave::sync_wave();

  uint32_t op = 0;
  uint32_t anchor = 0;
  LZM_PROF_DECL;
  if (any_match) {
    Image im;
    im.base = image;
    im.lo = 0, im.hi = 0, im.coming = ~0u, im.piece = 0;
    uint32_t ip = 0;   /* multiple of kStep */
    uint32_t skip = 0; /* leading positions of the step that the last match already covers */
    uint32_t hitless = 0; /* steps in a row without a hit */
    Probed pr;
    probe_step<Emitter>(pr, im, src, n, table, ip, last_start, last_start);
    while (ip <= last_start) {
      LZM_T(0);
      /* ---- the step's hits: the candidates that hold the position's word (the loads were requested a step ago). A hit
       * that ends within its first eight bytes is measured by that; the others wait for the queue ---- */
      uint32_t cand[kSub], mlen[kSub];
      uint32_t back = 0; /* four bits a sub-window: the bytes a hit may grow backwards */
      bool ok[kSub], longer[kSub];
      uint64_t hits[kSub];
      uint32_t total_hits = 0, total_long = 0;
#pragma unroll
      for (uint32_t k = 0; k < kSub; ++k) {
        const uint32_t rel = 64 * k + lane;
        cand[k] = pr.cand[k];
        const wave::u32x3 mine = image_u96(image, ip + rel - 4);
        const uint32_t xb = mine.x ^ pr.cbytes[k].x, xw = mine.y ^ pr.cbytes[k].y, xf = mine.z ^ pr.cbytes[k].z;
        ok[k] = ((pr.has >> k) & 1u) != 0 && rel >= skip && xw == 0;
        const uint32_t room = match_end - (ip + rel);
        uint32_t len = xf ? 4 + ((uint32_t)__builtin_ctz(xf) >> 3) : 8u;
        len = len < room ? len : room;
        longer[k] = ok[k] && xf == 0 && room > 8;
        const uint32_t grow = xb ? (uint32_t)__builtin_clz(xb) >> 3 : 4u;
        /* (Matches of 4 bytes are kept: dropping what is shorter than 5 after its growth backwards would take 17 % of the
         * sequences out of the stream -- 6 671 -> 5 516 a chunk, HC-12 writes 6 035 -- and 1.2 % off the ratio, 1.8549 ->
         * 1.8320: measured with the emulator, not shipped.) */
        mlen[k] = ok[k] ? len : 0u;
        back |= grow << (4 * k);
        hits[k] = wave::ballot(ok[k]);
        total_hits += wave::popc64(hits[k]);
      }
      LZM_T(2);
      if (total_hits == 0) {
        skip = skip > kStep ? skip - kStep : 0;
        ip += kStep;
        ++hitless;
        if (ip <= last_start) {
          const bool quiet = NVCOMP_LZMW_QUIET_STEPS != 0 && hitless >= NVCOMP_LZMW_QUIET_STEPS;
          probe_step<Emitter>(pr, im, src, n, table, ip, last_start, quiet && (lane & 3u) != 0 ? 0u : last_start);
        }
        LZM_T(1);
        continue;
      }
      hitless = 0;

      /* ---- long first match (runs, periodic columns): one cooperative probe decides (lz_match.hip.h) ---- */
      {
        const uint32_t k0 = skip >> 6; /* the first sub-window with searched positions */
        const uint64_t h0 = k0 == 0 ? hits[0] : k0 == 1 ? hits[1] : k0 == 2 ? hits[2] : hits[3];
        if (wave::popc64(h0) >= kDenseHits) {
          const uint32_t c0 = k0 == 0 ? cand[0] : k0 == 1 ? cand[1] : k0 == 2 ? cand[2] : cand[3];
          const uint32_t f0 = wave::ctz64(h0);
          uint32_t mpos = ip + 64 * k0 + f0;
          uint32_t mcand = wave::read_lane(c0, f0);
          const uint32_t p = mpos + kMinMatch + lane;
          const bool same = p < match_end && wave::gload_u8(src + p) == wave::gload_u8(src + mcand + kMinMatch + lane);
          const uint64_t diff = ~wave::ballot(same);
          uint32_t len0 = diff ? kMinMatch + wave::ctz64(diff) : extend_match(src, mpos, mcand, kMinMatch + 64, match_end);
          if (len0 >= kCap) {
            const uint32_t room = mpos - anchor < mcand ? mpos - anchor : mcand;
            const bool eq = lane < room && wave::gload_u8(src + mpos - 1 - lane) == wave::gload_u8(src + mcand - 1 - lane);
            const uint64_t ne = ~wave::ballot(eq);
            const uint32_t back = ne ? wave::ctz64(ne) : 64u;
            mpos -= back;
            mcand -= back;
            len0 += back;
            op += Emitter::match(dst + op, src + anchor, mpos - anchor, mpos - mcand, len0);
            anchor = mpos + len0;
            /* Runs and sorted or periodic columns go on at the SAME distance a few bytes further on (the bytes that
             * changed): look for that with the whole wave -- the first of the next 64 positions whose word stands
             * `offset` bytes back as well -- before a step of 256 positions is probed for it. (Sorted-key column:
             * one match of 170-680 bytes per step otherwise.) */
            const uint32_t offset = mpos - mcand;
            for (;;) {
              const uint32_t q = anchor + lane;
              const bool same_word = q <= last_start && wave::gload_u32(src + q) == wave::gload_u32(src + q - offset);
              const uint64_t at = wave::ballot(same_word);
              if (at == 0) {
                break;
              }
              const uint32_t rpos = anchor + wave::ctz64(at);
              const uint32_t rlen = extend_match(src, rpos, rpos - offset, kMinMatch, match_end);
              if (rlen < kCap) {
                break;
              }
              op += Emitter::match(dst + op, src + anchor, rpos - anchor, offset, rlen);
              anchor = rpos + rlen;
            }
            const uint32_t next = anchor;
            ip = next & ~(kStep - 1);
            skip = next - ip;
            LZM_T(3);
            if (ip <= last_start) {
              probe_step<Emitter>(pr, im, src, n, table, ip, last_start, last_start);
            }
            LZM_T(1);
            continue;
          }
        }
      }
      LZM_T(3);

      /* ---- the hits that are still open after eight bytes, compacted into a queue, are measured 64 at a time ---- */
      {
        uint32_t base = 0;
#pragma unroll
        for (uint32_t k = 0; k < kSub; ++k) {
          const uint64_t lm = wave::ballot(longer[k]);
          if (longer[k]) {
            const uint32_t rel = 64 * k + lane;
            queue[base + wave::prefix_popc(lm)] = rel | ((ip + rel - cand[k]) << 8);
          }
          base += wave::popc64(lm);
        }
        total_long = base;
        wave::sync_wave();
      }
#if NVCOMP_LZMW_FIRST_BATCH_EARLY
      /* (the lengths known so far wait in LDS, beside the ones the measurements will write: four registers less across the
       * probe, whose spills would be reloaded -- and waited for, with everything else in flight -- on the way) */
#pragma unroll
      for (uint32_t k = 0; k < kSub; ++k) {
        results[64 * k + lane] = (uint8_t)mlen[k];
      }
      wave::sync_wave();
      /* The first 64 of them are asked for in FRONT of the next step's probe and looked at behind it. The vector-memory
       * counter runs in issue order: what was asked for earlier can be waited for while what was asked for later -- the
       * probe's candidate loads -- stays in flight, not the other way round. (Only the 16 loaded bytes live across the
       * probe; the rest is read from the queue again.) */
      wave::u32x4 first_bytes = {0, 0, 0, 0};
      if (total_long != 0) {
        const bool act = lane < total_long;
        const uint32_t e = act ? queue[lane] : 8u << 8;
        const uint32_t p = act ? ip + (e & 255u) : ip + 8;
        first_bytes = measure_request(src, n, p, p - (e >> 8), act).b;
      }
#endif
      LZM_T(4);

      /* ---- the NEXT step's probe and word-check loads: they travel while this step is measured, selected and written.
       * (The step behind a long match starts elsewhere: then this probe only put positions inside that match into the
       * table.) `pr` is free: what this step needs of it are the candidates and the hit masks ---- */
      const uint32_t nip = ip + kStep;
#if NVCOMP_LZMW_FIRST_BATCH_EARLY
      const auto finish_first = [&]() {
        if (total_long != 0) {
          const bool act = lane < total_long;
          const uint32_t e = act ? queue[lane] : 8u << 8;
          const uint32_t rel = e & 255u;
          const uint32_t p = act ? ip + rel : ip + 8;
          Pending m = measure_describe(n, p, p - (e >> 8), act);
          m.b = first_bytes;
          const uint32_t r = measure_finish(src, n, image, m, match_end);
          if (act) {
            results[rel] = (uint8_t)r;
          }
        }
      };
      if (nip <= last_start) {
        probe_step<Emitter>(pr, im, src, n, table, nip, last_start, last_start);
        LZM_T(1);
        finish_first(); /* (twice in the code: behind a join with the path that has no probe the wait would be for every load) */
      } else {
        finish_first();
      }
      for (uint32_t b0 = 64; b0 < total_long; b0 += 128) {
#else
      if (nip <= last_start) {
        probe_step<Emitter>(pr, im, src, n, table, nip, last_start, last_start);
      }
      LZM_T(1);

      /* two batches a round: their loads travel together */
      for (uint32_t b0 = 0; b0 < total_long; b0 += 128) {
#endif
        const bool act0 = b0 + lane < total_long, act1 = b0 + 64 + lane < total_long;
        const uint32_t e0 = act0 ? queue[b0 + lane] : 8u << 8;
        const uint32_t e1 = act1 ? queue[b0 + 64 + lane] : 8u << 8;
        const uint32_t rel0 = e0 & 255u, rel1 = e1 & 255u;
        const uint32_t p0 = act0 ? ip + rel0 : ip + 8, p1 = act1 ? ip + rel1 : ip + 8;
        const Pending m0 = measure_request(src, n, p0, p0 - (e0 >> 8), act0);
        uint32_t r1 = 0;
        if (b0 + 64 < total_long) {
          const Pending m1 = measure_request(src, n, p1, p1 - (e1 >> 8), act1);
          r1 = measure_finish(src, n, image, m1, match_end);
        }
        const uint32_t r0 = measure_finish(src, n, image, m0, match_end);
        if (act0) {
          results[rel0] = (uint8_t)r0;
        }
        if (act1) {
          results[rel1] = (uint8_t)r1;
        }
      }
      wave::sync_wave();
      uint64_t capped[kSub];
#pragma unroll
      for (uint32_t k = 0; k < kSub; ++k) {
#if NVCOMP_LZMW_FIRST_BATCH_EARLY
        mlen[k] = results[64 * k + lane];
#else
        if (longer[k]) {
          mlen[k] = results[64 * k + lane];
        }
#endif
        hits[k] = wave::ballot(mlen[k] != 0);
        capped[k] = wave::ballot(mlen[k] >= kCap);
      }
      LZM_T(5);

      /* ---- select ---- */
      /* lazy evaluation, per lane: a match shorter than the one starting at the next position is passed over (the next
       * position's is looked at in its turn) */
#if NVCOMP_LZM_LAZY
#pragma unroll
      for (uint32_t k = 0; k < kSub; ++k) {
        uint32_t ahead = wave::next_lane(mlen[k]);
        if (k + 1 < kSub) {
          const uint32_t first = wave::read_lane(mlen[k + 1], 0);
          ahead = lane == 63 ? first : ahead;
        }
        hits[k] = wave::ballot(mlen[k] != 0 && !(mlen[k] < kCap && ahead > mlen[k]));
      }
#endif
      uint32_t cur = skip;
      uint64_t sel[kSub];
#pragma unroll
      for (uint32_t k = 0; k < kSub; ++k) {
        sel[k] = select_sub(src, ip, k, hits[k], capped[k], mlen[k], cand[k], match_end, cur);
      }
      const uint32_t n_sel = wave::popc64(sel[0]) + wave::popc64(sel[1]) + wave::popc64(sel[2]) + wave::popc64(sel[3]);
      LZM_T(6);
      if (n_sel != 0) {
        /* ---- 5. the selected matches, compacted into lanes ---- */
        wave::sync_wave(); /* the queue has been read */
        {
          uint32_t base = 0;
#pragma unroll
          for (uint32_t k = 0; k < kSub; ++k) {
            if ((sel[k] >> lane) & 1) {
              const uint32_t rel = 64 * k + lane;
              const uint32_t at = base + wave::prefix_popc(sel[k]);
              queue[2 * at] = rel | (((back >> (4 * k)) & 7u) << 8) | ((ip + rel - cand[k]) << 16);
              queue[2 * at + 1] = mlen[k];
            }
            base += wave::popc64(sel[k]);
          }
        }
        wave::sync_wave();
        const bool mine = lane < n_sel;
        const uint32_t s0 = mine ? queue[2 * lane] : 0u;
        const uint32_t pos = ip + (s0 & 255u);
        uint32_t my_len = mine ? queue[2 * lane + 1] : 0u;
        const uint32_t offset = s0 >> 16;
        /* a lane's literal run starts where the match of the lane below ends (the first one's: at the anchor) */
        uint32_t prev_end = wave::prev_lane(pos + my_len);
        prev_end = lane == 0 ? anchor : prev_end;
        uint32_t lit_len = mine ? pos - prev_end : 0u;
        {
          const uint32_t grow = (s0 >> 8) & 7u;
          const uint32_t b = grow < lit_len ? grow : lit_len; /* never into the match before */
          lit_len -= mine ? b : 0u;
          my_len += mine ? b : 0u;
        }
        const uint32_t size = mine ? Emitter::seq_size(lit_len, my_len, offset) : 0u;
        const uint32_t incl = wave::scan_add_inclusive(size);
        const uint32_t total = wave::read_lane(incl, 63);
        const bool small = mine && Emitter::is_small(lit_len, my_len);
        uint64_t big = wave::ballot(mine && !small);
        uint8_t* my_dst = dst + op + incl - size;
        if (small) {
          Emitter::emit_small_header(my_dst, lit_len, offset, my_len);
        }
        /* literal runs of the small sequences (<= 64 bytes, ending inside this step: always in the image) */
        {
          uint8_t* ld = my_dst + Emitter::lit_offset(lit_len);
          const bool lit4 = small && lit_len >= 4;
          for (uint32_t base4 = 0; wave::ballot(lit4 && lit_len > base4) != 0; base4 += 16) {
            if (lit4 && lit_len > base4) {
              const uint32_t lastoff = lit_len - 4;
              uint32_t v[4];
#pragma unroll
              for (uint32_t i = 0; i < 4; ++i) {
                const uint32_t o = base4 + 4 * i < lastoff ? base4 + 4 * i : lastoff;
                v[i] = image_u32(image, prev_end + o);
              }
#pragma unroll
              for (uint32_t i = 0; i < 4; ++i) {
                const uint32_t o = base4 + 4 * i < lastoff ? base4 + 4 * i : lastoff;
                lz::st_u32(ld + o, v[i]);
              }
            }
          }
          if (small && lit_len != 0 && lit_len < 4) {
            const uint32_t v = image_u32(image, prev_end);
            ld[0] = (uint8_t)v;
            if (lit_len > 1) {
              ld[1] = (uint8_t)(v >> 8);
            }
            if (lit_len > 2) {
              ld[2] = (uint8_t)(v >> 16);
            }
          }
        }
        LZM_T(7);
        /* the few sequences a single lane cannot write (long literal run after match-less steps, long match) */
        while (big) {
          const uint32_t j = wave::ctz64(big);
          big &= big - 1;
          const uint32_t jdst = op + wave::read_lane(incl - size, j);
          const uint32_t jlit = wave::read_lane(prev_end, j);
          Emitter::match(dst + jdst, src + jlit, wave::read_lane(lit_len, j), wave::read_lane(offset, j), wave::read_lane(my_len, j));
        }
        op += total;
        anchor = ip + cur;
      }
      LZM_T(8);
      if (cur >= 2 * kStep) { /* a long match measured by the whole wave: go on behind it */
        const uint32_t next = ip + cur;
        ip = next & ~(kStep - 1);
        skip = next - ip;
        if (ip <= last_start) {
          probe_step<Emitter>(pr, im, src, n, table, ip, last_start, last_start);
        }
        LZM_T(1);
      } else {
        ip = nip;
        skip = cur > kStep ? cur - kStep : 0;
      }
    }
  }
  op += Emitter::tail(dst + op, src + anchor, n - anchor);
  LZM_T(9);
  LZM_PROF_END;
  return op;
}

} // namespace wide
} // namespace lzm
