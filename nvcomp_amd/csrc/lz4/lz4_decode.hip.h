/*
 * lz4/lz4_decode.hip.h -- batched LZ4 block-format decoder for gfx950.
 *
 * Replaces the device side of nvcompBatchedLZ4DecompressAsync and
 * nvcompBatchedLZ4GetDecompressSizeAsync (reference call sites:
 * benchmarks/benchmark_template_chunked.cuh:520-530,
 * examples/lz4_cpu_compression.cu:121-131,
 * examples/low_level_quickstart_example.cpp:112-117). One wavefront per chunk;
 * see common/lz_common.hip.h for the execution model.
 *
 * Block format (public LZ4 specification; checked against liblz4 in tests):
 *   token: hi nibble = literal length, lo nibble = match length - 4; a nibble
 *   of 15 is extended by following bytes (each added, 255 continues); literals;
 *   2-byte little-endian offset; match-length extension. The last sequence is
 *   literals only and ends exactly at the end of the chunk.
 */
#pragma once

#include "common/lz_common.hip.h"

namespace lz4 {

/* For each of the 4 token candidates in a dword: distance to the next token if
 * neither nibble needs extension bytes (1 token + L literals + 2 offset), else 0. */
__device__ __forceinline__ uint32_t fast_deltas(uint32_t cw)
{
  const uint32_t hi = (cw >> 4) & 0x0f0f0f0fu;
  const uint32_t lo = cw & 0x0f0f0f0fu;
  const uint32_t ext = (((hi + 0x01010101u) | (lo + 0x01010101u)) >> 4) & 0x01010101u;
  return (hi + 0x03030303u) & ~(ext * 0xffu);
}

struct Chase
{
  lz::InWindow w;
  uint32_t dv; /* fast_deltas of the window */
  uint32_t q;  /* virtual position of the next token (uniform) */
};

__device__ __forceinline__ void chase_reload(Chase& c, uint32_t q)
{
  lz::window_load(c.w, q);
  c.dv = fast_deltas(c.w.cw);
}

__device__ __forceinline__ void chase_init(Chase& c, const uint8_t* in, uint32_t in_len)
{
  lz::window_init(c.w, in, in_len);
  c.q = c.w.vbeg;
  chase_reload(c, c.q);
}

/* Next token position for a token whose lengths use extension bytes (uniform,
 * scalar walk over the register window). Any value >= vend ends the chase;
 * the lane-parallel parse decides whether that is the legal end of the block. */
__device__ __forceinline__ uint32_t chase_slow_next(Chase& c)
{
  const uint32_t vend = c.w.vend;
  const uint32_t t = lz::window_byte(c.w, c.q);
  uint32_t pos = c.q + 1;
  uint32_t lit = t >> 4;
  if (lit == 15) {
    for (;;) {
      if (pos >= vend) {
        return vend + 1;
      }
      if (!lz::window_has(c.w, pos)) {
        chase_reload(c, pos);
      }
      const uint32_t b = lz::window_byte(c.w, pos);
      ++pos;
      lit += b;
      if (b != 255) {
        break;
      }
    }
  }
  if (lit >= vend - pos) { /* literals reach (or pass) the end: last sequence or overrun */
    return vend + 1;
  }
  pos += lit + 2;
  if ((t & 15u) == 15u) {
    for (;;) {
      if (pos >= vend) {
        return vend + 1;
      }
      if (!lz::window_has(c.w, pos)) {
        chase_reload(c, pos);
      }
      const uint32_t b = lz::window_byte(c.w, pos);
      ++pos;
      if (b != 255) {
        break;
      }
    }
  }
  return pos;
}

/* Record the virtual start positions of the next (up to 64) sequences:
 * lane k of seqpos <- start of sequence k. Returns the count (<= max_count <= 64). */
__device__ __forceinline__ uint32_t chase(Chase& c, uint32_t& seqpos, uint32_t max_count)
{
  uint32_t k = 0;
  while (k < max_count && c.q < c.w.vend) {
    if (!lz::window_has(c.w, c.q)) {
      chase_reload(c, c.q);
    }
    const uint32_t r = c.q - c.w.wb;
    const uint32_t d = (wave::read_lane(c.dv, r >> 2) >> ((r & 3u) * 8u)) & 0xffu;
    seqpos = wave::write_lane(seqpos, c.q, k);
    ++k;
    c.q = d ? c.q + d : chase_slow_next(c);
  }
  return k;
}

/* Lane-parallel field decode of the sequence whose token is at in[p]. */
__device__ __forceinline__ void parse(
    const uint8_t* __restrict__ in, uint32_t in_len, uint32_t p, bool active, lz::Seq& s, bool& bad)
{
  s.lit_src = 0;
  s.lit_len = 0;
  s.match_off = 0;
  s.match_len = 0;
  bad = false;
  if (!active) {
    return;
  }
  const uint32_t t = in[p];
  uint32_t pos = p + 1;
  uint32_t lit = t >> 4;
  if (lit == 15) {
    uint32_t b;
    do {
      if (pos >= in_len) {
        bad = true;
        return;
      }
      b = in[pos++];
      lit += b;
    } while (b == 255);
  }
  if (lit > in_len - pos) {
    bad = true;
    return;
  }
  s.lit_src = pos;
  s.lit_len = lit;
  pos += lit;
  if (pos == in_len) {
    return; /* last sequence: literals only */
  }
  if (in_len - pos < 2) {
    bad = true;
    return;
  }
  s.match_off = (uint32_t)in[pos] | ((uint32_t)in[pos + 1] << 8);
  pos += 2;
  uint32_t mlen = t & 15u;
  if (mlen == 15) {
    uint32_t b;
    do {
      if (pos >= in_len) {
        bad = true;
        return;
      }
      b = in[pos++];
      mlen += b;
    } while (b == 255);
  }
  if (mlen > 0x40000000u) {
    bad = true;
    return;
  }
  if (pos >= in_len) { /* a token must follow every match */
    bad = true;
    return;
  }
  s.match_len = mlen + 4;
}

/* Decode one chunk with the calling wave. Returns bytes produced, 0 on error. */
template <bool CHECKED, bool LANE_PARALLEL, bool SIZE_ONLY>
__device__ __forceinline__ uint32_t decode_chunk(
    const uint8_t* __restrict__ in, uint32_t in_len, uint8_t* out, uint32_t out_cap, uint32_t& err)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  uint32_t op = 0;
  err = lz::kErrNone;
  if (in_len == 0) {
    return 0;
  }
  Chase c;
  chase_init(c, in, in_len);
  while (c.q < c.w.vend) {
    uint32_t seqpos = 0;
    /* LANE_PARALLEL = false is the ablation baseline: one sequence per step,
     * every copy done by the whole wave. */
    const uint32_t count = chase(c, seqpos, LANE_PARALLEL ? 64u : 1u);
    lz::Seq s;
    bool bad;
    parse(in, in_len, seqpos - c.w.vbeg, lane < count, s, bad);
    if (wave::ballot(bad)) {
      err |= lz::kErrInput;
      return 0;
    }
    if (SIZE_ONLY) {
      /* a hostile stream may claim lengths whose 32-bit sum wraps to a small bogus size: the batch is summed in two
       * 16-bit halves (64 x 65535 fits) and the total checked against the largest chunk any decoder here accepts */
      const uint32_t len = s.lit_len + s.match_len; /* each part is at most 2^30 */
      const uint64_t batch = (uint64_t)wave::reduce_add(len & 0xffffu) + ((uint64_t)wave::reduce_add(len >> 16) << 16);
      if (batch + op > (1u << 26)) {
        err |= lz::kErrOutput;
        return 0;
      }
      op += (uint32_t)batch;
    } else {
      op += lz::execute_batch<CHECKED, LANE_PARALLEL>(in, in_len, out, out_cap, op, count, s, err);
      if (CHECKED && err) {
        return 0;
      }
    }
  }
  return op;
}

} // namespace lz4
