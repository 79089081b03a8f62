/*
 * snappy/snappy_decode.hip.h -- batched Snappy raw-format decoder for gfx950.
 *
 * Replaces the device side of nvcompBatchedSnappyDecompressAsync and
 * nvcompBatchedSnappyGetDecompressSizeAsync (reference call sites:
 * benchmarks/benchmark_snappy_synth.cpp:241-251,
 * benchmarks/benchmark_template_chunked.cuh:520-530). One wavefront per chunk,
 * same execution model as LZ4 (common/lz_common.hip.h): every Snappy element
 * (literal or copy) is one "sequence" with either a literal part or a match part.
 *
 * Raw format (public Snappy format description; checked against libsnappy):
 *   varint32 uncompressed length, then elements tagged by (tag & 3):
 *   0 literal (len-1 = tag>>2, or in the next 1..4 bytes when tag>>2 is 60..63),
 *   1 copy with 11-bit offset (len 4..11), 2 copy with 16-bit offset (len 1..64),
 *   3 copy with 32-bit offset (len 1..64). All four kinds are decoded, whatever
 *   the producing compressor chose to emit (CHANGELOG.md:182-184).
 */
#pragma once

#include "common/lz_common.hip.h"

namespace snappy {

/* Distance from a tag to the next tag when it does not depend on extra length
 * bytes, else 0; one byte per tag candidate in the dword. */
__device__ __forceinline__ uint32_t fast_deltas(uint32_t cw)
{
  uint32_t out = 0;
#pragma unroll
  for (uint32_t j = 0; j < 4; ++j) {
    const uint32_t t = (cw >> (8 * j)) & 0xffu;
    const uint32_t kind = t & 3u;
    const uint32_t n = t >> 2;
    uint32_t d;
    if (kind == 0) {
      d = n < 60 ? n + 2 : 0; /* tag + (n+1) literal bytes */
    } else {
      d = kind == 3 ? 5u : kind + 1; /* copy-1: 2, copy-2: 3, copy-4: 5 */
    }
    out |= d << (8 * j);
  }
  return out;
}

struct Chase
{
  lz::InWindow w;
  uint32_t dv;
  uint32_t q;
};

__device__ __forceinline__ void chase_reload(Chase& c, uint32_t q)
{
  lz::window_load(c.w, q);
  c.dv = fast_deltas(c.w.cw);
}

__device__ __forceinline__ uint32_t chase_byte(Chase& c, uint32_t pos)
{
  if (!lz::window_has(c.w, pos)) {
    chase_reload(c, pos);
  }
  return lz::window_byte(c.w, pos);
}

/* Parse the varint32 preamble (uniform). Returns false if malformed. */
__device__ __forceinline__ bool read_preamble(Chase& c, uint32_t& total)
{
  uint32_t v = 0;
  for (uint32_t shift = 0; shift <= 28; shift += 7) {
    if (c.q >= c.w.vend) {
      return false;
    }
    const uint32_t b = chase_byte(c, c.q);
    ++c.q;
    v |= (b & 127u) << shift;
    if (!(b & 128u)) {
      if (shift == 28 && b > 15) {
        return false;
      }
      total = v;
      return true;
    }
  }
  return false;
}

/* Next tag position after a long literal (tag>>2 >= 60); >= vend ends the chase. */
__device__ __forceinline__ uint32_t chase_slow_next(Chase& c)
{
  const uint32_t vend = c.w.vend;
  const uint32_t t = chase_byte(c, c.q);
  const uint32_t nb = (t >> 2) - 59;
  uint32_t pos = c.q + 1;
  if (vend - pos < nb) {
    return vend + 1;
  }
  uint32_t len = 0;
  for (uint32_t i = 0; i < nb; ++i) {
    len |= chase_byte(c, pos + i) << (8 * i);
  }
  pos += nb;
  if (len >= vend - pos) { /* len+1 bytes must fit */
    return vend + 1;
  }
  return pos + len + 1;
}

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

/* Lane-parallel decode of the element whose tag is at in[p]. */
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
  const uint32_t kind = t & 3u;
  uint32_t pos = p + 1;
  const uint32_t avail = in_len - pos;
  if (kind == 0) {
    uint32_t len = t >> 2;
    if (len >= 60) {
      const uint32_t nb = len - 59;
      if (avail < nb) {
        bad = true;
        return;
      }
      len = 0;
      for (uint32_t i = 0; i < nb; ++i) {
        len |= (uint32_t)in[pos + i] << (8 * i);
      }
      pos += nb;
    }
    if (len >= in_len - pos) { /* len+1 literal bytes must fit */
      bad = true;
      return;
    }
    s.lit_src = pos;
    s.lit_len = len + 1;
    return;
  }
  const uint32_t need = kind == 3 ? 4u : kind;
  if (avail < need) {
    bad = true;
    return;
  }
  if (kind == 1) {
    s.match_len = 4 + ((t >> 2) & 7u);
    s.match_off = ((t >> 5) << 8) | in[pos];
  } else if (kind == 2) {
    s.match_len = 1 + (t >> 2);
    s.match_off = (uint32_t)in[pos] | ((uint32_t)in[pos + 1] << 8);
  } else {
    s.match_len = 1 + (t >> 2);
    s.match_off = (uint32_t)in[pos] | ((uint32_t)in[pos + 1] << 8) | ((uint32_t)in[pos + 2] << 16)
                  | ((uint32_t)in[pos + 3] << 24);
  }
  if (s.match_off == 0) { /* also caught by the CHECKED offset test; keeps unchecked runs in bounds */
    bad = true;
  }
}

/* Uncompressed length from the preamble; 0 if malformed. */
__device__ __forceinline__ uint32_t decoded_size(const uint8_t* __restrict__ in, uint32_t in_len, bool& ok)
{
  Chase c;
  lz::window_init(c.w, in, in_len);
  c.q = c.w.vbeg;
  uint32_t total = 0;
  ok = false;
  if (in_len == 0) {
    return 0;
  }
  chase_reload(c, c.q);
  ok = read_preamble(c, total);
  return ok ? total : 0;
}

template <bool CHECKED, bool LANE_PARALLEL>
__device__ __forceinline__ uint32_t decode_chunk(
    const uint8_t* __restrict__ in, uint32_t in_len, uint8_t* out, uint32_t out_cap, uint32_t& err)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  err = lz::kErrNone;
  if (in_len == 0) {
    err = lz::kErrInput; /* even an empty buffer has a 1-byte preamble */
    return 0;
  }
  Chase c;
  lz::window_init(c.w, in, in_len);
  c.q = c.w.vbeg;
  chase_reload(c, c.q);
  uint32_t total = 0;
  if (!read_preamble(c, total)) {
    err = lz::kErrInput;
    return 0;
  }
  if (CHECKED && total > out_cap) {
    err = lz::kErrOutput;
    return 0;
  }
  /* the elements may not produce more than the preamble promises */
  const uint32_t limit = CHECKED ? total : out_cap;
  uint32_t op = 0;
  while (c.q < c.w.vend) {
    uint32_t seqpos = 0;
    const uint32_t count = chase(c, seqpos, LANE_PARALLEL ? 64u : 1u);
    lz::Seq s;
    bool bad;
    parse(in, in_len, seqpos - c.w.vbeg, lane < count, s, bad);
    if (wave::ballot(bad)) {
      err |= lz::kErrInput;
      return 0;
    }
    op += lz::execute_batch<CHECKED, LANE_PARALLEL>(in, in_len, out, limit, op, count, s, err);
    if (CHECKED && err) {
      return 0;
    }
  }
  if (CHECKED && (op != total || c.q != c.w.vend)) {
    err |= lz::kErrInput;
    return 0;
  }
  return op;
}

} // namespace snappy
