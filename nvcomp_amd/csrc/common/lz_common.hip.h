/*
 * common/lz_common.hip.h -- pieces shared by the LZ4 and Snappy decoders:
 * unaligned accessors, the register window over the compressed stream, the
 * wave-cooperative copies, and the sequence-parallel batch executor.
 *
 * Execution model (DESIGN.md "LZ decode"): one wavefront decodes one chunk.
 *   1. chase   - the wave walks the token chain of the compressed stream with
 *                wave-uniform (scalar) arithmetic over a 256-byte register
 *                window and records the start of up to 64 sequences;
 *   2. parse   - lane k decodes the fields of sequence k;
 *   3. scan    - a DPP prefix sum turns lengths into output positions;
 *   4. literals- every lane copies its own short literal run, long runs are
 *                copied by the whole wave;
 *   5. matches - multi-round resolution: in each round every lane whose source
 *                bytes are already final copies its match; a match that is long,
 *                self-overlapping with a period < 4 or otherwise awkward is
 *                copied by the whole wave when it becomes the oldest pending one.
 * Output goes straight to the chunk's output buffer in HBM; match sources are
 * read back through the CU's L1/L2 (one wave's vector-memory operations are
 * served in issue order, so a same-wave read-after-write needs no s_waitcnt).
 */
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common/wave.h"

namespace lz {

enum : uint32_t {
  kErrNone = 0,
  kErrInput = 1,  /* compressed stream truncated / malformed */
  kErrOutput = 2, /* output buffer too small */
  kErrOffset = 4  /* match offset 0 or beyond the produced output */
};

/* Lane-parallel thresholds: literal runs / matches up to these lengths are
 * copied by their own lane in 4-byte steps, longer ones by the whole wave. */
constexpr uint32_t kLitShort = 32;
constexpr uint32_t kMatchShort = 64;

/* ---- unaligned accessors (gfx950 global/LDS accesses may be unaligned) ---- */

__device__ __forceinline__ uint32_t ld_u32(const uint8_t* p)
{
  uint32_t v;
  __builtin_memcpy(&v, p, 4);
  return v;
}

__device__ __forceinline__ wave::u32x4 ld_u32x4(const uint8_t* p) /* any alignment */
{
  wave::u32x4 v;
  __builtin_memcpy(&v, p, 16);
  return v;
}

__device__ __forceinline__ void st_u32(uint8_t* p, uint32_t v)
{
  __builtin_memcpy(p, &v, 4);
}

__device__ __forceinline__ void st_u16(uint8_t* p, uint32_t v)
{
  const uint16_t h = (uint16_t)v;
  __builtin_memcpy(p, &h, 2);
}

struct __attribute__((packed)) Bytes16
{
  uint32_t w[4];
};

__device__ __forceinline__ void copy16(uint8_t* d, const uint8_t* s)
{
  Bytes16 t;
  __builtin_memcpy(&t, s, 16);
  __builtin_memcpy(d, &t, 16);
}

/* Store the low `rem` bytes (rem >= 1; 4 or more stores all four) of v at p. */
__device__ __forceinline__ void st_upto4(uint8_t* p, uint32_t v, uint32_t rem)
{
  if (rem >= 4) {
    st_u32(p, v);
  } else {
    if (rem & 2) {
      st_u16(p, v);
      if (rem & 1) {
        p[2] = (uint8_t)(v >> 16);
      }
    } else {
      p[0] = (uint8_t)v;
    }
  }
}

/* ---- 256-byte register window over the compressed stream ------------------
 * Lane l holds the aligned dword at "virtual position" wb + 4l, where virtual
 * position = byte index in the chunk + (chunk address & 3). A wave-uniform byte
 * is fetched with one v_readlane and scalar shifts: no memory round trip on the
 * token chain. */
struct InWindow
{
  const uint8_t* base; /* chunk pointer rounded down to 4 bytes (uniform) */
  uint32_t vbeg;       /* virtual position of the first chunk byte: chunk & 3 */
  uint32_t vend;       /* vbeg + chunk length */
  uint32_t wb;         /* virtual position of lane 0's dword (multiple of 4) */
  uint32_t cw;         /* this lane's dword */
};

__device__ __forceinline__ void window_init(InWindow& w, const uint8_t* in, uint32_t in_len)
{
  const uint32_t a = (uint32_t)((uintptr_t)in & 3u);
  w.base = in - a;
  w.vbeg = a;
  w.vend = a + in_len;
  w.wb = 0;
  w.cw = 0;
}

/* (Re)load the window so that it starts at the dword containing virtual
 * position q. Bytes outside the chunk read as zero and are never fetched. */
__device__ __forceinline__ void window_load(InWindow& w, uint32_t q)
{
  w.wb = q & ~3u;
  const uint32_t v = w.wb + 4u * (uint32_t)wave::lane_id();
  uint32_t x = 0;
  if (v >= w.vbeg && v + 4 <= w.vend) {
    x = *(const uint32_t*)(w.base + v);
  } else if (v + 4 > w.vbeg && v < w.vend) {
    for (uint32_t j = 0; j < 4; ++j) {
      if (v + j >= w.vbeg && v + j < w.vend) {
        x |= (uint32_t)w.base[v + j] << (8 * j);
      }
    }
  }
  w.cw = x;
}

__device__ __forceinline__ bool window_has(const InWindow& w, uint32_t q)
{
  return q - w.wb < 256u;
}

/* Byte at virtual position q (uniform; must be inside the window). */
__device__ __forceinline__ uint32_t window_byte(const InWindow& w, uint32_t q)
{
  const uint32_t r = q - w.wb;
  return (wave::read_lane(w.cw, r >> 2) >> ((r & 3u) * 8u)) & 0xffu;
}

/* ---- wave-cooperative copies ---------------------------------------------- */

/* dst[0,len) = src[0,len), non-overlapping, any alignment. */
__device__ __forceinline__ void wave_copy(uint8_t* dst, const uint8_t* src, uint32_t len)
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  uint32_t base = 0;
  for (; base + 1024 <= len; base += 1024) {
    copy16(dst + base + lane * 16, src + base + lane * 16);
  }
  for (; base + 256 <= len; base += 256) {
    st_u32(dst + base + lane * 4, ld_u32(src + base + lane * 4));
  }
  for (uint32_t i = base + lane; i < len; i += 64) {
    dst[i] = src[i];
  }
}

/* LZ77 match copy d[i] = d[i - off], i in [0,len), byte-serial semantics
 * (off < len replicates a pattern). The effective offset E is kept a multiple
 * of off and doubled as the pattern grows, so that a whole wave-wide step
 * (64 lanes x 16 / 4 / 1 bytes) never reads a byte the same step writes. */
__device__ __forceinline__ void wave_match_copy(uint8_t* d, uint32_t off, uint32_t len)
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  uint32_t done = 0;
  uint32_t E = off;
  while (done < len) {
    while (E < 1024 && 2 * E <= done + off) {
      E *= 2;
    }
    const uint32_t rem = len - done;
    if (E >= 1024 && rem >= 1024) {
      copy16(d + done + lane * 16, d + done + lane * 16 - E);
      done += 1024;
    } else if (E >= 256 && rem >= 256) {
      st_u32(d + done + lane * 4, ld_u32(d + done + lane * 4 - E));
      done += 256;
    } else {
      uint32_t n = E < 64 ? E : 64;
      n = n < rem ? n : rem;
      if (lane < n) {
        uint8_t* t = d + done + lane; /* pointer arithmetic: done + lane - E is negative as an index */
        *t = *(t - E);
      }
      done += n;
    }
    wave::sync();
  }
}

/* ---- per-lane sequence record --------------------------------------------- */

struct Seq
{
  uint32_t lit_src;   /* byte index of the first literal in the compressed chunk */
  uint32_t lit_len;
  uint32_t match_off; /* distance back from the match destination */
  uint32_t match_len; /* 0: no match (LZ4 last sequence / Snappy literal element) */
  /* Optional: the first literal bytes as the parser saw them, so that a short run is written without reading the
   * stream again. lit_lo = bytes 0-3, lit_hi bits 0-15 = bytes 4-5, lit_hi bits 16-18 = how many bytes are held
   * (0 = none; the run is then read from the stream). Only a run held completely is used. */
  uint32_t lit_lo = 0;
  uint32_t lit_hi = 0;
};

/*
 * Execute one batch of n (<= 64) parsed sequences; lane k owns sequence k.
 * `op` is the number of output bytes produced before the batch. Returns the
 * batch's output size; sets err bits on invalid input (CHECKED only).
 */
template <bool CHECKED, bool LANE_PARALLEL>
__device__ __forceinline__ uint32_t execute_batch(
    const uint8_t* __restrict__ in,
    uint32_t in_len,
    uint8_t* out,
    uint32_t out_cap,
    uint32_t op,
    uint32_t n,
    const Seq& s,
    uint32_t& err)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  const bool active = lane < n;
  const uint32_t lit_len = active ? s.lit_len : 0;
  const uint32_t match_len = active ? s.match_len : 0;
  const uint32_t len = lit_len + match_len;
  const uint32_t incl = wave::scan_add_inclusive(len);
  const uint32_t total = wave::read_lane(incl, 63);
  const uint32_t lit_dst = op + incl - len; /* absolute output position of this lane's literals */
  const uint32_t match_dst = lit_dst + lit_len;

  if (CHECKED) {
    /* 64-bit: a corrupt stream may claim lengths that wrap 32 bits */
    const uint64_t lane_end = (uint64_t)op + (uint64_t)incl;
    const bool bad_out = len > out_cap || lane_end > out_cap;
    const bool bad_off = match_len != 0 && (s.match_off == 0 || s.match_off > match_dst);
    const bool bad_in = lit_len != 0 && ((uint64_t)s.lit_src + lit_len > in_len);
    const uint64_t any_out = wave::ballot(bad_out);
    const uint64_t any_off = wave::ballot(bad_off);
    const uint64_t any_in = wave::ballot(bad_in);
    if (any_out | any_off | any_in) {
      err |= (any_out ? kErrOutput : 0u) | (any_off ? kErrOffset : 0u) | (any_in ? kErrInput : 0u);
      return 0;
    }
  }

  /* ---- literals ---- */
  {
    const uint32_t lit4 = (lit_len + 3u) & ~3u;
    const bool lit_lane = LANE_PARALLEL && lit_len != 0 && lit_len <= kLitShort
                          && (uint64_t)s.lit_src + lit4 <= in_len;
    const uint32_t max_lit = wave::reduce_max(lit_lane ? lit_len : 0u);
    const uint8_t* src = in + s.lit_src;
    uint8_t* dst = out + lit_dst;
    for (uint32_t i = 0; i < max_lit; i += 4) {
      if (lit_lane && i < lit_len) {
        st_upto4(dst + i, ld_u32(src + i), lit_len - i);
      }
    }
    uint64_t pending = wave::ballot(lit_len != 0 && !lit_lane);
    while (pending) {
      const uint32_t j = wave::ctz64(pending);
      pending &= pending - 1;
      const uint32_t jsrc = wave::read_lane(s.lit_src, j);
      const uint32_t jlen = wave::read_lane(lit_len, j);
      const uint32_t jdst = wave::read_lane(lit_dst, j);
      wave_copy(out + jdst, in + jsrc, jlen);
    }
  }
  wave::sync();

  /* ---- matches: multi-round resolution ---- */
  {
    const uint32_t match_src = match_dst - s.match_off;
    const bool match_lane = LANE_PARALLEL && match_len != 0 && match_len <= kMatchShort && s.match_off >= 4;
    uint64_t pending = wave::ballot(match_len != 0);
    const uint64_t lane_mask = wave::ballot(match_lane);
    const uint64_t lane_bit = 1ull << lane;
    while (pending) {
      const uint32_t f = wave::ctz64(pending);
      const uint32_t hw = wave::read_lane(match_dst, f); /* every byte below hw is final */
      if (!((lane_mask >> f) & 1)) {
        const uint32_t foff = wave::read_lane(s.match_off, f);
        const uint32_t flen = wave::read_lane(match_len, f);
        wave_match_copy(out + hw, foff, flen);
        pending &= ~(1ull << f);
        continue;
      }
      const bool ready = match_lane && (pending & lane_bit) && (lane == f || match_src + match_len <= hw);
      const uint32_t max_len = wave::reduce_max(ready ? match_len : 0u);
      const uint8_t* src = out + match_src;
      uint8_t* dst = out + match_dst;
      for (uint32_t i = 0; i < max_len; i += 4) {
        if (ready && i < match_len) {
          st_upto4(dst + i, ld_u32(src + i), match_len - i);
        }
      }
      wave::sync();
      pending &= ~wave::ballot(ready);
    }
  }
  return total;
}

} // namespace lz
