/*
 * cascaded/cascaded.hip.h -- batched Cascaded codec (RLE + delta + bit-packing)
 * for gfx950. Replaces the device side of nvcompBatchedCascaded{Compress,
 * Decompress,GetDecompressSize}Async (reference call sites:
 * benchmarks/benchmark_cascaded_chunked.cu:137-151; scheme:
 * doc/cascaded_overview.md:6-44).
 *
 * One wavefront per user chunk; the chunk is cut into sub-chunks of
 * opts.chunk_size bytes that are coded independently, one after the other, with
 * all intermediate streams in LDS:
 *   RLE    : run starts = ballot of (v[i] != v[i-1]), compacted with popcounts;
 *   delta  : d[i] = v[i] - v[i-1] per 64-element tile (back to front), inverse =
 *            DPP prefix sum with a carry between tiles;
 *   bitpack: wave-wide min/max, then every lane assembles whole 32-bit output
 *            words (no atomics); unpack is element-parallel.
 * The container is this library's own (DESIGN.md "Cascaded stream layout");
 * oracle/cascaded_ref.c is its bit-exact CPU model.
 */
#pragma once

#include "common/wave.h"

namespace casc {

constexpr uint32_t kMagic = 0x43534143u; /* 'CASC' */
constexpr uint32_t kRawMarker = 0xffffffffu;
constexpr uint32_t kMaxElems = 16384;

enum : uint32_t { kOk = 0, kErrInput = 1, kErrOutput = 2, kErrAlign = 4 };

struct Params
{
  uint32_t sub_bytes; /* opts.chunk_size */
  uint32_t type;      /* nvcompType_t 0..7 */
  uint32_t num_rles;
  uint32_t num_deltas;
  uint32_t use_bp;
  uint32_t max_bytes; /* compress: the chunk size the caller declared (its output slot is sized from it) */
};

__device__ __forceinline__ uint32_t type_width(uint32_t t)
{
  return 1u << (t >> 1); /* 0,1 -> 1; 2,3 -> 2; 4,5 -> 4; 6,7 -> 8 */
}

__device__ __forceinline__ bool type_signed(uint32_t t)
{
  return (t & 1u) == 0;
}

__device__ __forceinline__ uint64_t width_mask(uint32_t w)
{
  return w == 8 ? ~0ull : ((1ull << (8 * w)) - 1);
}

__device__ __forceinline__ uint32_t bits_for(uint64_t range)
{
  return range ? 64u - (uint32_t)__builtin_clzll(range) : 0u;
}

/* order-preserving key: signed values of width w -> unsigned order */
__device__ __forceinline__ uint64_t to_key(uint64_t v, uint32_t w, bool as_signed)
{
  if (!as_signed) {
    return v;
  }
  const uint32_t sh = 64 - 8 * w;
  return (uint64_t)((int64_t)(v << sh) >> sh) ^ 0x8000000000000000ull;
}

__device__ __forceinline__ uint64_t shuffle64(uint64_t v, uint32_t src)
{
  const uint32_t lo = wave::shuffle((uint32_t)v, src);
  const uint32_t hi = wave::shuffle((uint32_t)(v >> 32), src);
  return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint64_t reduce_min64(uint64_t v)
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  for (uint32_t m = 32; m >= 1; m >>= 1) {
    const uint64_t o = shuffle64(v, lane ^ m);
    v = o < v ? o : v;
  }
  return v;
}

__device__ __forceinline__ uint64_t reduce_max64(uint64_t v)
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  for (uint32_t m = 32; m >= 1; m >>= 1) {
    const uint64_t o = shuffle64(v, lane ^ m);
    v = o > v ? o : v;
  }
  return v;
}

/* inclusive prefix sum of 64-bit values across the wave */
__device__ __forceinline__ uint64_t scan_add64(uint64_t v)
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  for (uint32_t d = 1; d < 64; d <<= 1) {
    const uint64_t o = shuffle64(v, (lane - d) & 63u);
    if (lane >= d) {
      v += o;
    }
  }
  return v;
}

/* ---- element access: streams are arrays of T (values) or uint16_t (runs) ---- */

template <typename T>
struct Stream
{
  static constexpr uint32_t kBytes = sizeof(T);
  const T* p;
  __device__ __forceinline__ uint64_t get(uint32_t i) const { return (uint64_t)p[i]; }
  __device__ __forceinline__ uint32_t get32(uint32_t i) const { return (uint32_t)p[i]; }
};

/* min / range of a stream -> (min value as stored, bits). Elements of up to 4 bytes are ranged in 32-bit
 * arithmetic (one DPP reduction per bound); 8-byte elements take the 64-bit path. */
template <typename S>
__device__ __forceinline__ void stream_range(const S& s, uint32_t count, uint32_t w, bool as_signed, uint64_t& mn_out,
                                             uint32_t& bits_out)
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  if (count == 0) {
    mn_out = 0;
    bits_out = 0;
    return;
  }
  if constexpr (S::kBytes <= 4) {
    const uint32_t sh = 32 - 8 * w;
    const uint32_t flip = as_signed ? 0x80000000u : 0u;
    uint32_t lo = ~0u, hi = 0;
    for (uint32_t i = lane; i < count; i += 64) {
      const uint32_t v = s.get32(i);
      const uint32_t k = (as_signed ? (uint32_t)((int32_t)(v << sh) >> sh) : v) ^ flip;
      lo = k < lo ? k : lo;
      hi = k > hi ? k : hi;
    }
    lo = ~wave::reduce_max(~lo);
    hi = wave::reduce_max(hi);
    const uint32_t range = hi - lo;
    bits_out = range ? 32u - (uint32_t)__builtin_clz(range) : 0u;
    mn_out = (uint64_t)(lo ^ flip) & width_mask(w);
  } else {
    uint64_t lo = ~0ull, hi = 0;
    for (uint32_t i = lane; i < count; i += 64) {
      const uint64_t k = to_key(s.get(i), w, as_signed);
      lo = k < lo ? k : lo;
      hi = k > hi ? k : hi;
    }
    lo = reduce_min64(lo);
    hi = reduce_max64(hi);
    bits_out = bits_for(hi - lo);
    /* back from key space: the minimum as a w-byte value */
    mn_out = (as_signed ? (lo ^ 0x8000000000000000ull) : lo) & width_mask(w);
  }
}

__device__ __forceinline__ uint32_t stream_bytes(uint32_t count, uint32_t bits)
{
  return 12 + 4 * (uint32_t)(((uint64_t)count * bits + 31) / 32);
}

/* Write one packed stream at dst (global, 4-byte aligned). Every lane assembles whole output words from the
 * elements that overlap them. The first such element of word k is floor(32 k / bits): a multiply by the
 * stream's reciprocal (exact: 32 k < 2^21, bits <= 64), not a division per word; the last one is found by
 * walking. Elements of up to 4 bytes with bits <= 32 are assembled in 32-bit arithmetic. */
template <typename S>
__device__ __forceinline__ uint32_t pack_stream(uint8_t* dst, const S& s, uint32_t count, uint32_t w, uint64_t mn,
                                                uint32_t bits)
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  uint32_t* out = (uint32_t*)dst;
  if (lane == 0) {
    out[0] = bits;
    out[1] = (uint32_t)mn;
    out[2] = (uint32_t)(mn >> 32);
  }
  const uint32_t words = (count * bits + 31) / 32; /* count <= kMaxElems: no overflow */
  if (words == 0) {
    return 12;
  }
  const uint32_t recip = bits > 1 ? (uint32_t)((0x100000000ull + bits - 1) / bits) : 0u;
  if (S::kBytes <= 4 && bits <= 32) {
    const uint32_t mn32 = (uint32_t)mn;
    const uint32_t vmask = bits == 32 ? ~0u : ((1u << bits) - 1u); /* bits <= 8 w: the width mask is implied */
    if (bits > 16) {
      /* more than 16 bits an element (float columns: 24-28): a word holds parts of at most three elements -- straight-line
       * code, no per-lane loop (the loop below was 40 % of the compressor's time on such data: phase clock, round 5).
       * An element beyond the count is read as the last one and contributes nothing. */
      const uint32_t last = count - 1;
      for (uint32_t k = lane; k < words; k += 64) {
        const uint32_t bit0 = 32 * k;
        const uint32_t e = __umulhi(bit0, recip);
        const uint32_t back = bit0 - wave::mul24(e, bits); /* bits of element e in front of this word: 0 ... bits - 1 */
        const uint32_t x0 = (s.get32(e) - mn32) & vmask;
        const uint32_t x1 = e + 1 <= last ? (s.get32(e + 1) - mn32) & vmask : 0u;
        const uint32_t x2 = e + 2 <= last ? (s.get32(e + 2 <= last ? e + 2 : last) - mn32) & vmask : 0u;
        const uint32_t r1 = bits - back;     /* where element e + 1 starts in the word: 1 ... 32 */
        const uint32_t r2 = r1 + bits;       /* ... and e + 2: may lie behind the word */
        uint32_t word = x0 >> back;
        word |= r1 < 32 ? x1 << r1 : 0u;
        word |= r2 < 32 ? x2 << r2 : 0u;
        out[3 + k] = word;
      }
      return 12 + 4 * words;
    }
    for (uint32_t k = lane; k < words; k += 64) {
      const uint32_t bit0 = 32 * k;
      uint32_t e = bits > 1 ? __umulhi(bit0, recip) : bit0;
      int32_t rel = (int32_t)wave::mul24(e, bits) - (int32_t)bit0; /* in (-bits, 0] */
      uint32_t word = 0;
      for (; rel < 32 && e < count; ++e, rel += (int32_t)bits) {
        const uint32_t x = (s.get32(e) - mn32) & vmask;
        word |= rel >= 0 ? x << rel : x >> (-rel);
      }
      out[3 + k] = word;
    }
    return 12 + 4 * words;
  }
  const uint64_t wmask = width_mask(w);
  const uint64_t vmask = bits == 64 ? ~0ull : ((1ull << bits) - 1);
  for (uint32_t k = lane; k < words; k += 64) {
    const uint32_t bit0 = 32 * k;
    uint32_t e = bits > 1 ? __umulhi(bit0, recip) : bit0;
    int32_t rel = (int32_t)(e * bits) - (int32_t)bit0;
    uint32_t word = 0;
    for (; rel < 32 && e < count; ++e, rel += (int32_t)bits) {
      const uint64_t x = ((s.get(e) - mn) & wmask) & vmask;
      word |= rel >= 0 ? (uint32_t)(x << rel) : (uint32_t)(x >> (-rel));
    }
    out[3 + k] = word;
  }
  return 12 + 4 * words;
}

/* Unpack a stream from src (global, 4-byte aligned) into LDS dst[0..count). */
template <typename T>
__device__ __forceinline__ bool unpack_stream(const uint8_t* src, uint32_t avail, T* dst, uint32_t count, uint32_t w,
                                              uint32_t& used)
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  used = 0;
  if (avail < 12) {
    return false;
  }
  const uint32_t* in = (const uint32_t*)src;
  const uint32_t bits = in[0];
  const uint64_t mn = (uint64_t)in[1] | ((uint64_t)in[2] << 32);
  if (bits > 64) {
    return false;
  }
  const uint32_t words = (uint32_t)(((uint64_t)count * bits + 31) / 32);
  if (avail < 12 + 4 * (uint64_t)words) {
    return false;
  }
  const uint64_t wmask = width_mask(w);
  const uint64_t vmask = bits == 64 ? ~0ull : ((1ull << bits) - 1);
  /* several tiles per step: their loads are issued together -- one memory round trip instead of one per tile.
   * Elements of up to 4 bytes never straddle three words, so they get 8 tiles in flight for the same registers. */
  constexpr uint32_t kTiles = sizeof(T) <= 4 ? 8 : 4;
  if (sizeof(T) <= 4 && bits != 0 && bits <= 32 && count != 0) {
    /* Elements of up to 4 bytes in 32-bit arithmetic throughout (round 5: the unpack was 36 % of the decoder's time on
     * the float columns, most of it 64-bit multiplies and shifts the values never needed): the element's bit position
     * is a 24-bit product (count <= 16 384, bits <= 32), its bits come out of two words with one v_alignbit. Both words
     * are loaded whenever they exist: no lane-dependent branch around the second load. */
    const uint32_t vmask32 = bits == 32 ? ~0u : ((1u << bits) - 1u);
    const uint32_t mn32 = (uint32_t)mn;
    /* no lane-dependent branches: a lane beyond the count works on the last element once more (the same value to the same
     * place), the second word's index stops at the stream's last word (an element that needs its second word has one) */
    const uint32_t last = count - 1;
    for (uint32_t base = 0; base < count; base += 64 * kTiles) {
      uint32_t lo[kTiles], hi[kTiles];
#pragma unroll
      for (uint32_t u = 0; u < kTiles; ++u) {
        const uint32_t i = base + 64 * u + lane < last ? base + 64 * u + lane : last;
        const uint32_t k = wave::mul24(i, bits) >> 5;
#ifdef NVCOMP_CASC_ABLATE_UNPACK_LOADS /* profiling builds only (wrong output): what do the loads of the unpack cost? */
        lo[u] = k;
        hi[u] = i;
#else
        lo[u] = in[3 + k];
        hi[u] = in[3 + (k + 1 < words ? k + 1 : k)];
#endif
      }
#pragma unroll
      for (uint32_t u = 0; u < kTiles; ++u) {
        const uint32_t i = base + 64 * u + lane < last ? base + 64 * u + lane : last;
        const uint32_t sh = wave::mul24(i, bits) & 31u;
        const uint32_t x = wave::align_bits(hi[u], lo[u], sh) & vmask32;
        dst[i] = (T)(x + mn32);
      }
    }
    used = 12 + 4 * words;
    return true;
  }
  for (uint32_t base = 0; base < count; base += 64 * kTiles) {
    uint32_t w0[kTiles], w1[kTiles], w2[sizeof(T) <= 4 ? 1 : kTiles];
#pragma unroll
    for (uint32_t u = 0; u < kTiles; ++u) {
      const uint32_t i = base + 64 * u + lane;
      w0[u] = w1[u] = 0;
      if (sizeof(T) > 4) {
        w2[u] = 0;
      }
      if (bits && i < count) {
        const uint64_t bit = (uint64_t)i * bits;
        const uint32_t k = (uint32_t)(bit / 32);
        const uint32_t sh = (uint32_t)(bit % 32);
        w0[u] = in[3 + k];
        if (sh + bits > 32) {
          w1[u] = in[3 + k + 1];
        }
        if (sizeof(T) > 4 && sh + bits > 64) {
          w2[u] = in[3 + k + 2];
        }
      }
    }
#pragma unroll
    for (uint32_t u = 0; u < kTiles; ++u) {
      const uint32_t i = base + 64 * u + lane;
      if (i < count) {
        uint64_t x = 0;
        if (bits) {
          const uint32_t sh = (uint32_t)(((uint64_t)i * bits) % 32);
          x = (uint64_t)w0[u] >> sh;
          if (sh + bits > 32) {
            x |= (uint64_t)w1[u] << (32 - sh);
          }
          if (sizeof(T) > 4 && sh + bits > 64) {
            x |= (uint64_t)w2[u] << (64 - sh);
          }
          x &= vmask;
        }
        dst[i] = (T)((x + mn) & wmask);
      }
    }
  }
  used = 12 + 4 * words;
  return true;
}

/* dst <- src, whole wave, non-overlapping: dwords when both sides are 4-byte aligned */
__device__ __forceinline__ void copy_bytes(uint8_t* dst, const uint8_t* src, uint32_t bytes)
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  uint32_t done = 0;
  if ((((uintptr_t)dst | (uintptr_t)src) & 3u) == 0) {
    const uint32_t words = bytes / 4;
    for (uint32_t i = lane; i < words; i += 64) {
      ((uint32_t*)dst)[i] = ((const uint32_t*)src)[i];
    }
    done = words * 4;
  }
  for (uint32_t i = done + lane; i < bytes; i += 64) {
    dst[i] = src[i];
  }
}

/* ---- layers ---------------------------------------------------------------- */

constexpr uint32_t kRleOverflow = 0xffffffffu;

/* v of the lane below (lane 0: unspecified) and of lane l (wave-uniform), for elements of any width: DPP / v_readlane, no
 * memory access. */
template <typename T>
__device__ __forceinline__ T lane_below(T v)
{
  if constexpr (sizeof(T) <= 4) {
    return (T)wave::prev_lane((uint32_t)v);
  } else {
    const uint32_t lo = wave::prev_lane((uint32_t)v), hi = wave::prev_lane((uint32_t)((uint64_t)v >> 32));
    return (T)(((uint64_t)hi << 32) | lo);
  }
}
template <typename T>
__device__ __forceinline__ T lane_value(T v, uint32_t l)
{
  if constexpr (sizeof(T) <= 4) {
    return (T)wave::read_lane((uint32_t)v, l);
  } else {
    const uint32_t lo = wave::read_lane((uint32_t)v, l), hi = wave::read_lane((uint32_t)((uint64_t)v >> 32), l);
    return (T)(((uint64_t)hi << 32) | lo);
  }
}

/* Number of run heads in A[0..c): elements that differ from their predecessor (A may be HBM or LDS). Whole groups of 256
 * elements: four loads in flight, the predecessor is the value of the lane below (a tile's first element: the last one of
 * the tile before) -- one load and three or four vector instructions a tile where the plain loop (the tail's) has two
 * loads, their addresses and bounds: the count ran at 31 % of the compressor's time on run-poor data (phase clock). */
template <typename T>
__device__ __forceinline__ uint32_t count_heads(const T* A, uint32_t c)
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  uint32_t m = 0;
  uint32_t base = 0;
  T before = (T)0; /* element base - 1 */
  for (; base + 256 <= c; base += 256) {
    T v[4];
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
      v[u] = A[base + 64 * u + lane];
    }
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
      const T below = lane_below(v[u]);
      const T prev = lane == 0 ? before : below;
      const bool head = v[u] != prev || (u == 0 && base + lane == 0);
      m += wave::popc64(wave::ballot(head));
      before = lane_value(v[u], 63);
    }
  }
  for (; base < c; base += 64) {
    const uint32_t i = base + lane;
    const bool in = i < c;
    const T v = in ? A[i] : (T)0;
    const T prev = (in && i > 0) ? A[i - 1] : (T)0;
    m += wave::popc64(wave::ballot(in && (i == 0 || v != prev)));
  }
  return m;
}

/* RLE of A[0..c) -> values in B, run lengths in runs; returns the new count. B == A is allowed (compaction in place).
 * lengths_if_runs: the caller drops the run stream of a layer that found no runs (every length is 1), so the second pass
 * -- start indices to lengths, as long as the first on run-poor data such as float columns -- is skipped for it. */
template <typename T>
__device__ __forceinline__ uint32_t rle_encode(const T* A, uint32_t c, T* B, uint16_t* runs, uint32_t cap, bool lengths_if_runs = false)
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  uint32_t m = 0;
  /* pass 1: compact run starts; runs[] temporarily holds the start index of each run. 4 tiles per step: their
   * loads (the input may be HBM) are issued together, the compaction itself stays tile by tile. */
  for (uint32_t base4 = 0; base4 < c; base4 += 256) {
    T v4[4], p4[4];
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
      const uint32_t i = base4 + 64 * u + lane;
      v4[u] = i < c ? A[i] : (T)0;
      p4[u] = (i < c && i > 0) ? A[i - 1] : (T)0;
    }
    wave::sync(); /* B may be A (in place): everything this step reads has been read; outputs land at or below it */
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
      const uint32_t i = base4 + 64 * u + lane;
      const bool in = i < c;
      const T v = v4[u];
      const bool head = in && (i == 0 || v != p4[u]);
      const uint64_t mask = wave::ballot(head);
      const uint32_t rank = wave::popc64(mask & ((1ull << lane) - 1));
      if (m + wave::popc64(mask) > cap) {
        return kRleOverflow; /* more runs than B / runs can hold: the caller retries with a larger LDS slice */
      }
      if (head) {
        B[m + rank] = v;
        runs[m + rank] = (uint16_t)i;
      }
      m += wave::popc64(mask);
    }
    wave::sync();
  }
  wave::sync();
  if (lengths_if_runs && m == c) {
    return m;
  }
  /* pass 2: start indices -> run lengths (start of the next run minus own start) */
  for (uint32_t base = 0; base < m; base += 64) {
    const uint32_t j = base + lane;
    uint32_t len = 0;
    if (j < m) {
      const uint32_t next = j + 1 < m ? runs[j + 1] : c;
      len = next - runs[j];
    }
    wave::sync();
    if (j < m) {
      runs[j] = (uint16_t)len;
    }
    wave::sync();
  }
  return m;
}

/* in place: A[i] -= A[i-1] for i >= 1 (tiles back to front) */
template <typename T>
__device__ __forceinline__ void delta_encode(T* A, uint32_t c)
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  const uint32_t tiles = (c + 63) / 64;
  for (uint32_t t = tiles; t-- > 0;) {
    const uint32_t i = t * 64 + lane;
    T d = 0;
    if (i < c) {
      d = i > 0 ? (T)(A[i] - A[i - 1]) : A[i];
    }
    wave::sync();
    if (i < c) {
      A[i] = d;
    }
    wave::sync();
  }
}

/* wave-wide inclusive scan helpers in T's arithmetic, plus the value of lane 63 */
template <typename T>
__device__ __forceinline__ uint64_t scan_add_t(uint64_t v)
{
  return sizeof(T) <= 4 ? (uint64_t)wave::scan_add_inclusive((uint32_t)v) : scan_add64(v);
}

template <typename T>
__device__ __forceinline__ uint64_t last_lane_t(uint64_t v)
{
  if (sizeof(T) <= 4) {
    return (uint64_t)wave::read_lane((uint32_t)v, 63);
  }
  return ((uint64_t)wave::read_lane((uint32_t)(v >> 32), 63) << 32) | wave::read_lane((uint32_t)v, 63);
}

/* Inclusive prefix sum (inverse delta) of A[0..c) into out[0..c): out == A (in place), or the sub-chunk's place in memory
 * when nothing follows the delta -- a lane's four elements are 16 consecutive bytes, a step of the wave 1 KiB: the separate
 * copy out of LDS (a load, a store and the loop per 64 elements: 6 % of the decoder's time on the float columns) falls away.
 * A lane owns 4 consecutive elements per step, so a step of 256 elements costs one wave scan: the decoder is bound by its
 * chains of dependent LDS accesses and scans, not by instruction count, and this cuts the chain 4x. */
template <typename T>
__device__ __forceinline__ void delta_decode(T* A, uint32_t c, T* out)
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  if (c <= 64) { /* a well-compressed layer: one element per lane, one scan */
    const uint64_t v = lane < c ? (uint64_t)A[lane] : 0;
    const uint64_t incl = scan_add_t<T>(sizeof(T) <= 4 ? (uint64_t)(uint32_t)v : v);
    wave::sync();
    if (lane < c) {
      out[lane] = (T)incl;
    }
    wave::sync();
    return;
  }
  uint64_t carry = 0;
  for (uint32_t base = 0; base < c; base += 256) {
    const uint32_t i0 = base + 4 * lane;
    const bool whole = base + 256 <= c; /* (wave-uniform: the stores of a whole step carry no bounds) */
    uint64_t v[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
      v[k] = (whole || i0 + k < c) ? (uint64_t)A[i0 + k] : 0;
    }
    v[1] += v[0];
    v[2] += v[1];
    v[3] += v[2];
    const uint64_t incl = scan_add_t<T>(sizeof(T) <= 4 ? (uint64_t)(uint32_t)v[3] : v[3]);
    const uint64_t before = incl - v[3] + carry;
    wave::sync();
    if (whole) {
#pragma unroll
      for (uint32_t k = 0; k < 4; ++k) {
        out[i0 + k] = (T)(v[k] + before);
      }
    } else {
#pragma unroll
      for (uint32_t k = 0; k < 4; ++k) {
        if (i0 + k < c) {
          out[i0 + k] = (T)(v[k] + before);
        }
      }
    }
    carry += last_lane_t<T>(incl);
  }
  wave::sync();
}

/* Expand values A[0..c) with run lengths runs[0..c) into B[0..target) (B may be HBM).
 * marks: LDS scratch of `target` uint16 (rounded up to 4). Returns false if the runs are inconsistent.
 * 4 consecutive entries per lane and step, as in delta_decode. */
template <typename T>
__device__ __forceinline__ bool rle_decode(const T* A, const uint16_t* runs, uint32_t c, T* B, uint32_t target,
                                           uint16_t* marks)
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  for (uint32_t i = lane; i < (target + 1) / 2; i += 64) {
    ((uint32_t*)marks)[i] = 0;
  }
  wave::sync();
  /* exclusive scan of run lengths -> start of every run; mark it with (run index + 1) */
  uint32_t carry = 0;
  bool bad = false;
  if (c <= 64) { /* few runs: one per lane */
    const uint32_t r = lane < c ? runs[lane] : 0;
    const uint32_t incl = wave::scan_add_inclusive(r);
    if (lane < c) {
      if (r == 0 || incl > target) {
        bad = true;
      } else {
        marks[incl - r] = (uint16_t)(lane + 1);
      }
    }
    carry = wave::read_lane(incl, 63);
  }
  for (uint32_t base = 0; c > 64 && base < c; base += 256) {
    const uint32_t j0 = base + 4 * lane;
    uint32_t r[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
      r[k] = j0 + k < c ? runs[j0 + k] : 0;
    }
    const uint32_t mine = r[0] + r[1] + r[2] + r[3];
    const uint32_t incl = wave::scan_add_inclusive(mine);
    uint32_t start = incl - mine + carry;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
      if (j0 + k < c) {
        if (r[k] == 0 || start + r[k] > target) {
          bad = true;
        } else {
          marks[start] = (uint16_t)(j0 + k + 1);
        }
      }
      start += r[k];
    }
    carry += wave::read_lane(incl, 63);
  }
  if (wave::ballot(bad) || carry != target) {
    return false;
  }
  wave::sync();
  /* running maximum of the marks = index of the run every output element belongs to */
  uint32_t run_carry = 0;
  for (uint32_t base = 0; base < target; base += 256) {
    const uint32_t i0 = base + 4 * lane;
    uint32_t mk[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
      mk[k] = i0 + k < target ? marks[i0 + k] : 0;
    }
    mk[1] = mk[1] > mk[0] ? mk[1] : mk[0];
    mk[2] = mk[2] > mk[1] ? mk[2] : mk[1];
    mk[3] = mk[3] > mk[2] ? mk[3] : mk[2];
    const uint32_t incl = wave::scan_max_inclusive(mk[3]);
    /* maximum over the lanes before this one = the inclusive maximum of the lane below */
    uint32_t before = wave::shuffle(incl, (lane - 1) & 63u);
    before = lane == 0 ? 0u : before;
    before = before > run_carry ? before : run_carry;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
      if (i0 + k < target) {
        const uint32_t run = mk[k] > before ? mk[k] : before;
        B[i0 + k] = A[run - 1];
      }
    }
    const uint32_t last = wave::read_lane(incl, 63);
    run_carry = last > run_carry ? last : run_carry;
  }
  wave::sync();
  return true;
}

/* Expansion of layers whose runs are all short (at most kDirectMaxRun, at most kDirectMaxBits bits each in the
 * packed stream): the run lengths are extracted straight from the stream's words -- 4 consecutive runs per lane
 * out of two dwords -- so neither a run pool nor the marks exist in LDS, an inner layer expands IN PLACE, and the
 * sub-chunk decodes inside the smallest slice (8 waves per SIMD). This is the shape of smooth float columns
 * (BASELINE.json configs[3]): a few repeated neighbours per sub-chunk, short runs of equal deltas below that. */
constexpr uint32_t kDirectMaxBits = 6;
constexpr uint32_t kDirectMaxRun = 64;

__host__ __device__ inline bool runs_are_short(uint32_t bits, uint32_t mn_lo, uint32_t mn_hi)
{
  return bits <= kDirectMaxBits && mn_hi == 0 && mn_lo >= 1 && mn_lo + (1u << bits) - 1u <= kDirectMaxRun;
}

/* Runs j0..j0+3 (zero beyond c) and their values for this lane. packed: the stream's words after its header. */
template <typename T>
__device__ __forceinline__ void load_short_runs(const T* A, const uint32_t* packed, uint32_t bits, uint32_t mn, uint32_t c,
                                                uint32_t j0, uint32_t (&r)[4], T (&a)[4])
{
  const uint32_t words = (c * bits + 31) / 32;
  const uint32_t mask = (1u << bits) - 1u;
  uint64_t v = 0;
  if (bits && j0 < c) {
    const uint32_t bit = wave::mul24(j0, bits);
    const uint32_t k = bit >> 5;
    const uint32_t lo = packed[k];
    const uint32_t hi = k + 1 < words ? packed[k + 1] : 0u;
    v = (((uint64_t)hi << 32) | lo) >> (bit & 31u);
  }
#pragma unroll
  for (uint32_t k = 0; k < 4; ++k) {
    const bool live = j0 + k < c;
    r[k] = live ? mn + ((uint32_t)(v >> (k * bits)) & mask) : 0u;
    a[k] = live ? A[j0 + k] : (T)0;
  }
}

template <typename T>
__device__ __forceinline__ void store_short_runs(T* out, uint32_t pos, const uint32_t (&r)[4], const T (&a)[4])
{
#pragma unroll
  for (uint32_t k = 0; k < 4; ++k) {
    for (uint32_t q = 0; q < r[k]; ++q) {
      out[pos + q] = a[k];
    }
    pos += r[k];
  }
}

/* Outermost layer: A (LDS, c values) -> out (HBM, target elements), ascending. */
template <typename T>
__device__ __forceinline__ bool rle_expand_direct(const T* A, const uint32_t* packed, uint32_t bits, uint32_t mn, uint32_t c,
                                                  T* out, uint32_t target)
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  uint32_t carry = 0;
  bool bad = false;
  for (uint32_t base = 0; base < c; base += 256) {
    uint32_t r[4];
    T a[4];
    load_short_runs(A, packed, bits, mn, c, base + 4 * lane, r, a);
    const uint32_t mine = r[0] + r[1] + r[2] + r[3];
    const uint32_t incl = wave::scan_add_inclusive(mine);
    const uint32_t pos = incl - mine + carry;
    if (pos + mine > target) {
      bad = true;
    } else {
      store_short_runs(out, pos, r, a);
    }
    carry += wave::read_lane(incl, 63);
  }
  return !wave::ballot(bad) && carry == target;
}

/* Inner layer: A[0..c) -> A[0..target) in place. Tiles run from the top down: tile t's output starts at the sum of
 * all runs below it, which is at least 256 t (every run is >= 1) -- it lands on values this tile already holds in
 * registers or that higher tiles have consumed, never on values still to be read. */
template <typename T>
__device__ __forceinline__ bool rle_expand_inplace(T* A, const uint32_t* packed, uint32_t bits, uint32_t mn, uint32_t c,
                                                   uint32_t target)
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  uint32_t above = 0; /* elements produced by the tiles above */
  for (uint32_t t = (c + 255) / 256; t-- > 0;) {
    uint32_t r[4];
    T a[4];
    load_short_runs(A, packed, bits, mn, c, 256 * t + 4 * lane, r, a);
    const uint32_t mine = r[0] + r[1] + r[2] + r[3];
    const uint32_t incl = wave::scan_add_inclusive(mine);
    const uint32_t total = wave::read_lane(incl, 63);
    if (above + total > target) {
      return false;
    }
    wave::sync(); /* every lane holds its values before any lane overwrites them */
    store_short_runs(A, target - above - total + (incl - mine), r, a);
    wave::sync();
    above += total;
  }
  return above == target;
}

/* ---- phase clock (profiling builds only: -DNVCOMP_CASC_PROF) ---- */
#ifdef NVCOMP_CASC_PROF
constexpr uint32_t kProfSlots = 12;
__device__ unsigned long long g_prof[kProfSlots * 64]; /* 64 copies of every slot: the atomics of 8 000 waves on one word would be the profile */
struct ProfClock
{
  unsigned long long last;
  __device__ __forceinline__ void begin() { last = __builtin_readcyclecounter(); }
  __device__ __forceinline__ void mark(uint32_t slot)
  {
    const unsigned long long t = __builtin_readcyclecounter();
    if (wave::lane_id() == 0) {
      atomicAdd(&g_prof[slot * 64 + (blockIdx.x & 63u)], t - last);
    }
    last = __builtin_readcyclecounter();
  }
};
#define CASC_PROF_DECL casc::ProfClock prof_clock; prof_clock.begin()
#define CASC_T(slot) prof_clock.mark(slot)
#else
#define CASC_PROF_DECL ((void)0)
#define CASC_T(slot) ((void)0)
#endif

/* ---- sub-chunk codec ------------------------------------------------------- */

constexpr uint32_t kSubNeedsLds = 0xffffffffu; /* compress_sub: the streams do not fit the LDS slice it was given */

/* Per-wave bookkeeping of the decoder's layer loops, kept in LDS: indexing private arrays with the
 * run-time layer number would put them in scratch memory. (The compressor packs every layer as soon as it
 * has run and keeps nothing per layer.) */
struct LayerMeta
{
  uint32_t counts[8];
  uint32_t run_off[8];
  uint32_t ident[8];   /* RLE layer l found no runs: values pass through, every run length is 1 */
  uint32_t src_off[8]; /* decode: where layer l's run stream starts inside the sub-chunk */
};

/* Compress one sub-chunk with the calling wave into dst; returns its size, or kSubNeedsLds when the
 * intermediate streams do not fit the `budget` bytes of LDS at `lds` (nothing final has been decided then:
 * the caller repeats the whole chunk in a pass with a larger slice). With any layer at all the sub-chunk is staged into
 * V once (round 5) and every layer works on LDS; what a compressible sub-chunk saves is the run pool, not the staging.
 * Since round 5 an RLE layer that would take out fewer than one element in eight is written as an identity layer (run
 * lengths of 1): streams of that rule decode with every earlier build of the decoder -- an identity layer is an ordinary
 * run-length stream -- but they are not the bytes the earlier compressors wrote.
 * Every layer's run stream is packed into dst as soon as the layer has run -- the streams are laid out in layer
 * order, so its position is known -- which leaves ONE run pool, reused by every layer: the worst case is the value
 * buffer plus 2 bytes per element, whatever num_RLEs is. A sub-chunk that stops shrinking is stored raw; the
 * speculative stream writes never pass the raw size, so they stay inside what the caller reserved. */
template <typename T>
__device__ __forceinline__ uint32_t compress_sub(
    const uint8_t* src, uint32_t bytes, uint8_t* dst, const Params& p, uint8_t* lds, uint32_t budget)
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  const uint32_t w = sizeof(T);
  const uint32_t n = bytes / w;
  const T* in = (const T*)src;
  CASC_PROF_DECL;
  const uint32_t rl = p.num_rles;
  const uint32_t layers = rl > p.num_deltas ? rl : p.num_deltas;
  /* One value buffer V of n elements and the run pool behind it. Layer 0 reads the input from HBM and compacts
   * it into V; later layers compact V in place. A layer that finds no runs is the identity: its runs are dropped
   * and, with bit-packing, its run stream is just a header (the bytes are what the general path would write).
   * kSubNeedsLds when V or the run pool does not fit `budget`. */
  const bool can_skip = p.use_bp != 0;
  const uint32_t v_bytes = (n * w + 15u) & ~15u;
  if (v_bytes + 64 > budget) {
    return kSubNeedsLds;
  }
  T* V = (T*)lds;
  uint16_t* pool = (uint16_t*)(lds + v_bytes);
  const uint32_t pool_cap = (budget - v_bytes) / 2;
  const uint32_t raw_sz = 4 + ((bytes + 3u) & ~3u);
  uint32_t* out32 = (uint32_t*)dst;
  uint32_t pos = 4 + 4 * rl;
  bool raw = pos >= raw_sz;
  uint32_t c = n;
  const T* cur = in; /* HBM until the data has been put into V */
  if (layers != 0 && !raw) {
    /* The sub-chunk is staged into V once, eight tiles' loads in flight together (round 5): the layers -- the count of run
     * heads that decides whether an RLE layer pays, the compaction, the deltas -- then work on LDS. Counting the heads
     * straight from memory, a tile at a time, was 43 % of the compressor's time: sixteen dependent round trips a
     * sub-chunk (phase clock, scripts/casc_prof.py --compress). */
    for (uint32_t base = 0; base < n; base += 512) {
      T v[8];
#pragma unroll
      for (uint32_t u = 0; u < 8; ++u) {
        const uint32_t i = base + 64 * u + lane;
        v[u] = in[i < n ? i : n - 1];
      }
#pragma unroll
      for (uint32_t u = 0; u < 8; ++u) {
        const uint32_t i = base + 64 * u + lane;
        if (i < n) {
          V[i] = v[u];
        }
      }
    }
    wave::sync();
    cur = V;
  }
  for (uint32_t l = 0; l < layers && !raw; ++l) {
    if (l < rl) {
      /* A layer that would take out fewer than one element in eight is left out: its values pass through, every run
       * length is 1 (a legal run-length stream; with bit-packing it is just a header). Such a layer saves next to nothing
       * and costs the decoder a full expansion -- on smooth float columns, where two sub-chunks in five have a few equal
       * neighbours, the two expansions were 42 % of the decoder's time (phase clock, round 5). The heads are counted
       * first: the compaction works in place and cannot be undone. */
      bool id = false;
      if (can_skip) {
        const uint32_t heads = count_heads(cur, c);
        id = heads + (c >> 3) > c;
      }
      CASC_T(8); /* compress: run heads counted */
      if (!id) {
        const uint32_t m = rle_encode(cur, c, V, pool, pool_cap);
        if (m == kRleOverflow) {
          return kSubNeedsLds;
        }
        cur = V;
        c = m;
      }
      wave::sync();
      CASC_T(9); /* compress: run-length compaction */
      uint64_t mn;
      uint32_t bits;
      if (id) { /* all run lengths are 1: what stream_range finds for them, without the stream */
        mn = c ? 1 : 0;
        bits = 0;
      } else if (p.use_bp) {
        Stream<uint16_t> s{pool};
        stream_range(s, c, 2, false, mn, bits);
      } else {
        mn = 0;
        bits = 16;
      }
      const uint32_t sb = stream_bytes(c, bits);
      if (pos + sb >= raw_sz) {
        raw = true;
        break;
      }
      if (lane == 0) {
        out32[1 + l] = c;
      }
      Stream<uint16_t> s{pool};
      pack_stream(dst + pos, s, c, 2, mn, bits);
      pos += sb;
      wave::sync(); /* the pool is free for the next layer */
      CASC_T(10); /* compress: run stream ranged and packed */
    }
    if (l < p.num_deltas) {
      delta_encode(V, c); /* (the sub-chunk has been in V since the staging above) */
      CASC_T(11); /* compress: staging + delta */
    }
  }
  wave::sync();
  if (!raw) {
    const bool as_signed = p.num_deltas > 0 ? true : type_signed(p.type);
    uint64_t mn = 0;
    uint32_t bits = 8 * w;
    Stream<T> s{(T*)cur};
    if (p.use_bp) {
      stream_range(s, c, w, as_signed, mn, bits);
    }
    const uint32_t sb = stream_bytes(c, bits);
    if (pos + sb >= raw_sz) {
      raw = true;
    } else {
      CASC_T(6); /* compress: values ranged */
      pack_stream(dst + pos, s, c, w, mn, bits);
      pos += sb;
      CASC_T(7); /* compress: values packed */
    }
  }
  if (raw) { /* would not shrink: marker + the bytes, zero padded to a multiple of 4 */
    if (lane == 0) {
      out32[0] = kRawMarker;
    }
    const uint32_t padded = raw_sz - 4;
    copy_bytes(dst + 4, src, bytes);
    if (lane < padded - bytes) {
      dst[4 + bytes + lane] = 0;
    }
    return raw_sz;
  }
  if (lane == 0) {
    out32[0] = n;
  }
  return pos;
}

/* Result of decompress_sub. */
constexpr uint32_t kSubOk = 0;
constexpr uint32_t kSubBad = 1;     /* malformed stream */
constexpr uint32_t kSubNeedLds = 2; /* valid so far, but its streams do not fit `budget`: decode it in a larger pass */

__device__ __forceinline__ uint32_t align16(uint32_t v)
{
  return (v + 15u) & ~15u;
}

/* Decode one sub-chunk with the calling wave. `lds`/`budget`: this wave's LDS slice. The slice is carved from
 * the stream's ACTUAL counts: LayerMeta, then either ONE value buffer of counts[0] elements when every expanding
 * layer has short runs (rle_expand_direct / rle_expand_inplace: no pool, no marks), or two value buffers, the run
 * pools and the marks of the final expansion. Well-compressed or run-poor data needs a few KiB where the worst
 * case needs 3.5 x the sub-chunk; the outermost RLE layer expands straight into `dst` (HBM). */
template <typename T>
__device__ __forceinline__ uint32_t decompress_sub(
    const uint8_t* src, uint32_t avail, uint32_t head, uint8_t* dst, uint32_t bytes, uint32_t num_rles, uint32_t num_deltas,
    uint8_t* lds, uint32_t budget)
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  const uint32_t w = sizeof(T);
  const uint32_t n = bytes / w;
  CASC_PROF_DECL;
  if (avail < 4) {
    return kSubBad;
  }
  /* `head`: the first 256 bytes of the sub-chunk (lane i holds dword i, zero beyond `avail`), loaded by the caller
   * while the previous sub-chunk was being decoded: all of the header words (a compressible sub-chunk fits
   * entirely), read back with v_readlane instead of a dependent HBM round trip each */
  auto word_at = [&](uint32_t byte_pos) -> uint32_t {
    return byte_pos < 256 ? wave::read_lane(head, byte_pos >> 2) : wave::uniform(*(const uint32_t*)(src + byte_pos));
  };
  const uint32_t first = word_at(0);
  if (first == kRawMarker) {
    if (avail < 4 + bytes) {
      return kSubBad;
    }
    copy_bytes(dst, src + 4, bytes);
    return kSubOk;
  }
  if (first != n || avail < 4 + 4 * num_rles) {
    return kSubBad;
  }
  /* The layer table -- counts, run-pool offsets, identity flags, stream offsets, one entry per RLE layer -- lives in four
   * vector registers, entry l in lane l (v_readlane / v_writelane with the wave-uniform layer number): private arrays
   * indexed at run time would go to scratch memory, and until round 5 the table sat in LDS (a dependent LDS round trip
   * per look-up, two fences per sub-chunk: 15 % of the decoder's time went into this prologue). */
  uint32_t counts_v = 0, run_off_v = 0, ident_v = 0, src_off_v = 0;
  auto counts = [&](uint32_t l) { return wave::read_lane(counts_v, l); };
  auto run_off = [&](uint32_t l) { return wave::read_lane(run_off_v, l); };
  auto ident = [&](uint32_t l) { return wave::read_lane(ident_v, l); };
  auto src_off = [&](uint32_t l) { return wave::read_lane(src_off_v, l); };
  constexpr uint32_t meta_bytes = 0;
  uint32_t pos = 4;
  uint32_t prev = n;
  for (uint32_t l = 0; l < num_rles; ++l) {
    const uint32_t cl = word_at(pos);
    pos += 4;
    if (cl > prev || (cl == 0 && prev != 0)) {
      return kSubBad;
    }
    prev = cl;
    counts_v = wave::write_lane(counts_v, cl, l);
  }
  /* walk the run-stream headers: a layer whose runs are all 1 (bits 0, minimum 1, as many runs as outputs) is the
   * identity -- its values pass through and it needs neither its runs nor an expansion buffer */
  uint32_t pool_used = 0;    /* run entries that must be unpacked */
  uint32_t inner_real = 0;   /* expanding layers below the outermost one: they expand into LDS */
  uint32_t marks_elems = 0;  /* largest expansion target */
  bool all_short = true;     /* every expanding layer qualifies for rle_expand_direct / rle_expand_inplace */
  for (uint32_t l = 0; l < num_rles; ++l) {
    if (avail - pos < 12) {
      return kSubBad;
    }
    const uint32_t bits = word_at(pos);
    const uint32_t mn_lo = word_at(pos + 4);
    const uint32_t mn_hi = word_at(pos + 8);
    if (bits > 64) {
      return kSubBad;
    }
    const uint32_t cl = counts(l);
    const uint32_t target = l == 0 ? n : counts(l - 1);
    const uint64_t words = ((uint64_t)cl * bits + 31) / 32;
    if ((avail - pos - 12) / 4 < words) {
      return kSubBad;
    }
    const bool id = bits == 0 && mn_lo == 1 && mn_hi == 0 && cl == target;
    all_short = all_short && (id || runs_are_short(bits, mn_lo, mn_hi));
    ident_v = wave::write_lane(ident_v, id ? 1u : 0u, l);
    src_off_v = wave::write_lane(src_off_v, pos, l);
    run_off_v = wave::write_lane(run_off_v, pool_used, l);
    if (!id) {
      pool_used += cl;
      inner_real += l > 0 ? 1u : 0u;
      marks_elems = target > marks_elems ? target : marks_elems;
    }
    pos += 12 + 4 * (uint32_t)words;
  }
  wave::sync();
  /* value buffers: the outermost expanding layer writes to HBM when it is layer 0, so a buffer holds at most
   * counts[0] elements then; without that it holds the n elements of the sub-chunk */
  const bool outer_to_hbm = num_rles != 0 && ident(0) == 0;
  /* all runs short: one value buffer, expanded in place, no pool, no marks */
  const bool direct = all_short && pool_used != 0;
  if (direct) {
    pool_used = 0;
    marks_elems = 0;
    inner_real = 0;
  }
  const uint32_t top = outer_to_hbm ? counts(0) : n;
  const uint32_t val_bytes = align16(top * w);
  const uint32_t n_bufs = inner_real ? 2u : 1u;
  const uint32_t pool_bytes = align16(2 * pool_used);
  const uint32_t marks_bytes = align16(2 * marks_elems);
  if ((uint64_t)meta_bytes + (uint64_t)n_bufs * val_bytes + pool_bytes + marks_bytes > budget) {
    return kSubNeedLds;
  }
  CASC_T(0); /* headers, layer table, LDS plan */
  T* A = (T*)(lds + meta_bytes);
  T* B = (T*)(lds + meta_bytes + val_bytes); /* only touched when n_bufs == 2 */
  uint16_t* pool = (uint16_t*)(lds + meta_bytes + n_bufs * val_bytes);
  uint16_t* marks = (uint16_t*)(lds + meta_bytes + n_bufs * val_bytes + pool_bytes);
  for (uint32_t l = 0; l < num_rles; ++l) {
    if (ident(l) || direct) {
      continue;
    }
    uint32_t used;
    if (!unpack_stream<uint16_t>(src + src_off(l), avail - src_off(l), pool + run_off(l), counts(l), 2, used)) {
      return kSubBad;
    }
  }
  CASC_T(1); /* run pools unpacked */
  uint32_t c = num_rles ? counts(num_rles - 1) : n;
  {
    uint32_t used;
    if (!unpack_stream<T>(src + pos, avail - pos, A, c, w, used)) {
      return kSubBad;
    }
  }
  wave::sync();
  CASC_T(2); /* values unpacked */
  const uint32_t layers = num_rles > num_deltas ? num_rles : num_deltas;
  T* cur = A;
  T* oth = B;
  bool in_hbm = false;
  for (uint32_t l = layers; l-- > 0;) {
    if (l < num_deltas) {
      /* the last thing done to the sub-chunk (no expansion behind layer 0's delta): straight into memory */
      const bool last = l == 0 && (num_rles == 0 || ident(0));
      delta_decode(cur, c, last ? (T*)dst : cur);
      in_hbm = in_hbm || last;
      CASC_T(3); /* delta */
    }
    if (l < num_rles && !ident(l)) {
      const uint32_t target = l == 0 ? n : counts(l - 1);
      /* layer 0 writes the sub-chunk itself */
      if (direct) {
        const uint32_t so = src_off(l);
        const uint32_t bits = word_at(so);
        const uint32_t mn = word_at(so + 4);
        const uint32_t* packed = (const uint32_t*)(src + so + 12);
        if (l == 0 ? !rle_expand_direct(cur, packed, bits, mn, c, (T*)dst, target)
                   : !rle_expand_inplace(cur, packed, bits, mn, c, target)) {
          return kSubBad;
        }
        in_hbm = l == 0;
        c = target;
        CASC_T(4 + (l == 0 ? 1 : 0)); /* direct expansion: inner layers / the outermost one (to memory) */
        continue;
      }
      if (!rle_decode(cur, pool + run_off(l), c, l == 0 ? (T*)dst : oth, target, marks)) {
        return kSubBad;
      }
      in_hbm = l == 0;
      c = target;
      T* t = cur;
      cur = oth;
      oth = t;
      CASC_T(6); /* pool + marks expansion */
    }
  }
  if (!in_hbm) {
    T* out = (T*)dst;
    for (uint32_t i = lane; i < n; i += 64) {
      out[i] = cur[i];
    }
  }
  CASC_T(7); /* copy out */
  return kSubOk;
}

/* Worst case of compress_sub: the value buffer of n elements and one run pool of n entries (none without RLE). */
__host__ __device__ inline uint32_t compress_lds_per_wave(uint32_t sub_bytes, uint32_t width, uint32_t num_rles)
{
  const uint32_t n = sub_bytes / width;
  return 64 + ((sub_bytes + 15u) & ~15u) + (num_rles ? ((2 * n + 15u) & ~15u) : 0u);
}

/* Worst case of decompress_sub: LayerMeta, two value buffers, num_rles run pools, the marks of an expansion. */
__host__ __device__ inline uint32_t decompress_lds_per_wave(uint32_t sub_bytes, uint32_t width, uint32_t num_rles)
{
  const uint32_t n = sub_bytes / width;
  return ((uint32_t)((sizeof(LayerMeta) + 15u) & ~15u)) + 2 * ((sub_bytes + 15u) & ~15u) + ((2 * num_rles * n + 15u) & ~15u)
         + (num_rles ? ((2 * n + 15u) & ~15u) : 0u);
}

} // namespace casc
