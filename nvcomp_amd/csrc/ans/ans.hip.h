/*
 * ans/ans.hip.h -- order-0 rANS over bytes, one wavefront per chunk, 64 interleaved states.
 *
 * Reference behaviour: nvcompBatchedANS* is an entropy coder for byte data with a single
 * format type (benchmarks/benchmark_ans_chunked.cu:29-52); its bitstream is closed, so the
 * layout below is this library's own.
 *
 *   chunk := 'A' 'N' 'S' 0x01 | u32 n_bytes | u8 mode | 0 0 0 | body
 *   mode 0 (stored):  body = the n_bytes raw bytes
 *   mode 1 (rANS):    body = u32 n_words | u16 freq[256] (sum 1024) | u32 state[128] | u16 words[n_words]
 *
 * Symbol i of the chunk belongs to lane (i % 256) / 4; a lane codes 4 consecutive bytes of every 256-byte
 * group, so both directions move whole dwords, coalesced over the wave. Every lane runs TWO rANS states
 * (32 bit, lower bound 2^16, 10-bit probabilities, 16-bit renormalisation words): state A codes the even
 * groups, state B the odd ones, and a row of the coder is one byte of an even group and the byte of the odd
 * group behind it -- two independent dependency chains per lane, which is what hides the LDS latency of the
 * table lookups (the decoder spent 72 % of its wave cycles waiting with one chain). The encoder walks the
 * rows from the last to the first; in a row the lanes that must renormalise append their words to the stream,
 * A's in lane order, then B's (ballot + prefix count). The decoder starts from the stored states at the end
 * of the word stream and walks the rows forward, taking the same groups back. A chunk is stored when coding
 * would not make it smaller, so no output exceeds n_bytes + 12.
 *
 * LDS per wave: compress 4 KiB (four histograms, then the symbol table in their place); decompress 4 KiB decode
 * table (one dword per slot: symbol | freq << 8 | (slot - start) << 20) + 1 KiB stream ring +
 * the cumulative table.
 */
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common/wave.h"

namespace ans {

#ifndef NVCOMP_ANS_PROB_BITS
#define NVCOMP_ANS_PROB_BITS 10 /* format constant (profiles/archive/r01_ans_prob_bits.json); other values are for A/B builds only */
#endif
constexpr uint32_t kProbBits = NVCOMP_ANS_PROB_BITS;
constexpr uint32_t kProbScale = 1u << kProbBits;
constexpr uint32_t kStateLow = 1u << 16;
constexpr uint32_t kHeaderBytes = 12;
constexpr uint32_t kFreqOffset = 16;
constexpr uint32_t kStateOffset = kFreqOffset + 512;
constexpr uint32_t kWordsOffset = kStateOffset + 512; /* 64 A states, then 64 B states */
constexpr uint32_t kMinCodedBytes = 2048; /* smaller chunks are always stored */
constexpr uint32_t kRingWords = 512;
constexpr uint32_t kHistCopies = 4; /* the lanes spread their LDS atomics over this many histograms */
constexpr uint32_t kEncodeLds = 1024 * kHistCopies; /* >= the 2 KiB symbol table that replaces the histograms */
constexpr uint32_t kDecodeLds = kProbScale * 4 + kRingWords * 2; /* 5 KiB: 8 workgroups of 4 waves per CU */
static_assert(kRingWords * 2 >= 2 * 257 + 2, "the cumulative frequencies borrow the ring's space while the table is built");
constexpr uint32_t kErrNone = 0;
constexpr uint32_t kErrInput = 1;
constexpr uint32_t kErrOutput = 2;

__host__ __device__ inline size_t max_compressed_bytes(size_t n)
{
  return (n + kHeaderBytes + 7) & ~(size_t)7;
}

template <class T>
__device__ __forceinline__ T load_as(const uint8_t* p)
{
  T v;
  __builtin_memcpy(&v, p, sizeof(T));
  return v;
}

template <class T>
__device__ __forceinline__ void store_as(uint8_t* p, T v)
{
  __builtin_memcpy(p, &v, sizeof(T));
}

/* The 4 bytes lane `lane` codes in the group starting at byte `g` (missing bytes read as 0). */
__device__ __forceinline__ uint32_t load_group_dword(const uint8_t* src, uint32_t n, uint32_t g, uint32_t lane)
{
  const uint32_t at = g + 4 * lane;
  if (at + 4 <= n) {
    return load_as<uint32_t>(src + at);
  }
  uint32_t v = 0;
  for (uint32_t r = 0; r < 4; ++r) {
    if (at + r < n) {
      v |= (uint32_t)src[at + r] << (8 * r);
    }
  }
  return v;
}

__device__ __forceinline__ void write_header(uint8_t* dst, uint32_t n, uint32_t mode)
{
  dst[0] = 'A';
  dst[1] = 'N';
  dst[2] = 'S';
  dst[3] = 1;
  store_as<uint32_t>(dst + 4, n);
  dst[8] = (uint8_t)mode;
  dst[9] = 0;
  dst[10] = 0;
  dst[11] = 0;
}

__device__ __forceinline__ uint32_t store_raw(const uint8_t* __restrict__ src, uint32_t n, uint8_t* __restrict__ dst)
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  if (lane == 0) {
    write_header(dst, n, 0);
  }
  for (uint32_t i = lane; i < n; i += 64) {
    dst[kHeaderBytes + i] = src[i];
  }
  return kHeaderBytes + n;
}

/* Scale the histogram to kProbScale keeping every present symbol >= 1. Lane l holds the
 * counts of symbols 4l..4l+3 in c[] and receives their frequencies in f[]. Deterministic:
 * floor(c * 1024 / n) floored at 1, then the surplus or deficit goes to / comes from the most
 * frequent symbol (lowest index on ties), repeatedly if it cannot absorb all of it. */
__device__ __forceinline__ void normalise(const uint32_t c[4], uint32_t n, uint32_t f[4])
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  uint32_t sum = 0;
#pragma unroll
  for (uint32_t j = 0; j < 4; ++j) {
    const uint32_t q = (uint32_t)(((uint64_t)c[j] << kProbBits) / n);
    f[j] = c[j] == 0 ? 0u : (q == 0 ? 1u : q);
    sum += f[j];
  }
  sum = wave::reduce_add(sum);
  while (sum != kProbScale) {
    uint32_t key = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
      const uint32_t k = (f[j] << 8) | (255u - (4 * lane + j));
      key = k > key ? k : key;
    }
    key = wave::reduce_max(key);
    const uint32_t sym = 255u - (key & 255u);
    const uint32_t top = key >> 8;
    uint32_t now;
    if (sum < kProbScale) {
      now = top + (kProbScale - sum);
      sum = kProbScale;
    } else {
      const uint32_t excess = sum - kProbScale;
      const uint32_t take = excess < top - 1 ? excess : top - 1;
      now = top - take;
      sum -= take;
    }
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
      if (4 * lane + j == sym) {
        f[j] = now;
      }
    }
  }
}

/* Exclusive prefix sums of the frequencies: start[j] for the lane's 4 symbols. */
__device__ __forceinline__ void cumulate(const uint32_t f[4], uint32_t start[4])
{
  const uint32_t mine = f[0] + f[1] + f[2] + f[3];
  uint32_t run = wave::scan_add_inclusive(mine) - mine;
#pragma unroll
  for (uint32_t j = 0; j < 4; ++j) {
    start[j] = run;
    run += f[j];
  }
}

/* ---- compress ------------------------------------------------------------------ */

/* `lds`: kEncodeLds bytes of this wave. Returns the compressed size. */
__device__ __forceinline__ uint32_t encode_chunk(
    const uint8_t* __restrict__ src, uint32_t n, uint8_t* __restrict__ dst, uint8_t* lds)
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  if (n < kMinCodedBytes) {
    return store_raw(src, n, dst);
  }
  uint32_t* table = (uint32_t*)lds; /* histograms, then per symbol freq | start << 16 */
#pragma unroll
  for (uint32_t j = 0; j < 4 * kHistCopies; ++j) {
    table[64 * j + lane] = 0;
  }
  wave::sync();
  uint32_t* hist = table + 256 * (lane % kHistCopies);
  const uint32_t groups = (n + 255) / 256;
  for (uint32_t q0 = 0; q0 < groups; q0 += 4) { /* 4 groups of loads in flight */
    uint32_t v4[4];
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
      v4[u] = q0 + u < groups ? load_group_dword(src, n, 256 * (q0 + u), lane) : 0u;
    }
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
#pragma unroll
      for (uint32_t r = 0; r < 4; ++r) {
        if (256 * (q0 + u) + 4 * lane + r < n) {
          atomicAdd(&hist[(v4[u] >> (8 * r)) & 255u], 1u);
        }
      }
    }
  }
  wave::sync();
  uint32_t c[4], f[4], start[4];
#pragma unroll
  for (uint32_t j = 0; j < 4; ++j) {
    c[j] = 0;
    for (uint32_t k = 0; k < kHistCopies; ++k) {
      c[j] += table[256 * k + 4 * lane + j];
    }
  }
  normalise(c, n, f);
  cumulate(f, start);
  wave::sync();
  /* per symbol: { freq | start << 12 | log2ceil(freq) << 24, magic } -- x / freq by multiplication
   * (Granlund & Montgomery, "Division by invariant integers using multiplication", PLDI 1994, fig. 4.1) */
#pragma unroll
  for (uint32_t j = 0; j < 4; ++j) {
    const uint32_t d = f[j] ? f[j] : 1u;
    const uint32_t l = d > 1 ? 32u - (uint32_t)__builtin_clz(d - 1) : 0u;
    const uint32_t magic = (uint32_t)((((uint64_t)((1u << l) - d)) << 32) / d) + 1u;
    table[2 * (4 * lane + j)] = f[j] | (start[j] << 12) | (l << 24);
    table[2 * (4 * lane + j) + 1] = magic;
    store_as<uint16_t>(dst + kFreqOffset + 2 * (4 * lane + j), (uint16_t)f[j]);
  }
  wave::sync();

  const uint32_t limit_words = (n + kHeaderBytes - kWordsOffset) / 2; /* coded form must stay below the stored size */
  uint8_t* words = dst + kWordsOffset;
  uint32_t xs[2] = {kStateLow, kStateLow}; /* A: even groups, B: odd groups */
  uint32_t p = 0;
  const uint32_t pairs = (groups + 1) / 2;
  /* the input dwords of the next pair of groups are fetched while this one is coded */
  uint32_t v_next[2];
  v_next[0] = load_group_dword(src, n, 256 * (2 * (pairs - 1)), lane);
  v_next[1] = 2 * pairs - 1 < groups ? load_group_dword(src, n, 256 * (2 * pairs - 1), lane) : 0u;
  for (uint32_t q = pairs; q-- > 0;) {
    const uint32_t v[2] = {v_next[0], v_next[1]};
    if (q > 0) {
      v_next[0] = load_group_dword(src, n, 256 * (2 * q - 2), lane);
      v_next[1] = load_group_dword(src, n, 256 * (2 * q - 1), lane);
    }
#pragma unroll
    for (uint32_t rr = 0; rr < 4; ++rr) {
      const uint32_t r = 3 - rr;
      bool active[2], emit[2];
      uint32_t e[2], magic[2];
#pragma unroll
      for (uint32_t h = 0; h < 2; ++h) {
        active[h] = 256 * (2 * q + h) + 4 * lane + r < n;
        const uint32_t sym = (v[h] >> (8 * r)) & 255u;
        e[h] = table[2 * sym];
        magic[h] = table[2 * sym + 1];
        emit[h] = active[h] && (xs[h] >> (32 - kProbBits)) >= (e[h] & 0xfffu); /* x >= freq << (32 - kProbBits) */
      }
      const uint64_t m0 = wave::ballot(emit[0]);
      const uint64_t m1 = wave::ballot(emit[1]);
      const uint32_t c0 = wave::popc64(m0);
      const uint32_t c1 = wave::popc64(m1);
      if (p + c0 + c1 >= limit_words) {
        return store_raw(src, n, dst);
      }
      if (emit[0]) {
        store_as<uint16_t>(words + 2 * (p + wave::prefix_popc(m0)), (uint16_t)xs[0]);
        xs[0] >>= 16;
      }
      if (emit[1]) {
        store_as<uint16_t>(words + 2 * (p + c0 + wave::prefix_popc(m1)), (uint16_t)xs[1]);
        xs[1] >>= 16;
      }
      p += c0 + c1;
#pragma unroll
      for (uint32_t h = 0; h < 2; ++h) {
        if (active[h]) {
          const uint32_t freq = e[h] & 0xfffu;
          const uint32_t base = (e[h] >> 12) & 0xfffu;
          const uint32_t l = e[h] >> 24;
          const uint32_t t = __umulhi(magic[h], xs[h]);
          const uint32_t quot = (t + ((xs[h] - t) >> (l ? 1u : 0u))) >> (l ? l - 1u : 0u);
          xs[h] = (quot << kProbBits) + (xs[h] - wave::mul24(quot, freq)) + base; /* quot < 2^22 after renormalisation */
        }
      }
    }
  }
  store_as<uint32_t>(dst + kStateOffset + 4 * lane, xs[0]);
  store_as<uint32_t>(dst + kStateOffset + 256 + 4 * lane, xs[1]);
  if (lane == 0) {
    write_header(dst, n, 1);
    store_as<uint32_t>(dst + 12, p);
  }
  return kWordsOffset + 2 * p;
}

/* ---- decompress ---------------------------------------------------------------- */

struct WordRing
{
  const uint8_t* words; /* global: the chunk's word stream */
  uint16_t* ring;       /* LDS: kRingWords entries, word i at i % kRingWords */
  uint32_t n_words;
  uint32_t lo; /* lowest resident word index (multiple of 128) */
};

/* Keep [p - 64, p) resident, fetching 128-word blocks well ahead of their use. */
__device__ __forceinline__ void ring_fill(WordRing& w, uint32_t p)
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  bool loaded = false;
  if (w.lo > 0 && p < w.lo + 320) {
    wave::sync(); /* every lane has taken its words of the last row: a refill may reuse slots just above p */
  }
  while (w.lo > 0 && p < w.lo + 320) {
    w.lo -= 128;
    const uint32_t i = w.lo + 2 * lane;
    uint32_t v = 0;
    if (i + 2 <= w.n_words) {
      v = load_as<uint32_t>(w.words + 2 * i);
    } else if (i < w.n_words) {
      v = load_as<uint16_t>(w.words + 2 * i);
    }
    *(uint32_t*)(w.ring + (i & (kRingWords - 1))) = v;
    loaded = true;
  }
  if (loaded) {
    wave::sync();
  }
}

/* `lds`: kDecodeLds bytes of this wave, 16-byte aligned. */
__device__ __forceinline__ uint32_t decode_chunk(
    const uint8_t* __restrict__ in, uint32_t in_len, uint8_t* __restrict__ out, uint32_t out_cap, uint8_t* lds, uint32_t& err)
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  err = kErrNone;
  if (in_len < kHeaderBytes) {
    err = kErrInput;
    return 0;
  }
  const uint32_t magic = wave::uniform(load_as<uint32_t>(in));
  const uint32_t n = wave::uniform(load_as<uint32_t>(in + 4));
  const uint32_t mode = wave::uniform(load_as<uint32_t>(in + 8));
  if (magic != 0x01534e41u || mode > 1) {
    err = kErrInput;
    return 0;
  }
  if (n > out_cap) {
    err = kErrOutput;
    return 0;
  }
  if (mode == 0) {
    if (in_len - kHeaderBytes < n) {
      err = kErrInput;
      return 0;
    }
    for (uint32_t i = lane; i < n; i += 64) {
      out[i] = in[kHeaderBytes + i];
    }
    return n;
  }
  if (in_len < kWordsOffset) {
    err = kErrInput;
    return 0;
  }
  const uint32_t n_words = wave::uniform(load_as<uint32_t>(in + 12));
  if ((in_len - kWordsOffset) / 2 < n_words) {
    err = kErrInput;
    return 0;
  }

  uint32_t* table = (uint32_t*)lds;
  uint16_t* ring = (uint16_t*)(lds + kProbScale * 4);
  uint16_t* cum = ring; /* 257 entries, only while the table is built: the ring is filled afterwards */

  /* decode table from the frequencies */
  uint32_t f[4], start[4];
  uint32_t sum = 0;
#pragma unroll
  for (uint32_t j = 0; j < 4; ++j) {
    f[j] = load_as<uint16_t>(in + kFreqOffset + 2 * (4 * lane + j));
    sum += f[j];
  }
  if (wave::reduce_add(sum) != kProbScale) {
    err = kErrInput;
    return 0;
  }
  cumulate(f, start);
#pragma unroll
  for (uint32_t j = 0; j < 4; ++j) {
    cum[4 * lane + j] = (uint16_t)start[j];
  }
  if (lane == 0) {
    cum[256] = (uint16_t)kProbScale;
  }
  wave::sync();
  for (uint32_t slot = lane; slot < kProbScale; slot += 64) {
    /* the symbol whose range holds the slot: last s with cum[s] <= slot */
    uint32_t s = 0;
#pragma unroll
    for (uint32_t step = 128; step != 0; step >>= 1) {
      if (cum[s + step] <= slot) {
        s += step;
      }
    }
    const uint32_t lo = cum[s];
    table[slot] = s | ((cum[s + 1] - lo) << 8) | ((slot - lo) << 20);
  }

  wave::sync(); /* the last reads of cum[] precede the first ring words */

  WordRing w;
  w.words = in + kWordsOffset;
  w.ring = ring;
  w.n_words = n_words;
  w.lo = (n_words + 127) & ~127u;
  uint32_t p = n_words;
  ring_fill(w, p);
  wave::sync();

  uint32_t xs[2];
  xs[0] = load_as<uint32_t>(in + kStateOffset + 4 * lane);
  xs[1] = load_as<uint32_t>(in + kStateOffset + 256 + 4 * lane);
  const uint32_t groups = (n + 255) / 256;
  const uint32_t pairs = (groups + 1) / 2;
  uint32_t underflow = 0; /* uniform; a corrupt stream may ask for more words than there are */
  for (uint32_t q = 0; q < pairs; ++q) {
    const uint32_t at0 = 512 * q + 4 * lane; /* the lane's bytes in the even group; +256 in the odd one */
    if (512 * q + 512 <= n) {
      /* whole pair of groups: every lane decodes 4 + 4 symbols, two independent chains */
      uint32_t packed0 = 0, packed1 = 0;
#pragma unroll
      for (uint32_t r = 0; r < 4; ++r) {
        const uint32_t e0 = table[xs[0] & (kProbScale - 1)];
        const uint32_t e1 = table[xs[1] & (kProbScale - 1)];
        xs[0] = __umul24((e0 >> 8) & 0xfffu, xs[0] >> kProbBits) + (e0 >> 20); /* 12 x 22 bits */
        xs[1] = __umul24((e1 >> 8) & 0xfffu, xs[1] >> kProbBits) + (e1 >> 20);
        const bool need0 = xs[0] < kStateLow;
        const bool need1 = xs[1] < kStateLow;
        const uint64_t m0 = wave::ballot(need0);
        const uint64_t m1 = wave::ballot(need1);
        const uint32_t c0 = wave::popc64(m0);
        const uint32_t cnt = c0 + wave::popc64(m1);
        underflow |= cnt > p ? 1u : 0u;
        p -= cnt;
        if (need0) {
          xs[0] = (xs[0] << 16) | ring[(p + wave::prefix_popc(m0)) & (kRingWords - 1)];
        }
        if (need1) {
          xs[1] = (xs[1] << 16) | ring[(p + c0 + wave::prefix_popc(m1)) & (kRingWords - 1)];
        }
        packed0 |= (e0 & 255u) << (8 * r);
        packed1 |= (e1 & 255u) << (8 * r);
        if (r == 1) {
          ring_fill(w, p); /* two rows take at most 256 words; the ring is kept 320 words ahead */
        }
      }
      store_as<uint32_t>(out + at0, packed0);
      store_as<uint32_t>(out + at0 + 256, packed1);
    } else {
#pragma unroll
      for (uint32_t r = 0; r < 4; ++r) {
        bool active[2], need[2];
        uint32_t e[2], nx[2];
#pragma unroll
        for (uint32_t h = 0; h < 2; ++h) {
          active[h] = at0 + 256 * h + r < n;
          e[h] = table[xs[h] & (kProbScale - 1)];
          nx[h] = __umul24((e[h] >> 8) & 0xfffu, xs[h] >> kProbBits) + (e[h] >> 20);
          need[h] = active[h] && nx[h] < kStateLow;
        }
        const uint64_t m0 = wave::ballot(need[0]);
        const uint64_t m1 = wave::ballot(need[1]);
        const uint32_t c0 = wave::popc64(m0);
        const uint32_t cnt = c0 + wave::popc64(m1);
        underflow |= cnt > p ? 1u : 0u;
        p -= cnt;
        if (need[0]) {
          nx[0] = (nx[0] << 16) | ring[(p + wave::prefix_popc(m0)) & (kRingWords - 1)];
        }
        if (need[1]) {
          nx[1] = (nx[1] << 16) | ring[(p + c0 + wave::prefix_popc(m1)) & (kRingWords - 1)];
        }
#pragma unroll
        for (uint32_t h = 0; h < 2; ++h) {
          if (active[h]) {
            xs[h] = nx[h];
            out[at0 + 256 * h + r] = (uint8_t)e[h];
          }
        }
        if (r == 1) {
          ring_fill(w, p);
        }
      }
    }
    if (underflow) {
      err = kErrInput;
      return 0;
    }
    ring_fill(w, p);
  }
  /* a valid stream is consumed exactly and every state is back at its start value */
  if (p != 0 || wave::ballot(xs[0] != kStateLow || xs[1] != kStateLow)) {
    err = kErrInput;
    return 0;
  }
  return n;
}

} // namespace ans
