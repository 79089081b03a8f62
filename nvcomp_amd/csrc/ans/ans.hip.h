/*
 * ans/ans.hip.h -- order-0 rANS over bytes, one wavefront per chunk, 64 interleaved states.
 *
 * Reference behaviour: nvcompBatchedANS* is an entropy coder for byte data with a single
 * format type (benchmarks/benchmark_ans_chunked.cu:29-52); its bitstream is closed, so the
 * layout below is this library's own.
 *
 *   chunk := 'A' 'N' 'S' 0x01 | u32 n_bytes | u8 mode | 0 0 0 | body
 *   mode 0 (stored):  body = the n_bytes raw bytes
 *   mode 1 (rANS):    body = u32 n_words | u16 freq[256] (sum 1024) | u32 state[64] | u16 words[n_words]
 *
 * Symbol i of the chunk belongs to lane (i % 256) / 4 and row 4 (i / 256) + i % 4: a lane
 * codes 4 consecutive bytes of every 256-byte group, so both directions move whole dwords,
 * coalesced over the wave. Every lane runs its own rANS state (32 bit, lower bound 2^16,
 * 10-bit probabilities, 16-bit renormalisation words). The encoder walks the rows from the
 * last to the first; in a row the lanes that must renormalise append their words to the
 * stream in lane order (ballot + prefix count). The decoder starts from the stored states at
 * the end of the word stream and walks the rows forward, taking the same groups back. A
 * chunk is stored when coding would not make it smaller, so no output exceeds n_bytes + 12.
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
#define NVCOMP_ANS_PROB_BITS 10 /* format constant (profiles/r01_ans_prob_bits.json); other values are for A/B builds only */
#endif
constexpr uint32_t kProbBits = NVCOMP_ANS_PROB_BITS;
constexpr uint32_t kProbScale = 1u << kProbBits;
constexpr uint32_t kStateLow = 1u << 16;
constexpr uint32_t kHeaderBytes = 12;
constexpr uint32_t kFreqOffset = 16;
constexpr uint32_t kStateOffset = kFreqOffset + 512;
constexpr uint32_t kWordsOffset = kStateOffset + 256;
constexpr uint32_t kMinCodedBytes = 1024; /* smaller chunks are always stored */
constexpr uint32_t kRingWords = 512;
constexpr uint32_t kHistCopies = 4; /* the lanes spread their LDS atomics over this many histograms */
constexpr uint32_t kEncodeLds = 1024 * kHistCopies; /* >= the 2 KiB symbol table that replaces the histograms */
constexpr uint32_t kDecodeLds = kProbScale * 4 + kRingWords * 2 + 528;
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
  const uint32_t lane = (uint32_t)wave::lane_id();
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
  const uint32_t lane = (uint32_t)wave::lane_id();
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
  const uint32_t lane = (uint32_t)wave::lane_id();
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
  uint32_t x = kStateLow;
  uint32_t p = 0;
  /* the input dword of the next group (q - 1) is fetched while this one is coded: the load never sits on the
   * dependent chain of the states */
  uint32_t v_next = load_group_dword(src, n, 256 * (groups - 1), lane);
  for (uint32_t q = groups; q-- > 0;) {
    const uint32_t v = v_next;
    if (q > 0) {
      v_next = load_group_dword(src, n, 256 * (q - 1), lane);
    }
#pragma unroll
    for (uint32_t rr = 0; rr < 4; ++rr) {
      const uint32_t r = 3 - rr;
      const bool active = 256 * q + 4 * lane + r < n;
      const uint32_t sym = (v >> (8 * r)) & 255u;
      const uint32_t e = table[2 * sym];
      const uint32_t magic = table[2 * sym + 1];
      const uint32_t freq = e & 0xfffu;
      const uint32_t base = (e >> 12) & 0xfffu;
      const bool emit = active && (x >> (32 - kProbBits)) >= freq; /* x >= freq << (32 - kProbBits) */
      const uint64_t m = wave::ballot(emit);
      const uint32_t cnt = wave::popc64(m);
      if (p + cnt >= limit_words) {
        return store_raw(src, n, dst);
      }
      if (emit) {
        store_as<uint16_t>(words + 2 * (p + wave::prefix_popc(m)), (uint16_t)x);
        x >>= 16;
      }
      p += cnt;
      if (active) {
        const uint32_t l = e >> 24;
        const uint32_t t = __umulhi(magic, x);
        const uint32_t quot = (t + ((x - t) >> (l ? 1u : 0u))) >> (l ? l - 1u : 0u);
        x = (quot << kProbBits) + (x - quot * freq) + base;
      }
    }
  }
  store_as<uint32_t>(dst + kStateOffset + 4 * lane, x);
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
  const uint32_t lane = (uint32_t)wave::lane_id();
  bool loaded = false;
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
  const uint32_t lane = (uint32_t)wave::lane_id();
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
  uint16_t* cum = (uint16_t*)(lds + kProbScale * 4 + kRingWords * 2); /* 257 entries */

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

  WordRing w;
  w.words = in + kWordsOffset;
  w.ring = ring;
  w.n_words = n_words;
  w.lo = (n_words + 127) & ~127u;
  uint32_t p = n_words;
  ring_fill(w, p);
  wave::sync();

  uint32_t x = load_as<uint32_t>(in + kStateOffset + 4 * lane);
  const uint32_t groups = (n + 255) / 256;
  uint32_t underflow = 0; /* uniform; a corrupt stream may ask for more words than there are */
  for (uint32_t q = 0; q < groups; ++q) {
    const uint32_t at = 256 * q + 4 * lane;
    uint32_t packed = 0;
    if (256 * q + 256 <= n) {
      /* whole group: every lane decodes 4 symbols */
#pragma unroll
      for (uint32_t r = 0; r < 4; ++r) {
        const uint32_t e = table[x & (kProbScale - 1)];
        x = __umul24((e >> 8) & 0xfffu, x >> kProbBits) + (e >> 20); /* 12 x 22 bits */
        const bool need = x < kStateLow;
        const uint64_t m = wave::ballot(need);
        const uint32_t cnt = wave::popc64(m);
        underflow |= cnt > p ? 1u : 0u;
        p -= cnt;
        if (need) {
          x = (x << 16) | ring[(p + wave::prefix_popc(m)) & (kRingWords - 1)];
        }
        packed |= (e & 255u) << (8 * r);
      }
      store_as<uint32_t>(out + at, packed);
    } else {
#pragma unroll
      for (uint32_t r = 0; r < 4; ++r) {
        const bool active = at + r < n;
        const uint32_t e = table[x & (kProbScale - 1)];
        uint32_t nx = __umul24((e >> 8) & 0xfffu, x >> kProbBits) + (e >> 20);
        const bool need = active && nx < kStateLow;
        const uint64_t m = wave::ballot(need);
        const uint32_t cnt = wave::popc64(m);
        underflow |= cnt > p ? 1u : 0u;
        p -= cnt;
        if (need) {
          nx = (nx << 16) | ring[(p + wave::prefix_popc(m)) & (kRingWords - 1)];
        }
        if (active) {
          x = nx;
          out[at + r] = (uint8_t)e;
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
  if (p != 0 || wave::ballot(x != kStateLow)) {
    err = kErrInput;
    return 0;
  }
  return n;
}

} // namespace ans
