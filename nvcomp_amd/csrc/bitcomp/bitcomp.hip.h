/*
 * bitcomp/bitcomp.hip.h -- Bitcomp codec, one wavefront per chunk.
 *
 * Scheme (reference behaviour: a delta + bit-packing compressor for numerical data with
 * a default and a "sparse" algorithm, benchmarks/benchmark_bitcomp_chunked.cu:32-60; the
 * bitstream itself is undocumented, README.md:13, so this layout is our own):
 *
 *   chunk  := header(12 B) block* tail
 *   header := 'B' 'T' 'C' 0x01 | u8 algo | u8 log2(S) | u16 0 | u32 n_bytes        (S = element size)
 *   block  := covers up to 2048 elements = R rows of 64 (R = ceil(count / 64))
 *             either FF 00 00 00                        (every value of the block is zero)
 *             or     R width bytes, zero-padded to a multiple of 4,
 *                    D x 64 dwords, D = ceil(sum(widths) / 32)
 *   tail   := the n_bytes % S bytes that do not form an element, raw
 *
 * For 4- and 8-byte elements row r of a block holds elements 64 r + l, l = lane; for 1- and 2-byte elements a
 * lane owns E = 4 / S consecutive elements of a row (a dword), so a row is 64 E elements, a block 2048 E, and the
 * lane's bit string carries its E values of a row one after the other. Its values (algo 0: zigzag of the
 * difference to the previous element of the chunk, modulo 2^(8S); algo 1: the element
 * itself) are stored with width[r] = bits of the row's largest value. Lane l owns its own
 * bit string -- its value of row 0, then row 1, ... LSB first -- cut into dwords; dword k
 * of lane l sits at dword index 64 k + l of the block's payload. Every load and store of
 * both directions is therefore a full-wave coalesced access and packing needs no
 * cross-lane traffic; the only wave operations are the per-row max (compress) and the
 * per-row prefix sum that undoes the delta (decompress). Elements past the end of the last
 * row are encoded as value 0.
 */
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common/wave.h"

namespace bitcomp {

constexpr uint32_t kHeaderBytes = 12;
constexpr uint32_t kRows = 32;
constexpr uint32_t kBlockElems = 64 * kRows;
constexpr uint32_t kErrNone = 0;
constexpr uint32_t kErrInput = 1;
constexpr uint32_t kErrOutput = 2;

/* elements of one lane in one row: a dword's worth for 1- and 2-byte types */
__host__ __device__ inline uint32_t lane_elems(uint32_t elem_size)
{
  return elem_size < 4 ? 4 / elem_size : 1;
}

/* Worst case: every row at full width. Host and device. */
__host__ __device__ inline size_t block_bound(size_t rows, uint32_t elem_size)
{
  return ((rows + 3) & ~(size_t)3) + (rows * 8 * elem_size * lane_elems(elem_size) + 31) / 32 * 256;
}

__host__ __device__ inline size_t max_compressed_bytes(size_t n, uint32_t elem_size)
{
  const size_t nelem = n / elem_size;
  const size_t row_elems = 64 * lane_elems(elem_size);
  const size_t block_elems = row_elems * kRows;
  const size_t full = nelem / block_elems;
  const size_t rest = nelem % block_elems;
  return kHeaderBytes + full * block_bound(kRows, elem_size)
         + (rest ? block_bound((rest + row_elems - 1) / row_elems, elem_size) : 0) + n % elem_size;
}

template <class T>
__device__ __forceinline__ T load_elem(const uint8_t* p)
{
  T v;
  __builtin_memcpy(&v, p, sizeof(T));
  return v;
}

template <class T>
__device__ __forceinline__ void store_elem(uint8_t* p, T v)
{
  __builtin_memcpy(p, &v, sizeof(T));
}

__device__ __forceinline__ uint32_t load_u32(const uint8_t* p)
{
  return load_elem<uint32_t>(p);
}

template <class T>
__device__ __forceinline__ T zigzag(T d)
{
  constexpr uint32_t W = sizeof(T) * 8;
  return (T)((T)(d << 1) ^ (T)(0 - (T)(d >> (W - 1))));
}

template <class T>
__device__ __forceinline__ T unzigzag(T z)
{
  return (T)((T)(z >> 1) ^ (T)(0 - (T)(z & 1)));
}

template <class T>
__device__ __forceinline__ uint32_t bit_width(T z)
{
  if (sizeof(T) == 8) {
    return z ? 64u - (uint32_t)__builtin_clzll((unsigned long long)z) : 0u;
  }
  const uint32_t v = (uint32_t)z;
  return v ? 32u - (uint32_t)__builtin_clz(v) : 0u;
}

/* Inclusive prefix sum over the wave in T's modular arithmetic. */
template <class T>
__device__ __forceinline__ T scan_add(T v)
{
  if (sizeof(T) == 8) {
    const uint32_t lane = (uint32_t)wave::lane_id();
    uint64_t x = (uint64_t)v;
#pragma unroll
    for (uint32_t d = 1; d < 64; d *= 2) {
      const uint32_t lo = wave::shuffle((uint32_t)x, (lane - d) & 63u);
      const uint32_t hi = wave::shuffle((uint32_t)(x >> 32), (lane - d) & 63u);
      if (lane >= d) {
        x += ((uint64_t)hi << 32) | lo;
      }
    }
    return (T)x;
  }
  return (T)wave::scan_add_inclusive((uint32_t)v);
}

template <class T>
__device__ __forceinline__ T last_lane(T v)
{
  if (sizeof(T) == 8) {
    const uint64_t x = (uint64_t)v;
    return (T)(((uint64_t)wave::read_lane((uint32_t)(x >> 32), 63) << 32) | wave::read_lane((uint32_t)x, 63));
  }
  return (T)wave::read_lane((uint32_t)v, 63);
}

/* the value of the lane below (lane 0: unspecified) */
template <class T>
__device__ __forceinline__ T lane_below(T v)
{
  if (sizeof(T) == 8) {
    const uint64_t x = (uint64_t)v;
    return (T)(((uint64_t)wave::prev_lane((uint32_t)(x >> 32)) << 32) | wave::prev_lane((uint32_t)x));
  }
  return (T)wave::prev_lane((uint32_t)v);
}

__device__ __forceinline__ uint32_t pad4(uint32_t n)
{
  return (n + 3u) & ~3u;
}

/* ---- compress ------------------------------------------------------------------ */

/* Lane-local bit string writer: dword k of this lane goes to payload[(64 k + lane) * 4]. */
struct BitWriter
{
  uint8_t* payload;
  uint64_t acc;
  uint32_t fill; /* wave-uniform */
  uint32_t k;    /* wave-uniform */

  __device__ __forceinline__ void put(uint32_t v, uint32_t w) /* w <= 32, v < 2^w */
  {
    acc |= (uint64_t)v << fill;
    fill += w;
    if (fill >= 32) {
      store_elem<uint32_t>(payload + (64u * k + (uint32_t)wave::lane_id()) * 4u, (uint32_t)acc);
      acc >>= 32;
      fill -= 32;
      ++k;
    }
  }
  __device__ __forceinline__ void finish()
  {
    if (fill != 0) {
      store_elem<uint32_t>(payload + (64u * k + (uint32_t)wave::lane_id()) * 4u, (uint32_t)acc);
      ++k;
    }
  }
};

template <class T>
struct PackedRow
{
  using type = T;
};
template <>
struct PackedRow<uint8_t>
{
  using type = uint32_t;
};
template <>
struct PackedRow<uint16_t>
{
  using type = uint32_t;
};

template <class T, bool DELTA>
__device__ __forceinline__ uint32_t encode_chunk(const uint8_t* __restrict__ src, uint32_t n, uint8_t* __restrict__ dst)
{
  constexpr uint32_t S = sizeof(T);
  const uint32_t lane = (uint32_t)wave::lane_id();
  const uint32_t nelem = n / S;
  if (lane == 0) {
    dst[0] = 'B';
    dst[1] = 'T';
    dst[2] = 'C';
    dst[3] = 1;
    dst[4] = DELTA ? 0 : 1;
    dst[5] = S == 1 ? 0 : S == 2 ? 1 : S == 4 ? 2 : 3;
    dst[6] = 0;
    dst[7] = 0;
    store_elem<uint32_t>(dst + 8, n);
  }
  constexpr uint32_t E = S < 4 ? 4 / S : 1; /* elements of a lane per row */
  constexpr uint32_t kRowElems = 64 * E;
  constexpr uint32_t kBlock = kRowElems * kRows;
  using Z = typename PackedRow<T>::type; /* a lane's values of one row: the element itself, or E small ones in a dword */
  uint32_t op = kHeaderBytes;
  Z before = 0; /* whole blocks: the last row's last lane of the block before (its element, or its dword of E elements) */
  for (uint32_t base = 0; base < nelem; base += kBlock) {
    const uint32_t count = nelem - base < kBlock ? nelem - base : kBlock;
    const uint32_t rows = (count + kRowElems - 1) / kRowElems;
    Z z[kRows];
    uint32_t widths = 0; /* lane r: width of row r */
    if (count == kBlock) {
      /* A whole block: its 32 rows are requested at once, and an element's predecessor comes from the lane below (the
       * first lane's: from the row before) instead of from memory. Row by row -- a load or two, their wait, the row's
       * maximum -- a block was 32 round trips to memory, one after the other, and they were the compressor's time: 256 a
       * chunk, 223 us a chunk at eight waves a SIMD (round 6). */
#pragma unroll
      for (uint32_t r = 0; r < kRows; ++r) {
        z[r] = load_elem<Z>(src + (size_t)(base + kRowElems * r + E * lane) * S);
      }
#pragma unroll
      for (uint32_t r = 0; r < kRows; ++r) {
        const Z raw = z[r];
        uint32_t width_here;
        if (E == 1) {
          if (DELTA) {
            const Z up = lane_below<Z>(raw); /* (every lane takes part: not inside the choice below) */
            const Z prev = lane == 0 ? before : up;
            z[r] = (Z)zigzag<T>((T)(raw - prev));
          }
          width_here = bit_width<T>((T)z[r]);
        } else {
          uint32_t packed = (uint32_t)raw;
          if (DELTA) {
            /* the dword one element earlier: this one's low elements behind the last element of the dword below */
            const uint32_t up = lane_below<uint32_t>((uint32_t)raw);
            const uint32_t below = lane == 0 ? (uint32_t)before : up;
            const uint32_t pv = ((uint32_t)raw << (8 * S)) | (below >> (32 - 8 * S));
            packed = 0;
#pragma unroll
            for (uint32_t k = 0; k < E; ++k) {
              const T zz = zigzag<T>((T)((T)((uint32_t)raw >> (8 * S * k)) - (T)(pv >> (8 * S * k))));
              packed |= (uint32_t)zz << (8 * S * k);
            }
          }
          z[r] = (Z)packed;
          uint32_t any = packed;
#pragma unroll
          for (uint32_t k = 1; k < E; ++k) {
            any |= packed >> (8 * S * k);
          }
          width_here = bit_width<T>((T)any);
        }
        before = last_lane<Z>(raw);
        const uint32_t w = wave::reduce_max(width_here);
        widths = lane == r ? w : widths;
        if (E > 1 && DELTA) {
          wave::sched_fence(); /* (the rows' E-element bodies interleaved: 100 spilled registers for one-byte elements) */
        }
      }
    } else {
#pragma unroll
      for (uint32_t r = 0; r < kRows; ++r) {
        z[r] = 0;
        if (r < rows) {
          const uint32_t i = base + kRowElems * r + E * lane;
          uint32_t width_here = 0;
          if (E == 1) {
            if (i < nelem) {
              const T e = load_elem<T>(src + (size_t)i * S);
              if (DELTA) {
                const T prev = i ? load_elem<T>(src + (size_t)(i - 1) * S) : (T)0;
                z[r] = (Z)zigzag<T>((T)(e - prev));
              } else {
                z[r] = (Z)e;
              }
            }
            width_here = bit_width<T>((T)z[r]);
          } else {
            /* E elements from one dword; their predecessors from the dword one element earlier */
            uint32_t packed = 0;
            if (i + E <= nelem) {
              const uint32_t v = load_elem<uint32_t>(src + (size_t)i * S);
              uint32_t pv = 0;
              if (DELTA) {
                pv = i ? load_elem<uint32_t>(src + (size_t)(i - 1) * S) : v << (8 * S);
              }
#pragma unroll
              for (uint32_t k = 0; k < E; ++k) {
                const T e = (T)(v >> (8 * S * k));
                const T zz = DELTA ? zigzag<T>((T)(e - (T)(pv >> (8 * S * k)))) : e;
                packed |= (uint32_t)zz << (8 * S * k);
              }
            } else {
#pragma unroll
              for (uint32_t k = 0; k < E; ++k) {
                if (i + k < nelem) {
                  const T e = load_elem<T>(src + (size_t)(i + k) * S);
                  const T prev = (DELTA && i + k) ? load_elem<T>(src + (size_t)(i + k - 1) * S) : (T)0;
                  const T zz = DELTA ? zigzag<T>((T)(e - prev)) : e;
                  packed |= (uint32_t)zz << (8 * S * k);
                }
              }
            }
            z[r] = (Z)packed;
            uint32_t any = packed; /* the OR of the E values has the width of the largest */
#pragma unroll
            for (uint32_t k = 1; k < E; ++k) {
              any |= packed >> (8 * S * k);
            }
            width_here = bit_width<T>((T)any);
          }
          const uint32_t w = wave::reduce_max(width_here);
          widths = lane == r ? w : widths;
        }
      }
    }
    if (wave::ballot(widths != 0) == 0) { /* constant run (algo 0) / all zero (algo 1) */
      if (lane < 4) {
        dst[op + lane] = lane == 0 ? 0xFF : 0;
      }
      op += 4;
      continue;
    }
    const uint32_t wbytes = pad4(rows);
    if (lane < wbytes) {
      dst[op + lane] = (uint8_t)(lane < rows ? widths : 0);
    }
    BitWriter bw;
    bw.payload = dst + op + wbytes;
    bw.acc = 0;
    bw.fill = 0;
    bw.k = 0;
#pragma unroll
    for (uint32_t r = 0; r < kRows; ++r) {
      if (r < rows) {
        const uint32_t w = wave::read_lane(widths, r);
        if (S == 8) {
          const uint64_t v = (uint64_t)z[r];
          const uint32_t lo_w = w < 32 ? w : 32;
          bw.put((uint32_t)v, lo_w);
          if (w > 32) {
            bw.put((uint32_t)(v >> 32), w - 32);
          }
        } else if (E == 1) {
          bw.put((uint32_t)z[r], w);
        } else {
#pragma unroll
          for (uint32_t k = 0; k < E; ++k) {
            bw.put((uint32_t)(T)((uint32_t)z[r] >> (8 * S * k)), w);
          }
        }
      }
    }
    bw.finish();
    op += wbytes + bw.k * 256u;
  }
  const uint32_t tail = n - nelem * S;
  if (lane < tail) {
    dst[op + lane] = src[nelem * S + lane];
  }
  return op + tail;
}

/* ---- decompress ---------------------------------------------------------------- */

template <class T, bool DELTA, bool CHECKED>
__device__ __forceinline__ uint32_t decode_body(
    const uint8_t* __restrict__ in, uint32_t in_len, uint8_t* __restrict__ out, uint32_t n, uint32_t& err)
{
  constexpr uint32_t S = sizeof(T);
  constexpr uint32_t W = 8 * S;
  constexpr uint32_t E = S < 4 ? 4 / S : 1; /* elements of a lane per row */
  constexpr uint32_t kRowElems = 64 * E;
  constexpr uint32_t kBlock = kRowElems * kRows;
  const uint32_t lane = (uint32_t)wave::lane_id();
  const uint32_t nelem = n / S;
  uint32_t ip = kHeaderBytes;
  T carry = 0;
  /* A block's first bytes (its marker and its row widths, a byte a lane) are requested while the block before it is
   * unpacked, and its payload eight dwords a lane at a time with the next eight under way: a block used to be three
   * dependent round trips to memory (the widths, then each group of eight dwords) with nothing in flight meanwhile -- at
   * eight waves a SIMD a third of a wave's time (round 6). Bytes behind the input are not read. */
  auto first_bytes = [&](uint32_t at) -> uint32_t {
    return at < in_len && in_len - at > lane && lane < kRows ? (uint32_t)in[at + lane] : 0u;
  };
  uint32_t ahead = first_bytes(ip);
  for (uint32_t base = 0; base < nelem; base += kBlock) {
    const uint32_t count = nelem - base < kBlock ? nelem - base : kBlock;
    const uint32_t rows = (count + kRowElems - 1) / kRowElems;
    if (CHECKED && (ip > in_len || in_len - ip < 4)) {
      err = kErrInput;
      return 0;
    }
    const uint32_t bytes = ahead;
    const bool zero_block = wave::read_lane(bytes, 0) == 0xFFu;
    const uint32_t wbytes = zero_block ? 4u : pad4(rows);
    if (CHECKED && in_len - ip < wbytes) {
      err = kErrInput;
      return 0;
    }
    const uint32_t widths = !zero_block && lane < rows ? bytes : 0u;
    if (CHECKED && wave::ballot(widths > W)) {
      err = kErrInput;
      return 0;
    }
    const uint32_t total_bits = E * wave::reduce_add(widths);
    const uint32_t dwords = (total_bits + 31) / 32;
    const uint8_t* payload = in + ip + wbytes;
    if (CHECKED && (in_len - ip - wbytes) / 256u < dwords) {
      err = kErrInput;
      return 0;
    }
    ip += wbytes + dwords * 256u;
    if (base + kBlock < nelem) {
      ahead = first_bytes(ip);
    }

    uint64_t acc = 0;
    uint32_t fill = 0;   /* uniform */
    uint32_t row = 0;    /* uniform */
    uint32_t part = 0;   /* uniform: a 64-bit element is taken in two parts, a small-type row in E */
    uint32_t held = 0;   /* the parts taken so far */
    uint32_t w = wave::read_lane(widths, 0);

    /* take every value that is complete in the accumulator */
    auto drain = [&]() {
      for (;;) {
        if (row >= rows) {
          return;
        }
        uint32_t need = w;
        if (S == 8) {
          need = part == 0 ? (w < 32 ? w : 32) : (w > 32 ? w - 32 : 0);
        }
        if (fill < need) {
          return;
        }
        const uint32_t bits = (uint32_t)(acc & ((1ull << need) - 1ull));
        acc >>= need;
        fill -= need;
        if (S == 8) {
          if (part == 0) {
            held = bits;
            part = 1;
            continue;
          }
          part = 0;
          const T v = (T)(((uint64_t)bits << 32) | held);
          T e = v;
          if (DELTA) {
            const T incl = (T)(scan_add<T>(unzigzag<T>(v)) + carry);
            carry = last_lane<T>(incl);
            e = incl;
          }
          const uint32_t i = base + 64 * row + lane;
          if (i < nelem) {
            store_elem<T>(out + (size_t)i * S, e);
          }
        } else if (E == 1) {
          const T v = (T)bits;
          T e = v;
          if (DELTA) {
            const T incl = (T)(scan_add<T>(unzigzag<T>(v)) + carry);
            carry = last_lane<T>(incl);
            e = incl;
          }
          const uint32_t i = base + 64 * row + lane;
          if (i < nelem) {
            store_elem<T>(out + (size_t)i * S, e);
          }
        } else {
          held |= bits << (8 * S * part);
          if (part + 1 < E) {
            ++part;
            continue;
          }
          /* the lane's E values of this row are complete */
          uint32_t vals = held;
          held = 0;
          part = 0;
          if (DELTA) {
            uint32_t sum[E];
            uint32_t run = 0;
#pragma unroll
            for (uint32_t k = 0; k < E; ++k) {
              run += (uint32_t)unzigzag<T>((T)(vals >> (8 * S * k)));
              sum[k] = run;
            }
            const uint32_t incl = wave::scan_add_inclusive(run & ((1u << W) - 1u));
            const uint32_t before = incl - (run & ((1u << W) - 1u)) + (uint32_t)carry;
            vals = 0;
#pragma unroll
            for (uint32_t k = 0; k < E; ++k) {
              vals |= (uint32_t)(T)(before + sum[k]) << (8 * S * k);
            }
            carry = (T)(wave::read_lane(incl, 63) + (uint32_t)carry);
          }
          const uint32_t i = base + kRowElems * row + E * lane;
          if (i + E <= nelem) {
            store_elem<uint32_t>(out + (size_t)i * S, vals);
          } else {
#pragma unroll
            for (uint32_t k = 0; k < E; ++k) {
              if (i + k < nelem) {
                store_elem<T>(out + (size_t)(i + k) * S, (T)(vals >> (8 * S * k)));
              }
            }
          }
        }
        ++row;
        w = wave::read_lane(widths, row & 63u);
      }
    };

    uint32_t d[8], e[8];
    auto fetch = [&](uint32_t kb, uint32_t (&to)[8]) {
#pragma unroll
      for (uint32_t j = 0; j < 8; ++j) {
        to[j] = 0;
        if (kb + j < dwords) {
          to[j] = load_u32(payload + (64u * (kb + j) + lane) * 4u);
        }
      }
    };
    fetch(0, e);
    for (uint32_t kb = 0; kb < dwords; kb += 8) {
      /* ONE wait a group: loads and stores share the counter (vmcnt) and the compiler, unable to tell their completions apart,
       * waits for EVERYTHING in flight in front of the first use of every loaded register -- a dword at a time that was a wait
       * for the stores of the rows just written, ten times a block. All eight are "used" here, before the next group's loads and
       * this group's stores are issued. */
#pragma unroll
      for (uint32_t j = 0; j < 8; ++j) {
        wave::touch(e[j]);
        d[j] = e[j];
      }
      fetch(kb + 8, e);
#pragma unroll
      for (uint32_t j = 0; j < 8; ++j) {
        if (kb + j < dwords) {
          drain();
          acc |= (uint64_t)d[j] << fill;
          fill += 32;
        }
      }
    }
    drain();
    if (CHECKED && row < rows) {
      err = kErrInput;
      return 0;
    }
  }
  const uint32_t tail = n - nelem * S;
  if (CHECKED && (ip > in_len || in_len - ip < tail)) {
    err = kErrInput;
    return 0;
  }
  if (lane < tail) {
    out[nelem * S + lane] = in[ip + lane];
  }
  return n;
}

struct Header
{
  bool ok;
  uint32_t algo;
  uint32_t log2_size;
  uint32_t n;
};

__device__ __forceinline__ Header read_header(const uint8_t* in, uint32_t in_len)
{
  Header h;
  h.ok = false;
  h.algo = 0;
  h.log2_size = 0;
  h.n = 0;
  if (in_len < kHeaderBytes) {
    return h;
  }
  const uint32_t magic = wave::uniform(load_u32(in));
  const uint32_t kind = wave::uniform(load_u32(in + 4));
  h.n = wave::uniform(load_u32(in + 8));
  h.algo = kind & 0xffu;
  h.log2_size = (kind >> 8) & 0xffu;
  h.ok = magic == 0x01435442u && h.algo <= 1 && h.log2_size <= 3 && (kind >> 16) == 0;
  return h;
}

template <bool CHECKED>
__device__ __forceinline__ uint32_t decode_chunk(
    const uint8_t* __restrict__ in, uint32_t in_len, uint8_t* __restrict__ out, uint32_t out_cap, uint32_t& err)
{
  err = kErrNone;
  const Header h = read_header(in, in_len);
  if (!h.ok) {
    err = kErrInput;
    return 0;
  }
  if (h.n > out_cap) {
    err = kErrOutput;
    return 0;
  }
  const uint32_t kind = h.log2_size * 2 + h.algo;
  switch (kind) {
  case 0: return decode_body<uint8_t, true, CHECKED>(in, in_len, out, h.n, err);
  case 1: return decode_body<uint8_t, false, CHECKED>(in, in_len, out, h.n, err);
  case 2: return decode_body<uint16_t, true, CHECKED>(in, in_len, out, h.n, err);
  case 3: return decode_body<uint16_t, false, CHECKED>(in, in_len, out, h.n, err);
  case 4: return decode_body<uint32_t, true, CHECKED>(in, in_len, out, h.n, err);
  case 5: return decode_body<uint32_t, false, CHECKED>(in, in_len, out, h.n, err);
  case 6: return decode_body<uint64_t, true, CHECKED>(in, in_len, out, h.n, err);
  default: return decode_body<uint64_t, false, CHECKED>(in, in_len, out, h.n, err);
  }
}

} // namespace bitcomp
