/*
 * deflate/deflate_encode.hip.h -- batched DEFLATE compressor for gfx950.
 *
 * Replaces the device side of nvcompBatchedDeflateCompressAsync (reference call site:
 * examples/deflate_cpu_decompression.cu:93-103; the output must be accepted by libdeflate / zlib inflate,
 * :128-170). One wavefront per chunk, one block per chunk:
 *   - the LZ77 sequences come from the wave-parallel greedy match finder the LZ4 and Snappy compressors use
 *     (common/lz_match.hip.h: 64 positions per step, LDS hash table, LDS image of the input), with the match distance
 *     capped at DEFLATE's 32 768;
 *   - they are written with the FIXED Huffman code of RFC 1951 3.2.6 (no code construction: the fastest setting of
 *     the CPU libraries' "level 1" class). A sequence is a bit string here, not a byte string: every selected lane
 *     sizes its own (8 or 9 bits per literal, length and distance codes with their extra bits), a DPP prefix sum
 *     over the bit counts gives the lanes their bit offsets, and the lanes OR their code words into an LDS staging
 *     area (ds_or_b32) that is flushed to the chunk's output in whole dwords after every step;
 *   - a chunk the fixed code would expand (9 bits per literal, no matches: random bytes) is written as STORED blocks
 *     instead.
 * This is nvcompBatchedDeflateOpts_t.algo 0; algo 1 and 2 write per-chunk Huffman codes
 * (deflate_encode_dynamic.hip.h). The reference's values select CPU-library-like effort levels
 * (benchmarks/benchmark_deflate_chunked.cu:43); every value must produce standard streams.
 */
#pragma once

#include "common/lz_match.hip.h"

namespace deflate {

constexpr uint32_t kStoredMax = 65535;
constexpr uint32_t kStageBytes = 2048;   /* LDS staging of one wave: a step's bit strings (64 x 175 bits) + the carry */
constexpr uint32_t kLaneLits = 16;       /* literal run a lane writes by itself */
constexpr uint32_t kMaxMatch = 258, kMinMatch = 3;

__host__ __device__ inline size_t stored_size(size_t n)
{
  return n + 5 * (n / kStoredMax + 1);
}

/* Worst case of an ATTEMPT, which is written into the slot before the stored form is chosen instead: 9 bits per byte
 * (the fixed code; a per-chunk Huffman code built over counts that start at one costs at most one bit per symbol more
 * than the 8-bit code) + block header + end of block + the per-chunk code's own header (316 code lengths of 7 bits and
 * the 19 of the code-length code: under 300 bytes), rounded up. */
__host__ __device__ inline size_t max_compressed_size(size_t n)
{
  const size_t fixed = n + n / 8 + 16 + 320;
  return fixed > stored_size(n) ? fixed : stored_size(n);
}

/* ---- stored blocks (RFC 1951 3.2.4) ---- */
__device__ __forceinline__ uint32_t encode_stored(const uint8_t* __restrict__ src, uint32_t n, uint8_t* dst)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  uint32_t ip = 0, op = 0;
  do {
    const uint32_t len = n - ip < kStoredMax ? n - ip : kStoredMax;
    const bool final_block = ip + len == n;
    if (lane == 0) {
      dst[op] = final_block ? 1 : 0; /* BFINAL, BTYPE = 00, padding to the byte boundary */
      dst[op + 1] = (uint8_t)len;
      dst[op + 2] = (uint8_t)(len >> 8);
      dst[op + 3] = (uint8_t)~len;
      dst[op + 4] = (uint8_t)(~len >> 8);
    }
    lz::wave_copy(dst + op + 5, src + ip, len);
    ip += len;
    op += 5 + len;
  } while (ip < n);
  return op;
}

/* ---- the bit sink ---- */
struct BitSink
{
  uint8_t* dst;
  uint32_t* stage; /* LDS, kStageBytes: bit `base` of the stream is bit 0 of stage[0]; zero beyond the bits written */
  uint32_t bits;   /* bits of the stream written so far (wave-uniform) */
  uint32_t base;   /* multiple of 32 */
};

__device__ __forceinline__ void sink_init(BitSink& s, uint8_t* dst, uint8_t* lds)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  s.dst = dst;
  s.stage = (uint32_t*)lds;
  s.bits = 0;
  s.base = 0;
  for (uint32_t i = lane; i < kStageBytes / 4; i += 64) {
    s.stage[i] = 0;
  }
  wave::sync();
}

/* n <= 32 bits of `code` (LSB first) at bit position p of the stream, per lane */
__device__ __forceinline__ void put(const BitSink& s, uint32_t p, uint32_t code, uint32_t n)
{
  (void)n;
  const uint32_t at = (p - s.base) >> 5;
  const uint64_t v = (uint64_t)code << (p & 31u);
  wave::lds_or(s.stage + at, (uint32_t)v);
  if ((uint32_t)(v >> 32) != 0) {
    wave::lds_or(s.stage + at + 1, (uint32_t)(v >> 32));
  }
}

/* Whole dwords of the staging area go out to the chunk; the incomplete one moves to the front. */
__device__ __forceinline__ void flush(BitSink& s)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  wave::sync();
  const uint32_t k = (s.bits - s.base) >> 5;
  if (k == 0) {
    return;
  }
  uint8_t* out = s.dst + (s.base >> 3);
  for (uint32_t i = lane; i < k; i += 64) {
    lz::st_u32(out + 4 * i, s.stage[i]);
  }
  const uint32_t rest = s.stage[k];
  wave::sync();
  for (uint32_t i = lane; i <= k; i += 64) {
    s.stage[i] = i == 0 ? rest : 0u;
  }
  wave::sync();
  s.base += 32 * k;
}

/* The stream's last bytes (after the end-of-block code). Returns its size in bytes. */
__device__ __forceinline__ uint32_t finish(BitSink& s)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  flush(s);
  const uint32_t bytes = (s.bits + 7) >> 3;
  const uint32_t have = s.base >> 3;
  if (lane < bytes - have) { /* < 4 */
    s.dst[have + lane] = (uint8_t)(s.stage[0] >> (8 * lane));
  }
  return bytes;
}

/* ---- the fixed code ---- */
__device__ __forceinline__ uint32_t rev(uint32_t code, uint32_t n) /* Huffman codes go in most significant bit first */
{
  return wave::bit_reverse(code) >> (32 - n);
}

/* literal byte b: code word (bit-reversed) and its length (8 or 9) */
__device__ __forceinline__ uint32_t literal_code(uint32_t b, uint32_t& n)
{
  n = b < 144 ? 8u : 9u;
  return rev(b < 144 ? 0x30u + b : 0x190u + (b - 144), n);
}

/* length 3..258 -> symbol 257..285, its extra-bit count and value */
__device__ __forceinline__ uint32_t length_symbol(uint32_t mlen, uint32_t& k, uint32_t& extra)
{
  const uint32_t m = mlen - 3;
  k = 0, extra = 0;
  if (mlen == 258) {
    return 285;
  }
  if (m < 8) {
    return 257 + m;
  }
  k = 29 - (uint32_t)__builtin_clz(m); /* floor(log2 m) - 2 */
  extra = m & ((1u << k) - 1u);
  return 261 + 4 * k + ((m >> k) & 3u);
}

/* distance 1..32768 -> symbol 0..29, its extra-bit count and value */
__device__ __forceinline__ uint32_t distance_symbol(uint32_t dist, uint32_t& k, uint32_t& extra)
{
  const uint32_t d = dist - 1;
  k = 0, extra = 0;
  if (d < 4) {
    return d;
  }
  k = 30 - (uint32_t)__builtin_clz(d); /* floor(log2 d) - 1 */
  extra = d & ((1u << k) - 1u);
  return 2 * k + 2 + ((d >> k) & 1u);
}

/* <length 3..258, distance 1..32768> in the fixed code: the whole pair as one bit string of at most 31 bits */
__device__ __forceinline__ uint32_t match_code(uint32_t mlen, uint32_t dist, uint32_t& n)
{
  uint32_t k, extra;
  const uint32_t sym = length_symbol(mlen, k, extra);
  uint32_t bits, used;
  if (sym < 280) { /* 256..279: 7 bits, 0000000.. */
    bits = rev(sym - 256, 7);
    used = 7;
  } else { /* 280..287: 8 bits, 11000000.. */
    bits = rev(0xc0u + (sym - 280), 8);
    used = 8;
  }
  bits |= extra << used;
  used += k;
  uint32_t k2, dextra;
  const uint32_t dsym = distance_symbol(dist, k2, dextra);
  bits |= rev(dsym, 5) << used; /* distances: 5 bits each */
  used += 5;
  bits |= dextra << used;
  n = used + k2;
  return bits;
}

/* How a match longer than 258 is cut: every piece at least 3 long. */
__device__ __forceinline__ uint32_t next_piece(uint32_t left)
{
  return left <= kMaxMatch ? left : left - kMaxMatch < kMinMatch ? kMaxMatch - kMinMatch : kMaxMatch;
}

struct Emitter
{
  static constexpr bool kStream = true;
  static constexpr uint32_t kReach = 32768;

  /* Whole wave, one sequence of any size: literals 64 at a time, then the match in pieces. */
  static __device__ __forceinline__ void one(BitSink& s, const uint8_t* lit, uint32_t lit_len, uint32_t offset, uint32_t match_len)
  {
    const uint32_t lane = (uint32_t)wave::lane_id();
    for (uint32_t base = 0; base < lit_len; base += 64) {
      uint32_t n = 0, code = 0;
      if (base + lane < lit_len) {
        code = literal_code(lit[base + lane], n);
      }
      const uint32_t incl = wave::scan_add_inclusive(n);
      if (n != 0) {
        put(s, s.bits + incl - n, code, n);
      }
      s.bits += wave::read_lane(incl, 63);
      flush(s);
    }
    while (match_len != 0) {
      const uint32_t piece = next_piece(match_len);
      uint32_t n;
      const uint32_t code = match_code(piece, offset, n);
      if (lane == 0) {
        put(s, s.bits, code, n);
      }
      s.bits += n;
      match_len -= piece;
      if (s.bits - s.base > 8 * (kStageBytes - 64)) {
        flush(s);
      }
    }
    flush(s);
  }

  /*
   * One step's selected sequences, lane-parallel. lit_from / lit_len: the lane's literal run in src; before8: the 8
   * bytes right before the lane's position when before_ok (the run is then the last `run` of them, of which the
   * first lit_len are literals -- the rest joined the match when it grew backwards).
   */
  static __device__ __forceinline__ void window(
      BitSink& s, const uint8_t* __restrict__ src, bool sel, uint32_t lit_from, uint32_t lit_len, uint32_t match_len,
      uint32_t offset, uint64_t before8, bool before_ok, uint32_t run)
  {
    const uint32_t lane = (uint32_t)wave::lane_id();
    /* a lane writes its sequence alone when the literals are few and in reach without a bounds question */
    const bool alone = sel && match_len <= kMaxMatch && lit_len <= kLaneLits && (before_ok || lit_len == 0 || lit_len >= 4);
    const uint64_t hard = wave::ballot(sel && !alone);
    const uint32_t first_hard = hard ? wave::ctz64(hard) : 64u;
    const bool mine = alone && lane < first_hard;

    /* the literal bytes, up to 16, as two 64-bit words */
    uint64_t lo = 0, hi = 0;
    if (mine && lit_len != 0) {
      if (before_ok) {
        lo = before8 >> (8 * (8 - run));
      } else {
        /* dword loads up to 3 bytes past the run stay inside the chunk: the match behind it is at least 4 long */
        const uint8_t* p = src + lit_from;
        lo = wave::gload_u32(p);
        if (lit_len > 4) {
          lo |= (uint64_t)wave::gload_u32(p + 4) << 32;
        }
        if (lit_len > 8) {
          hi = wave::gload_u32(p + 8);
        }
        if (lit_len > 12) {
          hi |= (uint64_t)wave::gload_u32(p + 12) << 32;
        }
      }
    }
    /* bits: 8 per literal + 1 for each byte >= 144, and the pair */
    uint32_t nine = 0;
    {
      const uint64_t keep_lo = lit_len >= 8 ? ~0ull : (1ull << (8 * lit_len)) - 1ull;
      const uint64_t keep_hi = lit_len >= 16 ? ~0ull : lit_len > 8 ? (1ull << (8 * (lit_len - 8))) - 1ull : 0ull;
      const uint64_t m7 = 0x7f7f7f7f7f7f7f7full, add = 0x7070707070707070ull, top = 0x8080808080808080ull;
      const uint64_t a = lo & keep_lo, b = hi & keep_hi;
      nine = (uint32_t)__builtin_popcountll(((a & m7) + add) & a & top) + (uint32_t)__builtin_popcountll(((b & m7) + add) & b & top);
    }
    uint32_t pair_bits = 0, pair_code = 0;
    if (mine) {
      pair_code = match_code(match_len, offset, pair_bits);
    }
    const uint32_t size = mine ? 8 * lit_len + nine + pair_bits : 0u;
    const uint32_t incl = wave::scan_add_inclusive(size);
    uint32_t p = s.bits + incl - size;
    /* literals, three code words (at most 27 bits) per turn and per LDS update */
    for (uint32_t i = 0; wave::ballot(mine && i < lit_len) != 0; i += 3) {
      if (mine && i < lit_len) {
        uint32_t word = 0, filled = 0;
#pragma unroll
        for (uint32_t j = 0; j < 3; ++j) {
          if (i + j < lit_len) {
            uint32_t n;
            const uint32_t code = literal_code((uint32_t)lo & 0xffu, n);
            word |= code << filled;
            filled += n;
            lo = (lo >> 8) | (hi << 56);
            hi >>= 8;
          }
        }
        put(s, p, word, filled);
        p += filled;
      }
    }
    if (mine) {
      put(s, p, pair_code, pair_bits);
    }
    s.bits += wave::read_lane(incl, 63);
    flush(s);
    /* from the first sequence a lane cannot write alone on: one after the other */
    uint64_t rest = first_hard < 64 ? wave::ballot(sel) & (~0ull << first_hard) : 0ull;
    while (rest) {
      const uint32_t j = wave::ctz64(rest);
      rest &= rest - 1;
      one(s, src + wave::read_lane(lit_from, j), wave::read_lane(lit_len, j), wave::read_lane(offset, j),
          wave::read_lane(match_len, j));
    }
  }
};

/* LDS of one wave: the match finder's table and input image + the staging area */
constexpr uint32_t kEncLdsPerWave = 2 * lzm::kTableU16 + lzm::kStageBytes + kStageBytes;

/* Compress src[0, n) into dst (capacity >= max_compressed_size(n)) with the calling wave. Returns the size. */
__device__ __forceinline__ uint32_t encode_chunk(const uint8_t* __restrict__ src, uint32_t n, uint8_t* dst, uint8_t* lds)
{
  uint16_t* table = (uint16_t*)lds;
  uint8_t* image = lds + 2 * lzm::kTableU16;
  uint8_t* stage = image + lzm::kStageBytes;
  if (n < 16) {
    return encode_stored(src, n, dst);
  }
  BitSink s;
  sink_init(s, dst, stage);
  if (wave::lane_id() == 0) {
    put(s, 0, 1u | (1u << 1), 3); /* BFINAL = 1, BTYPE = 01 (fixed Huffman codes) */
  }
  s.bits = 3;
  (void)lzm::encode_chunk<Emitter, 1, BitSink>(src, n, dst, table, image, n - 4, n, true, &s);
  if (wave::lane_id() == 0) {
    put(s, s.bits, 0, 7); /* end of block: 0000000 */
  }
  s.bits += 7;
  const uint32_t size = finish(s);
  if (size > stored_size(n)) {
    wave::sync();
    return encode_stored(src, n, dst);
  }
  return size;
}

} // namespace deflate
