/*
 * deflate/deflate_decode.hip.h -- batched DEFLATE (RFC 1951) / gzip (RFC 1952) decoder for gfx950.
 *
 * Replaces the device side of nvcompBatchedDeflateDecompressAsync / nvcompBatchedGzipDecompressAsync (reference call
 * sites: examples/deflate_cpu_compression.cu:133-187 -- raw deflate streams written by libdeflate, zlib compress2 with
 * the wrapper cut off, or deflateInit2(-15) -- and examples/gzip_gpu_decompression.cu:110-164 -- deflateInit2(15 | 16)).
 * The wire formats are the public ones; the CPU peers are the oracle (zlib in tests/).
 *
 * One wavefront per chunk, in two halves:
 *   front end  Huffman symbols -> sequence records {literal run, match length, distance} + literal bytes in a 2 KiB
 *              literal ring in LDS. A symbol's length is only known once it is decoded, so the wave decodes
 *              SPECULATIVELY at every bit position of a 256-bit window (four per lane: one lookup for the
 *              literal/length code, one for the distance code), turns the lengths into jump tables by pointer
 *              doubling (the token chase of common/lz_window.hip.h, on bit positions) and reads the true chain of
 *              symbol starts off them, 32 at a time; lane k then decodes symbol k for good and the records are
 *              assembled with ballots. What the lookups cannot tell (a code longer than the lookup, end of block,
 *              an illegal symbol) is left to a wave-uniform decoder that takes one symbol at a time (the scalar
 *              unit: a 64-bit bit buffer in SGPRs); block headers and table construction are wave-uniform /
 *              wave-cooperative too. The symbol-at-a-time decoder alone runs at 23 GB/s: the CU's single scalar
 *              unit is the bound (profiles/archive/r02_deflate.json);
 *   back end   64 records at a time (at most lzw::kBatchMax output bytes) are executed by the LZ window executor the
 *              LZ4 and Snappy decoders use (common/lz_window.hip.h): lane k copies sequence k inside the LDS output
 *              window, far matches (up to 32 KiB back) come from HBM, the window is flushed in aligned 16-byte stores.
 * The literal ring is handed to the executor as its "input ring": a literal run is what LZ4 calls the literals of a
 * sequence, only that here the bytes were decoded rather than copied from the stream.
 *
 * Decoding tables (per wave, LDS): a 10-bit lookup for the literal/length code and an 8-bit one for the distance
 * code, entry = symbol << 4 | code length; longer codes (rare: each has probability < 2^-10 / 2^-8) are decoded
 * canonically, bit by bit, from the per-length counts and the symbols sorted by code. Tables are built by the whole
 * wave: counts and sorted symbols with ballots, lookup entries 16 per lane.
 */
#pragma once

/* DEFLATE's own window and ring sizes. Its LDS (lookup tables, the literal ring, the bit-position jump tables) is the
 * occupancy limiter, and its batches are cut by the front end at 40-64 records -- about 400 bytes of a zlib stream's
 * output -- long before they are full: batches of 768 bytes (+ 32 of history) and two rings of 1 KiB (the compressed
 * stream, refilled 512 bytes at a time, and the decoded literals, of which a batch holds at most its own 768 bytes)
 * keep a wave under 8 KiB = 20 waves per CU (rounds 2-3: 928 / 2 KiB / 2 KiB = 10 KiB, 16 waves). */
#ifndef NVCOMP_LZW_BATCHMAX
#define NVCOMP_LZW_BATCHMAX 768
#endif
#ifndef NVCOMP_LZW_INRING
#define NVCOMP_LZW_INRING 1024
#endif
#include "common/lz_window.hip.h"

namespace deflate {

constexpr uint32_t kMaxBits = 15;
constexpr uint32_t kLitLenSyms = 288;
constexpr uint32_t kDistSyms = 32;
#ifndef NVCOMP_DEFLATE_LUT_BITS
#define NVCOMP_DEFLATE_LUT_BITS 10 /* A/B: 9 frees 1 KiB of LDS per wave (16 instead of 14 waves per CU), more symbols go the slow way */
#endif
constexpr uint32_t kLutBits = NVCOMP_DEFLATE_LUT_BITS;
constexpr uint32_t kDistLutBits = 8;
constexpr uint32_t kClLutBits = 7; /* the code-length code: at most 7 bits, always decoded by lookup */
constexpr uint32_t kRunMax = 255;  /* literal bytes per sequence record (8 bits of the record) */
constexpr uint32_t kRunClose = 192; /* a pending run this long is closed as a record of its own between rounds */
/* 512 positions per window (8 per lane) halve the table builds and cut the enumerations by a third (counted on the host
 * emulation: 2.3 -> 1.16 builds and 3.7 -> 2.6 enumerations per round of 49 symbols). On the card that is worth 2-3 %
 * (profiles/archive/r03_deflate_scan.jsonl: at 10 KiB + 1.25 KiB of LDS per wave it was neutral against 256 at 10 KiB; with the
 * rings at 1 KiB, 512 at 9.1 KiB = 17 waves per CU reads 80.1 / 97.3 GB/s at 1 / 4 GiB against 77.7 / 95.8 for 256 at
 * 7.8 KiB = 20 waves): the decoder is bound by the NUMBER of instructions per symbol -- 17 vector + 14 scalar, of which
 * the speculative decode of every bit position is the largest part and does not depend on the window -- more than by the
 * length of its dependent chains. */
#ifndef NVCOMP_DEFLATE_SCAN_WIN
#define NVCOMP_DEFLATE_SCAN_WIN 512
#endif
constexpr uint32_t kScanWin = NVCOMP_DEFLATE_SCAN_WIN; /* bit positions one speculative window covers: 256 or 512 */
constexpr uint32_t kScanPer = kScanWin / 64;            /* ... of which a lane decodes this many (consecutive ones) */
constexpr uint32_t kScanLevels = 5;  /* jump tables for 1, 2, 4, 8, 16 symbols ahead; an enumeration follows them for 64 */
static_assert(kScanPer == 4 || kScanPer == 8, "a lane's table entries are one or two dwords");
constexpr uint32_t kTopBias = 64;    /* the 16-symbol table holds distances of 64 .. 318 bits, minus this */

/* ---- LDS of one wave, behind the executor's window and the stream ring ---- */
constexpr uint32_t kOffLit = 0;                                  /* literal ring: lzw::kInLds bytes */
constexpr uint32_t kOffLutLL = kOffLit + lzw::kInLds;            /* uint16[1 << kLutBits] */
constexpr uint32_t kOffLutD = kOffLutLL + (2u << kLutBits);      /* uint16[1 << kDistLutBits] (also the code-length code's) */
constexpr uint32_t kOffCntLL = kOffLutD + (2u << kDistLutBits);  /* uint16[16] */
constexpr uint32_t kOffCntD = kOffCntLL + 32;                    /* uint16[16] */
constexpr uint32_t kOffSymLL = kOffCntD + 32;                    /* uint16[288] */
constexpr uint32_t kOffSymD = kOffSymLL + 2 * kLitLenSyms;       /* uint16[32] */
constexpr uint32_t kOffRec = kOffSymD + 2 * kDistSyms;           /* 64 records x 8 bytes; the 320 code lengths of a */
constexpr uint32_t kRecBytes = 64 * 8;                           /* dynamic header while it is read */
constexpr uint32_t kOffScan = kOffRec + kRecBytes;               /* kScanLevels byte tables of kScanWin entries */
constexpr uint32_t kFrontLds = kOffScan + kScanLevels * kScanWin;
static_assert(kRecBytes >= kLitLenSyms + kDistSyms, "the code lengths share the record area");
constexpr uint32_t kLdsPerWave = lzw::kOutLds + lzw::kInLds + kFrontLds; /* window + stream ring + the above */
static_assert(kFrontLds % 16 == 0 && (lzw::kOutLds + lzw::kInLds) % 16 == 0, "16-byte alignment of the rings");

enum : uint32_t { kGzip = 1 };

/* ---- the bit reader: wave-uniform, over the stream ring ---- */
struct Bits
{
  uint64_t buf;  /* bits not yet consumed, LSB first */
  uint32_t cnt;  /* how many */
  uint32_t next; /* virtual position (multiple of 4) of the next dword to load */

  __device__ __forceinline__ uint32_t dword(const lzw::InRing& ir, uint32_t v) const
  {
    return wave::uniform(*(const uint32_t*)(ir.ring + (v & (lzw::kInRing - 1))));
  }
  /* continue at virtual byte position v */
  __device__ __forceinline__ void seek(const lzw::InRing& ir, uint32_t v)
  {
    next = v & ~3u;
    buf = (uint64_t)dword(ir, next) >> (8 * (v & 3u));
    cnt = 32 - 8 * (v & 3u);
    next += 4;
  }
  /* at least 32 bits in hand */
  __device__ __forceinline__ void refill(const lzw::InRing& ir)
  {
    if (cnt < 32) {
      buf |= (uint64_t)dword(ir, next) << cnt;
      cnt += 32;
      next += 4;
    }
  }
  __device__ __forceinline__ uint32_t peek(uint32_t n) const { return (uint32_t)buf & ((1u << n) - 1u); }
  __device__ __forceinline__ void drop(uint32_t n)
  {
    buf >>= n;
    cnt -= n;
  }
  __device__ __forceinline__ uint32_t take(uint32_t n)
  {
    const uint32_t v = peek(n);
    drop(n);
    return v;
  }
  /* virtual position of the first byte no bit of which was consumed */
  __device__ __forceinline__ uint32_t byte_pos() const { return next - (cnt >> 3); }
  /* the same in bits (8 x virtual byte position + bit) */
  __device__ __forceinline__ uint32_t bit_pos() const { return 8 * next - cnt; }
  __device__ __forceinline__ void seek_bit(const lzw::InRing& ir, uint32_t q)
  {
    next = (q >> 5) << 2;
    buf = (uint64_t)dword(ir, next) >> (q & 31u);
    cnt = 32 - (q & 31u);
    next += 4;
  }
};

/* ---- tables ---- */
struct Code
{
  uint16_t* lut;
  uint16_t* cnt;  /* codes per length, [0..15] */
  uint16_t* syms; /* symbols in code order */
};

/* What a lookup entry says besides symbol << 4 | code length, so that the speculative decoder needs no arithmetic on
 * the symbol: a length symbol (257..285) is stored as 0x8000 | extra bits << 9 | (symbol - 257) << 4 | code length,
 * a distance symbol as extra bits << 9 | symbol << 4 | code length. */
enum : uint32_t { kPlainCode, kLitLenCode, kDistCode };

__device__ __forceinline__ uint32_t lut_entry(uint32_t kind, uint32_t sym, uint32_t len)
{
  if (kind == kLitLenCode && sym >= 257 && sym <= 285) {
    const uint32_t ls = sym - 257;
    const uint32_t k = ls < 8 || ls == 28 ? 0u : (ls - 4) >> 2;
    return 0x8000u | (k << 9) | (ls << 4) | len;
  }
  if (kind == kDistCode) {
    const uint32_t k = sym < 4 ? 0u : (sym >> 1) - 1; /* 30, 31: illegal symbols, refused where they are met */
    return ((k & 15u) << 9) | (sym << 4) | len;
  }
  return (sym << 4) | len;
}

/* the symbol an entry stands for */
__device__ __forceinline__ uint32_t entry_symbol(uint32_t kind, uint32_t e)
{
  if (kind == kLitLenCode) {
    return e & 0x8000u ? 257 + ((e >> 4) & 31u) : (e >> 4) & 0x1ffu;
  }
  return kind == kDistCode ? (e >> 4) & 31u : e >> 4;
}

/*
 * Build the decoding tables of one canonical Huffman code from its code lengths (lens[0, n), n <= 320 -- 0 = unused
 * symbol). Returns false for an over-subscribed set of lengths. An incomplete set is accepted: a bit pattern
 * without a symbol fails when (if) it is met.
 */
template <uint32_t LUT_BITS, uint32_t KIND>
__device__ __forceinline__ bool build_code(const Code& code, const uint8_t* lens, uint32_t n)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  uint32_t count[kMaxBits + 1];
#pragma unroll
  for (uint32_t l = 0; l <= kMaxBits; ++l) {
    count[l] = 0;
  }
  for (uint32_t base = 0; base < n; base += 64) {
    const uint32_t len = base + lane < n ? lens[base + lane] : 0u;
#pragma unroll
    for (uint32_t l = 1; l <= kMaxBits; ++l) {
      count[l] += wave::popc64(wave::ballot(len == l));
    }
  }
  int32_t left = 1;
  uint32_t offs[kMaxBits + 2];
  offs[1] = 0;
#pragma unroll
  for (uint32_t l = 1; l <= kMaxBits; ++l) {
    left = 2 * left - (int32_t)count[l];
    offs[l + 1] = offs[l] + count[l];
  }
  if (lane <= kMaxBits) {
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t l = 1; l <= kMaxBits; ++l) {
      mine = lane == l ? count[l] : mine;
    }
    code.cnt[lane] = (uint16_t)mine;
  }
  if (left < 0) {
    return false;
  }
  /* symbols in code order: by length, then by symbol */
  for (uint32_t base = 0; base < n; base += 64) {
    const uint32_t len = base + lane < n ? lens[base + lane] : 0u;
#pragma unroll
    for (uint32_t l = 1; l <= kMaxBits; ++l) {
      const uint64_t same = wave::ballot(len == l);
      if (len == l) {
        code.syms[offs[l] + wave::prefix_popc(same)] = (uint16_t)(base + lane);
      }
      offs[l] += wave::popc64(same);
    }
  }
  wave::sync();
  /* lookup entries: the bits of the index, first bit read = bit 0, walked as a canonical code */
  for (uint32_t e = lane; e < (1u << LUT_BITS); e += 64) {
    uint32_t entry = 0, c = 0, first = 0, index = 0;
#pragma unroll
    for (uint32_t l = 1; l <= LUT_BITS; ++l) {
      c |= (e >> (l - 1)) & 1u;
      if (entry == 0 && c - first < count[l]) {
        entry = lut_entry(KIND, code.syms[index + (c - first)], l);
      }
      index += count[l];
      first = (first + count[l]) << 1;
      c <<= 1;
    }
    code.lut[e] = (uint16_t)entry;
  }
  wave::sync();
  return true;
}

/* A symbol whose code is longer than the lookup (or does not exist): canonical walk, wave-uniform. Returns the
 * symbol, or ~0u when no code matches the next 15 bits; `len` = bits it took. */
__device__ __forceinline__ uint32_t slow_symbol(const Code& code, uint32_t bits, uint32_t& len)
{
  uint32_t c = 0, first = 0, index = 0;
  for (uint32_t l = 1; l <= kMaxBits; ++l) {
    c |= (bits >> (l - 1)) & 1u;
    const uint32_t count = wave::uniform(code.cnt[l]);
    if (c - first < count) {
      len = l;
      return wave::uniform(code.syms[index + (c - first)]);
    }
    index += count;
    first = (first + count) << 1;
    c <<= 1;
  }
  len = kMaxBits;
  return ~0u;
}

template <uint32_t LUT_BITS, uint32_t KIND>
__device__ __forceinline__ uint32_t next_symbol(const Code& code, Bits& b, bool& bad)
{
  const uint32_t e = wave::uniform(code.lut[b.peek(LUT_BITS)]);
  uint32_t len = e & 15u;
  uint32_t sym = entry_symbol(KIND, e);
  if (len == 0) {
    sym = slow_symbol(code, (uint32_t)b.buf, len);
    bad = bad || sym == ~0u;
  }
  b.drop(len);
  return sym;
}

/* ---- what the front end keeps between batches ---- */
struct Front
{
  lzw::InRing lit;   /* the literal ring, in the executor's clothes: lo/hi = the literal positions it may read */
  uint32_t lw;       /* literal bytes written so far (= virtual literal position of the next one) */
  uint32_t run;      /* literal bytes of the sequence being assembled (the last `run` bytes before lw) */
  uint32_t* rec;     /* 64 records: {lit_src, packed} */
  uint32_t n;        /* records in hand */
  uint32_t bytes;    /* output bytes they produce */
  uint32_t carry0, carry1; /* a record that did not fit the batch */
  bool carried;
  uint64_t produced; /* SIZE_ONLY: bytes so far */
};

/* literal run (<= 255) | match length code (0 = none, else length - 2) << 8 | (distance - 1) << 17 */
__device__ __forceinline__ uint32_t pack_record(uint32_t lit_len, uint32_t match_len, uint32_t match_off)
{
  return match_len ? lit_len | ((match_len - 2) << 8) | ((match_off - 1) << 17) : lit_len;
}

/* Execute the records in hand. Returns false on error. */
template <bool CHECKED>
__device__ __forceinline__ bool run_batch(Front& f, lzw::OutWindow& ow, uint32_t limit, uint32_t& op, uint32_t& err)
{
  if (f.n == 0) {
    return true;
  }
  const uint32_t lane = (uint32_t)wave::lane_id();
  wave::sync();
  lz::Seq s;
  s.lit_src = 0, s.lit_len = 0, s.match_off = 0, s.match_len = 0;
  if (lane < f.n) {
    const uint32_t a = f.rec[2 * lane], b = f.rec[2 * lane + 1];
    const uint32_t code = (b >> 8) & 0x1ffu;
    s.lit_src = a;
    s.lit_len = b & 0xffu;
    s.match_len = code ? code + 2 : 0u;
    s.match_off = code ? (b >> 17) + 1 : 0u;
  }
  f.lit.lo = wave::read_lane(s.lit_src, 0);
  f.lit.hi = f.lw;
  bool big;
  const uint32_t took = lzw::execute_window_batch<CHECKED, true>(f.lit, ow, limit, op, f.n, s, err, big);
  if (err) {
    return false;
  }
  if (took != f.n || big) { /* cannot happen: a batch is cut at lzw::kBatchMax bytes */
    err |= lz::kErrInput;
    return false;
  }
  wave::sync();
  f.n = 0;
  f.bytes = 0;
  return true;
}

/* Run the batch in hand; a record that did not fit becomes the first of the next batch. */
template <bool CHECKED>
__device__ __forceinline__ bool drain(Front& f, lzw::OutWindow& ow, uint32_t limit, uint32_t& op, uint32_t& err)
{
  if (!run_batch<CHECKED>(f, ow, limit, op, err)) {
    return false;
  }
  if (f.carried) {
    f.rec[0] = f.carry0; /* every lane writes the same words */
    f.rec[1] = f.carry1;
    f.n = 1;
    const uint32_t code = (f.carry1 >> 8) & 0x1ffu;
    f.bytes = (f.carry1 & 0xffu) + (code ? code + 2 : 0u);
    f.carried = false;
  }
  return true;
}

/* Close a sequence, symbol-at-a-time path: literal run f.run (ending at f.lw) + a match. */
template <bool SIZE_ONLY>
__device__ __forceinline__ void close_sequence(Front& f, uint32_t match_len, uint32_t match_off)
{
  const uint32_t size = f.run + match_len;
  if (SIZE_ONLY) {
    f.produced += size;
    f.run = 0;
    return;
  }
  const uint32_t r0 = f.lw - f.run;
  const uint32_t r1 = pack_record(f.run, match_len, match_off);
  f.run = 0;
  if (f.n == 64 || f.bytes + size > lzw::kBatchMax) {
    f.carry0 = r0; /* the caller runs the batch and puts this record first in the next one */
    f.carry1 = r1;
    f.carried = true;
    return;
  }
  f.rec[2 * f.n] = r0; /* every lane writes the same words */
  f.rec[2 * f.n + 1] = r1;
  f.n += 1;
  f.bytes += size;
}

template <bool SIZE_ONLY>
__device__ __forceinline__ void put_literal(Front& f, uint32_t byte)
{
  if (!SIZE_ONLY) {
    const uint32_t at = f.lw & (lzw::kInRing - 1);
    f.lit.ring[at] = (uint8_t)byte;
    if (at < 16) {
      f.lit.ring[lzw::kInRing + at] = (uint8_t)byte; /* the mirror the executor's dword reads rely on */
    }
  }
  f.lw += 1;
  f.run += 1;
}

/* ---- the lane-parallel front end ---- */

/* 64 stream bits from bit position p on (the top ones may be missing: at least 48 are there), per lane. */
__device__ __forceinline__ uint64_t bits_at(const lzw::InRing& ir, uint32_t p)
{
  const uint32_t a0 = (p >> 5) << 2;
  const uint32_t s = p & 31u;
  const uint32_t d0 = *(const uint32_t*)(ir.ring + (a0 & (lzw::kInRing - 1)));
  const uint32_t d1 = *(const uint32_t*)(ir.ring + ((a0 + 4) & (lzw::kInRing - 1)));
  const uint32_t d2 = *(const uint32_t*)(ir.ring + ((a0 + 8) & (lzw::kInRing - 1)));
  const uint64_t lo = (((uint64_t)d1 << 32) | d0) >> s;
  return s ? lo | ((uint64_t)d2 << (64 - s)) : lo;
}

/*
 * The symbol that starts with bit 0 of w, by table lookups alone, per lane: a literal (value = the byte, dist = 0)
 * or a length/distance pair with its extra bits (value = match length, dist = distance). Returns the bits it
 * takes (<= 48), or 0 when the lookups cannot tell: a code longer than a lookup, end of block, an illegal symbol.
 */
__device__ __forceinline__ uint32_t decode_at(const Code& ll, const Code& dd, uint64_t w, uint32_t& value, uint32_t& dist)
{
  /* straight-line: both lookups are made whatever the first one says, so that the four positions a lane decodes
   * speculatively have their LDS reads in flight together; the entries carry the extra-bit counts (lut_entry) */
  const uint32_t e = ll.lut[(uint32_t)w & ((1u << kLutBits) - 1u)];
  const uint32_t len = e & 15u;
  const bool is_len = (e & 0x8000u) != 0;
  const uint32_t sym = (e >> 4) & 0x1ffu; /* literal / 256 / 286+ when !is_len */
  const bool is_lit = !is_len && sym < 256;
  const uint32_t k = is_len ? (e >> 9) & 7u : 0u;
  const uint32_t used = len + k; /* <= 20 */
  const uint32_t e2 = dd.lut[(uint32_t)(w >> used) & ((1u << kDistLutBits) - 1u)];
  const uint32_t len2 = e2 & 15u, dsym = (e2 >> 4) & 31u, k2 = (e2 >> 9) & 15u;
  const uint32_t used2 = used + len2;
  const bool pair_ok = is_len && len2 != 0 && dsym <= 29;
  /* the fields (dead code where only the length is wanted: the speculative windows) */
  const uint32_t ls = (e >> 4) & 31u;
  const uint32_t base = ls < 8 ? ls + 3 : ls == 28 ? 258u : ((4 + (ls & 3u)) << k) + 3;
  const uint32_t mlen = base + ((uint32_t)(w >> len) & ((1u << k) - 1u));
  const uint32_t dbase = dsym < 4 ? dsym + 1 : ((2 + (dsym & 1u)) << k2) + 1;
  const uint32_t far = dbase + ((uint32_t)(w >> used2) & ((1u << k2) - 1u));
  value = is_lit ? sym : mlen;
  dist = is_lit ? 0u : far;
  return len == 0 ? 0u : is_lit ? len : pair_ok ? used2 + k2 : 0u;
}

/* The same wave-uniformly for the symbol at bit position p, codes longer than the lookups included (canonical walk).
 * 0: end of block or an illegal symbol -- the symbol-at-a-time decoder deals with those. */
__device__ __forceinline__ uint32_t uniform_symbol(
    const lzw::InRing& ir, const Code& ll, const Code& dd, uint32_t p, uint32_t& value, uint32_t& dist)
{
  const uint64_t w = wave::uniform64(bits_at(ir, p));
  const uint32_t e = wave::uniform(ll.lut[(uint32_t)w & ((1u << kLutBits) - 1u)]);
  uint32_t len = e & 15u, sym = entry_symbol(kLitLenCode, e);
  value = 0, dist = 0;
  if (len == 0) {
    sym = slow_symbol(ll, (uint32_t)w, len);
  }
  if (sym < 256) {
    value = sym;
    return len;
  }
  if (sym == 256 || sym > 285) {
    return 0;
  }
  const bool plain = sym < 265 || sym == 285;
  const uint32_t k = plain ? 0u : (sym - 261) >> 2;
  const uint32_t base = sym < 265 ? sym - 254 : sym == 285 ? 258u : ((4 + ((sym - 261) & 3u)) << k) + 3;
  value = base + ((uint32_t)(w >> len) & ((1u << k) - 1u));
  uint32_t used = len + k;
  const uint32_t e2 = wave::uniform(dd.lut[(uint32_t)(w >> used) & ((1u << kDistLutBits) - 1u)]);
  uint32_t len2 = e2 & 15u, dsym = entry_symbol(kDistCode, e2);
  if (len2 == 0) {
    dsym = slow_symbol(dd, (uint32_t)(w >> used), len2);
  }
  if (dsym > 29) {
    return 0;
  }
  used += len2;
  const uint32_t k2 = dsym < 4 ? 0u : (dsym >> 1) - 1;
  dist = (dsym < 4 ? dsym + 1 : ((2 + (dsym & 1u)) << k2) + 1) + ((uint32_t)(w >> used) & ((1u << k2) - 1u));
  return used + k2;
}

struct Scan
{
  uint32_t wb;    /* bit position of window slot 0 */
  uint32_t nx[kScanPer / 4]; /* lane l: bits the symbol at wb + kScanPer l + k would take, a byte each (0 = the lookups cannot tell) */
  uint8_t* tab;   /* LDS: kScanLevels tables of kScanWin bytes */
  bool built;     /* the tables are those of this block's codes */
};

/* Tables of the window that starts at bit position q: J_i[p] = bits from p to the 2^i-th symbol after it (255 = it
 * leaves the window, a symbol on the way cannot be told, or the sum does not fit a byte). lz_window.hip.h: chase_build,
 * on bit positions. */
__device__ __forceinline__ void scan_build(Scan& c, const lzw::InRing& ir, const Code& ll, const Code& dd, uint32_t q)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  c.wb = q;
  c.built = true;
  const uint32_t p0 = q + kScanPer * lane;
  const uint32_t a0 = (p0 >> 5) << 2;
  const uint32_t s0 = p0 & 31u;
  const uint32_t d0 = *(const uint32_t*)(ir.ring + (a0 & (lzw::kInRing - 1)));
  const uint32_t d1 = *(const uint32_t*)(ir.ring + ((a0 + 4) & (lzw::kInRing - 1)));
  const uint32_t d2 = *(const uint32_t*)(ir.ring + ((a0 + 8) & (lzw::kInRing - 1)));
  const uint64_t lo64 = ((uint64_t)d1 << 32) | d0;
  uint32_t a[kScanPer], nx[kScanPer];
#pragma unroll
  for (uint32_t k = 0; k < kScanPer; ++k) {
    const uint32_t s = s0 + k; /* <= 38: 58 bits at least, a symbol takes 48 at most */
    const uint64_t w = s ? (lo64 >> s) | ((uint64_t)d2 << (64 - s)) : lo64;
    uint32_t value, dist;
    nx[k] = decode_at(ll, dd, w, value, dist);
    a[k] = nx[k] != 0 && kScanPer * lane + k + nx[k] < kScanWin ? nx[k] : 255u;
  }
  /* the distances of a lane travel as dwords of two 16-bit lanes (positions 0|1, 2|3, ...): a doubling round is a packed
   * add + a packed saturation per pair, and one byte permute per four positions packs the table words */
  uint32_t pa[kScanPer / 2];
#pragma unroll
  for (uint32_t j = 0; j < kScanPer / 2; ++j) {
    pa[j] = a[2 * j] | (a[2 * j + 1] << 16);
  }
#pragma unroll
  for (uint32_t j = 0; j < kScanPer / 4; ++j) {
    c.nx[j] = nx[4 * j] | (nx[4 * j + 1] << 8) | (nx[4 * j + 2] << 16) | (nx[4 * j + 3] << 24);
    *(uint32_t*)(c.tab + kScanPer * lane + 4 * j) = wave::perm_bytes(pa[2 * j + 1], pa[2 * j], 0x06040200u);
  }
  wave::sync();
#pragma unroll
  for (uint32_t i = 1; i < kScanLevels; ++i) {
    const uint8_t* prev = c.tab + (i - 1) * kScanWin + kScanPer * lane;
    /* a == 255 reads past its table (into the next one): the sum saturates anyway */
    uint32_t g[kScanPer];
#pragma unroll
    for (uint32_t k = 0; k < kScanPer; ++k) {
      g[k] = prev[k + ((pa[k / 2] >> (16 * (k & 1u))) & 0xffffu)];
    }
    if (i + 1 < kScanLevels) {
#pragma unroll
      for (uint32_t j = 0; j < kScanPer / 2; ++j) {
        pa[j] = wave::pk_add_sat255(pa[j], g[2 * j] | (g[2 * j + 1] << 16));
      }
    } else {
      /* the top table (16 symbols ahead) holds the distance MINUS kTopBias: sixteen symbols of a zlib stream take 190 bits
       * on average and often more than a byte can say (matches are 20 to 48 bits each), and an entry that saturates ends
       * the enumeration at its rank */
#pragma unroll
      for (uint32_t j = 0; j < kScanPer / 2; ++j) {
        uint32_t both = 0;
#pragma unroll
        for (uint32_t h = 0; h < 2; ++h) {
          const uint32_t x = (pa[j] >> (16 * h)) & 0xffffu, y = g[2 * j + h];
          const uint32_t sum = x + y - kTopBias; /* wraps when the sum is below the bias */
          both |= (x != 255u && y != 255u && sum < 255u ? sum : 255u) << (16 * h);
        }
        pa[j] = both;
      }
    }
#pragma unroll
    for (uint32_t j = 0; j < kScanPer / 4; ++j) {
      *(uint32_t*)(c.tab + i * kScanWin + kScanPer * lane + 4 * j) = wave::perm_bytes(pa[2 * j + 1], pa[2 * j], 0x06040200u);
    }
    wave::sync();
  }
}

/*
 * One round of the lane-parallel front end: up to 64 symbols from bit position q on become records and literal
 * bytes. Stops early where a symbol cannot be told by lookups (`stuck`: the caller decodes that one wave-uniformly),
 * where the batch is full, or where the resident part of the stream ends. Returns the bit position to go on from.
 */
__device__ __forceinline__ uint32_t scan_round(
    Scan& c, Front& f, const lzw::InRing& ir, const Code& ll, const Code& dd, uint32_t q, bool& stuck)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  uint32_t tokpos = 0; /* lane k: bit position of the round's k-th symbol */
  bool told = false;   /* ... which was decoded wave-uniformly (a long code): its fields are here */
  uint32_t told_value = 0, told_dist = 0;
  uint32_t t = 0;
  stuck = false;
  const uint32_t cap = 64 - f.n; /* a symbol makes at most one record */
  while (t < cap) {
    if (!c.built || q - c.wb >= kScanWin) {
      if ((q >> 3) + kScanWin / 8 + 32 > ir.hi && ir.hi < ir.vend) {
        break; /* the window would look at bytes that are not resident yet */
      }
      LZW_T(15);
      scan_build(c, ir, ll, dd, q);
      LZ_STAT("deflate_builds", 1);
      LZW_T(1);
    }
    /* Lane n: the n-th symbol from q, if the chain gets there inside this window. Lane arithmetic only (lz_window.hip.h:
     * chase_tokens): a lane follows the levels named by the bits of n; a level it does not take adds 0; 255 poisons the
     * lane through `worst`; the position wraps inside the table. Ranks 32 and up start at the 32nd symbol, two steps of the
     * 16-symbol table from the start (every lane reads the same two bytes). */
    const uint32_t pos0 = q - c.wb;
    const uint32_t top = (kScanLevels - 1) * kScanWin;
    const uint32_t t1 = c.tab[top + pos0];
    const uint32_t t2 = c.tab[top + ((pos0 + t1 + kTopBias) & (kScanWin - 1))];
    const bool upper = (lane & 32u) != 0;
    uint32_t pos = upper ? (pos0 + t1 + t2 + 2 * kTopBias) & (kScanWin - 1) : pos0;
    uint32_t worst = upper ? (t1 > t2 ? t1 : t2) : 0u;
#pragma unroll
    for (uint32_t i = 0; i < kScanLevels; ++i) {
      const uint32_t a = c.tab[i * kScanWin + pos];
      const uint32_t take_it = (uint32_t)(-(int32_t)((lane >> i) & 1u));
      const uint32_t adv = a & take_it;
      worst = adv > worst ? adv : worst;
      pos = (pos + adv + (i + 1 == kScanLevels ? kTopBias & take_it : 0u)) & (kScanWin - 1);
    }
    /* Every n appears in exactly one lane; lane 0 is always valid. The valid n are NOT always a prefix here: a jump of 8 or
     * 16 symbols can be 255 bits and more (long matches take 20 to 48 bits each) without leaving a window of 512, and the
     * byte tables cannot say so -- the lanes behind the first such rank wait for the next enumeration, which starts at it. */
    const uint64_t invalid = ~wave::ballot(worst != 255u);
    uint32_t count = invalid ? wave::ctz64(invalid) : 64u;
    /* the chain's last symbol in hand: its own length says where to go on */
    const uint32_t last = wave::read_lane(pos, count - 1);
    const uint32_t word = kScanPer == 8 && (last & 4u) ? c.nx[kScanPer / 4 - 1] : c.nx[0];
    uint32_t d = (wave::read_lane(word, last / kScanPer) >> (8 * (last & 3u))) & 0xffu;
    uint32_t uval = 0, udist = 0;
    bool resolved = false;
    LZ_STAT("deflate_enumerations", 1);
    LZ_STAT("deflate_enumerated_symbols", count);
    if (d == 0) { /* the lookups could not tell: a long code is walked here, the chain goes on */
      LZ_STAT("deflate_uniform_symbols", 1);
      d = uniform_symbol(ir, ll, dd, c.wb + last, uval, udist);
      resolved = d != 0;
      if (!resolved) {
        count -= 1; /* end of block, or illegal: not part of the round */
      }
    }
    const uint32_t room = cap - t;
    const uint32_t take = count < room ? count : room;
    const uint32_t shifted = t ? wave::shuffle(pos, (lane - t) & 63u) : pos;
    if (lane >= t && lane < t + take) {
      tokpos = c.wb + shifted;
    }
    if (resolved && take == count && lane == t + count - 1) {
      told = true;
      told_value = uval;
      told_dist = udist;
    }
    t += take;
    if (take < count) {
      q = c.wb + wave::read_lane(pos, take);
      break;
    }
    q = c.wb + last + d; /* d == 0: stays on the symbol that cannot be told */
    if (d == 0) {
      stuck = true;
      break;
    }
    LZW_T(2);
  }
  LZW_T(2);
  if (t == 0) {
    return q;
  }
  LZ_STAT("deflate_rounds", 1);
  LZ_STAT("deflate_round_symbols", t);
  /* ---- lane k decodes symbol k ---- */
  uint32_t value = 0, dist = 0;
  const bool mine0 = lane < t;
  if (mine0) {
    (void)decode_at(ll, dd, bits_at(ir, tokpos), value, dist);
  }
  if (told) {
    value = told_value;
    dist = told_dist;
  }
  const bool is_match0 = mine0 && dist != 0;
  const uint32_t size = mine0 ? (is_match0 ? value : 1u) : 0u;
  const uint32_t incl = wave::scan_add_inclusive(size);
  /* the batch holds kBatchMax output bytes: the pending literal run counts, it joins the first record */
  const uint32_t budget = lzw::kBatchMax - f.bytes - f.run;
  const uint64_t over = wave::ballot(mine0 && incl > budget);
  if (over) {
    const uint32_t cut = wave::ctz64(over); /* >= 1: one symbol is at most 258 bytes */
    q = wave::read_lane(tokpos, cut);
    t = cut;
    stuck = false;
  }
  const bool mine = lane < t;
  const bool is_match = mine && dist != 0;
  const bool is_lit = mine && dist == 0;
  const uint64_t lits = wave::ballot(is_lit);
  const uint64_t matches = wave::ballot(is_match);
  const uint64_t below = (1ull << lane) - 1ull;
  const uint32_t cum = wave::popc64(lits & below); /* literal symbols before this lane */
  if (is_lit) {
    const uint32_t at = (f.lw + cum) & (lzw::kInRing - 1);
    f.lit.ring[at] = (uint8_t)value;
    if (at < 16) {
      f.lit.ring[lzw::kInRing + at] = (uint8_t)value;
    }
  }
  const uint32_t n_lit = wave::popc64(lits);
  const uint32_t n_match = wave::popc64(matches);
  if (is_match) {
    const uint64_t prev = matches & below;
    const uint32_t before = prev ? wave::popc64(lits & ((1ull << (63 - (uint32_t)__builtin_clzll(prev))) - 1ull)) : 0u;
    const uint32_t run = prev ? cum - before : cum + f.run;
    const uint32_t src = prev ? f.lw + before : f.lw - f.run;
    const uint32_t r = f.n + wave::popc64(prev);
    f.rec[2 * r] = src;
    f.rec[2 * r + 1] = pack_record(run, value, dist);
  }
  if (n_match) {
    const uint32_t jl = 63 - (uint32_t)__builtin_clzll(matches);
    const uint32_t upto = wave::read_lane(incl, jl);
    f.bytes += f.run + upto;
    f.run = n_lit - wave::popc64(lits & ((1ull << jl) - 1ull));
    f.n += n_match;
  } else {
    f.run += n_lit;
  }
  f.lw += n_lit;
  LZW_T(3);
  return q;
}

/*
 * Decode one chunk. `flags & kGzip`: the chunk is a gzip member (header skipped, ISIZE checked). SIZE_ONLY: nothing
 * is written, the return value is the uncompressed size (mod 2^32). Returns the bytes produced.
 */
template <bool CHECKED, bool SIZE_ONLY>
__device__ __forceinline__ uint32_t decode_chunk(
    const uint8_t* __restrict__ in, uint32_t in_len, uint8_t* out, uint32_t out_cap, uint8_t* lds, uint32_t flags, uint32_t& err)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  err = lz::kErrNone;
  if (in_len == 0) {
    err = lz::kErrInput;
    return 0;
  }
  lzw::InRing ir;
  lzw::OutWindow ow;
  lzw::in_init(ir, in, in_len, lds + lzw::kOutLds);
  lzw::out_init(ow, out, lds);
  uint8_t* front = lds + lzw::kOutLds + lzw::kInLds;
  lzw::in_ensure(ir, ir.vbeg, ir.vbeg + 2 * lzw::kInBlock);

  const Code ll = {(uint16_t*)(front + kOffLutLL), (uint16_t*)(front + kOffCntLL), (uint16_t*)(front + kOffSymLL)};
  const Code dd = {(uint16_t*)(front + kOffLutD), (uint16_t*)(front + kOffCntD), (uint16_t*)(front + kOffSymD)};
  uint8_t* lens = front + kOffRec;

  Front f;
  f.lit.base = nullptr;
  f.lit.ring = front + kOffLit;
  f.lit.vbeg = 0, f.lit.vend = ~0u, f.lit.lo = 0, f.lit.hi = 0;
  f.lw = 0, f.run = 0, f.n = 0, f.bytes = 0, f.carry0 = 0, f.carry1 = 0, f.carried = false, f.produced = 0;
  f.rec = (uint32_t*)(front + kOffRec);

  uint32_t start = ir.vbeg;
  uint32_t stream_end = ir.vend; /* where the deflate stream must have ended by */
  if (flags & kGzip) {
    /* RFC 1952: ID1 ID2 CM FLG MTIME(4) XFL OS [XLEN + extra] [name 0] [comment 0] [CRC16] ... CRC32 ISIZE */
    if (in_len < 18) {
      err = lz::kErrInput;
      return 0;
    }
    const uint32_t id = lzw::in_byte_uniform(ir, start) | (lzw::in_byte_uniform(ir, start + 1) << 8)
                        | (lzw::in_byte_uniform(ir, start + 2) << 16);
    const uint32_t flg = lzw::in_byte_uniform(ir, start + 3);
    if (id != 0x088b1fu || (flg & 0xe0u)) {
      err = lz::kErrInput;
      return 0;
    }
    uint32_t p = start + 10;
    if (flg & 4u) {
      if (p + 2 > ir.vend) {
        err = lz::kErrInput;
        return 0;
      }
      p += 2 + (lzw::in_byte_uniform(ir, p) | (lzw::in_byte_uniform(ir, p + 1) << 8));
    }
    for (uint32_t field = 8; field <= 16; field += 8) { /* FNAME, FCOMMENT: zero-terminated */
      if (flg & field) {
        while (p < ir.vend && lzw::in_byte_uniform(ir, p) != 0) {
          ++p;
        }
        ++p;
      }
    }
    if (flg & 2u) {
      p += 2;
    }
    if (p + 8 > ir.vend) {
      err = lz::kErrInput;
      return 0;
    }
    start = p;
    stream_end = ir.vend - 8;
  }

  Scan sc;
  sc.wb = 0, sc.built = false, sc.tab = front + kOffScan;
  for (uint32_t j = 0; j < kScanPer / 4; ++j) {
    sc.nx[j] = 0;
  }
  Bits b;
  lzw::in_ensure(ir, start, (start & ~(lzw::kInBlock - 1)) + 2 * lzw::kInBlock);
  b.seek(ir, start);
  uint32_t op = 0;
  const uint32_t limit = out_cap;
  bool bad = false;

  /* One loop, one place where a batch is executed (the executor is large: a single copy of it in the kernel). Each
   * turn does one thing -- read a block header, decode symbols, move stored bytes -- and then runs whatever records
   * are in hand. A header is read only when none are: the code lengths are written where the records are kept. */
  enum : uint32_t { kHeader, kSymbols, kStored, kFlush, kDone };
  uint32_t state = kHeader;
  bool last = false;         /* the block being decoded is the final one */
  uint32_t stored_from = 0;  /* kStored: the next byte of the block in the stream, and how many are left */
  uint32_t stored_left = 0;
  for (;;) {
    if (state == kHeader && f.n == 0 && !f.carried) {
      if (last) {
        state = kFlush;
      } else {
        lzw::in_ensure(ir, b.byte_pos() & ~3u, (b.byte_pos() & ~(lzw::kInBlock - 1)) + 2 * lzw::kInBlock);
        b.refill(ir);
        last = b.take(1) != 0;
        const uint32_t type = b.take(2);
        if (type == 3 || b.byte_pos() > stream_end) {
          bad = true;
          break;
        }
        if (type == 0) {
          /* stored: LEN, ~LEN, then LEN bytes from the next byte boundary */
          b.drop(b.cnt & 7u);
          b.refill(ir);
          const uint32_t len = b.take(16);
          const uint32_t nlen = b.take(16);
          stored_from = b.byte_pos();
          stored_left = len;
          if ((len ^ nlen) != 0xffffu || stored_from + len > stream_end) {
            bad = true;
            break;
          }
          if (SIZE_ONLY) {
            f.produced += len;
            stored_from += len;
            stored_left = 0;
          } else if (f.run != 0) {
            close_sequence<SIZE_ONLY>(f, 0, 0);
          }
          state = kStored;
        } else {
          uint32_t n_d = kDistSyms;
          if (type == 1) {
            /* fixed code: 8 bits for 0-143, 9 for 144-255, 7 for 256-279, 8 for 280-287; distances 5 bits */
            for (uint32_t i = lane; i < kLitLenSyms + kDistSyms; i += 64) {
              lens[i] = (uint8_t)(i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : i < 288 ? 8 : 5);
            }
            wave::sync();
          } else {
            b.refill(ir);
            const uint32_t n_ll = 257 + b.take(5);
            n_d = 1 + b.take(5);
            const uint32_t n_cl = 4 + b.take(4);
            if (n_ll > 286 || n_d > 30) {
              bad = true;
              break;
            }
            /* the code-length code: 3-bit lengths in the order 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15 */
            uint32_t cl_len[19];
#pragma unroll
            for (uint32_t i = 0; i < 19; ++i) {
              cl_len[i] = 0;
              if (i < n_cl) {
                b.refill(ir);
                cl_len[i] = b.take(3);
              }
            }
            if (lane < 19) {
              constexpr uint8_t kOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
              uint32_t mine = 0;
#pragma unroll
              for (uint32_t i = 0; i < 19; ++i) {
                mine = kOrder[i] == lane ? cl_len[i] : mine;
              }
              lens[lane] = (uint8_t)mine;
            }
            wave::sync();
            if (!build_code<kClLutBits, kPlainCode>(dd, lens, 19)) {
              bad = true;
              break;
            }
            /* the literal/length and distance code lengths, run-length coded with that code; the distance lengths
             * are kept behind the 288 literal/length ones */
            uint32_t i = 0, prev = 0;
            while (i < n_ll + n_d && !bad) {
              if (b.next + 16 > ir.hi && ir.hi < ir.vend) {
                lzw::in_ensure(ir, b.byte_pos() & ~3u, (b.byte_pos() & ~(lzw::kInBlock - 1)) + 2 * lzw::kInBlock);
              }
              b.refill(ir);
              const uint32_t sym = next_symbol<kClLutBits, kPlainCode>(dd, b, bad);
              uint32_t rep = 1, val = sym;
              if (sym == 16) {
                rep = 3 + b.take(2);
                val = prev;
                bad = bad || i == 0;
              } else if (sym == 17) {
                rep = 3 + b.take(3);
                val = 0;
              } else if (sym == 18) {
                rep = 11 + b.take(7);
                val = 0;
              } else if (sym > 18) {
                bad = true;
              }
              if (bad || i + rep > n_ll + n_d || b.byte_pos() > stream_end) {
                bad = true;
                break;
              }
              for (uint32_t k = 0; k < rep; ++k) {
                const uint32_t at = i + k < n_ll ? i + k : kLitLenSyms + (i + k - n_ll);
                lens[at] = (uint8_t)val; /* every lane writes the same byte */
              }
              prev = val;
              i += rep;
            }
            if (bad) {
              break;
            }
            wave::sync();
            for (uint32_t k = n_ll + lane; k < kLitLenSyms; k += 64) {
              lens[k] = 0;
            }
            wave::sync();
            if (wave::uniform(lens[256]) == 0) { /* no end-of-block code */
              bad = true;
              break;
            }
          }
          if (!build_code<kLutBits, kLitLenCode>(ll, lens, kLitLenSyms) || !build_code<kDistLutBits, kDistCode>(dd, lens + kLitLenSyms, n_d)) {
            bad = true;
            break;
          }
          sc.built = false;
          state = kSymbols;
          LZW_T(0);
        }
      }
    } else if (state == kSymbols && !f.carried) {
      {
        /* the round reads the stream ring from the byte the current BIT lies in (the bit buffer may already hold it) */
        const uint32_t at = b.bit_pos() >> 3;
        lzw::in_ensure(ir, at & ~3u, (at & ~(lzw::kInBlock - 1)) + 2 * lzw::kInBlock);
      }
      bool one_symbol = SIZE_ONLY; /* SIZE_ONLY: the whole block symbol by symbol; else: what the rounds could not tell */
      LZW_T(15);
      if (!SIZE_ONLY) {
        /* rounds until the batch is nearly full (the executor costs the same for 20 records as for 64), a symbol
         * cannot be told, or the stream ring has to move on */
        uint32_t q = b.bit_pos();
        for (;;) {
          if (f.run >= kRunClose) {
            close_sequence<SIZE_ONLY>(f, 0, 0);
          }
          if (f.carried || f.n > 40 || f.bytes + f.run > lzw::kBatchMax - 320 || (q >> 3) + 256 > ir.hi + (ir.hi >= ir.vend ? 4096u : 0u)) {
            break;
          }
          const uint32_t from = q;
          q = scan_round(sc, f, ir, ll, dd, q, one_symbol);
          if (one_symbol || q == from || (q >> 3) > stream_end) {
            break;
          }
        }
        b.seek_bit(ir, q);
        if (b.byte_pos() > stream_end) {
          bad = true;
          break;
        }
      }
      LZW_T(15);
      while (one_symbol && !f.carried) {
        if (b.next + 16 > ir.hi && ir.hi < ir.vend) {
          lzw::in_ensure(ir, b.byte_pos() & ~3u, (b.byte_pos() & ~(lzw::kInBlock - 1)) + 2 * lzw::kInBlock);
        }
        if (b.byte_pos() > stream_end) {
          bad = true;
          break;
        }
        one_symbol = SIZE_ONLY;
        b.refill(ir);
        const uint32_t sym = next_symbol<kLutBits, kLitLenCode>(ll, b, bad);
        if (sym < 256) {
          if (f.run == kRunMax) {
            close_sequence<SIZE_ONLY>(f, 0, 0);
          }
          put_literal<SIZE_ONLY>(f, sym);
          continue;
        }
        if (sym == 256) {
          state = kHeader;
          break;
        }
        if (sym > 285 || bad) {
          bad = true;
          break;
        }
        uint32_t mlen;
        if (sym < 265) {
          mlen = sym - 254;
        } else if (sym == 285) {
          mlen = 258;
        } else {
          const uint32_t k = (sym - 261) >> 2;
          mlen = ((4 + ((sym - 261) & 3u)) << k) + 3 + b.take(k);
        }
        b.refill(ir);
        const uint32_t dsym = next_symbol<kDistLutBits, kDistCode>(dd, b, bad);
        if (dsym > 29 || bad) {
          bad = true;
          break;
        }
        uint32_t dist;
        if (dsym < 4) {
          dist = dsym + 1;
        } else {
          const uint32_t k = (dsym >> 1) - 1;
          dist = ((2 + (dsym & 1u)) << k) + 1 + b.take(k);
        }
        close_sequence<SIZE_ONLY>(f, mlen, dist);
      }
      LZW_T(10);
      if (bad) {
        break;
      }
    } else if (state == kStored && f.n == 0 && !f.carried) {
      /* the bytes travel stream ring -> literal ring -> window, at most kBatchMax per turn */
      if (stored_left == 0) {
        lzw::in_ensure(ir, stored_from & ~3u, (stored_from & ~(lzw::kInBlock - 1)) + 2 * lzw::kInBlock);
        b.seek(ir, stored_from);
        state = kHeader;
      } else {
        const uint32_t now = stored_left < lzw::kBatchMax ? stored_left : lzw::kBatchMax;
        lzw::in_ensure(ir, stored_from, stored_from + now);
        for (uint32_t i = lane; i < now; i += 64) {
          const uint32_t at = (f.lw + i) & (lzw::kInRing - 1);
          const uint8_t v = (uint8_t)lzw::in_byte(ir, stored_from + i);
          f.lit.ring[at] = v;
          if (at < 16) {
            f.lit.ring[lzw::kInRing + at] = v;
          }
        }
        const uint32_t recs = (now + kRunMax - 1) / kRunMax;
        if (lane < recs) {
          f.rec[2 * lane] = f.lw + kRunMax * lane;
          f.rec[2 * lane + 1] = lane + 1 < recs ? kRunMax : now - kRunMax * (recs - 1);
        }
        f.lw += now;
        f.n = recs;
        f.bytes = now;
        stored_from += now;
        stored_left -= now;
      }
    } else if (state == kFlush && !f.carried) {
      if (b.byte_pos() > stream_end) { /* bits were taken from behind the end of the stream */
        bad = true;
        break;
      }
      if (!SIZE_ONLY && f.run != 0) {
        close_sequence<SIZE_ONLY>(f, 0, 0);
      }
      state = kDone;
    }
    if (!SIZE_ONLY && (f.n != 0 || f.carried)) {
      if (!drain<CHECKED>(f, ow, limit, op, err)) {
        return 0;
      }
    } else if (state == kDone) {
      break;
    }
  }
  if (bad) {
    err |= lz::kErrInput;
    return 0;
  }
  if (SIZE_ONLY) {
    return (uint32_t)(f.produced + f.run);
  }
  lzw::out_flush_all(ow, op);
  if (flags & kGzip) {
    const uint32_t t = ir.vend - 4;
    const uint32_t isize = lzw::in_byte_uniform(ir, t) | (lzw::in_byte_uniform(ir, t + 1) << 8)
                           | (lzw::in_byte_uniform(ir, t + 2) << 16) | (lzw::in_byte_uniform(ir, t + 3) << 24);
    if (CHECKED && isize != op) {
      err |= lz::kErrInput;
      return 0;
    }
  }
  return op;
}

} // namespace deflate
