/*
 * deflate/deflate_encode_dynamic.hip.h -- DEFLATE blocks with per-chunk ("dynamic") Huffman codes, RFC 1951 3.2.7:
 * nvcompBatchedDeflateOpts_t.algo >= 1. One wavefront per chunk, three steps:
 *   1. count:  the match finder (common/lz_match.hip.h) runs over the chunk with a sink that only counts the
 *              literal/length and distance symbols its sequences would use (LDS histograms, ds_add_u32);
 *   2. build:  two length-limited canonical Huffman codes from the counts -- rank sort by the lanes, the two-queue
 *              merge and the overflow repair of the classic CPU construction wave-uniformly (a few thousand scalar
 *              instructions per chunk against the millions of the match finder), code words by the lanes;
 *   3. emit:   the match finder runs again and the sequences are written with those codes, through the same bit
 *              sink as the fixed-code compressor (deflate_encode.hip.h).
 * The two runs of the match finder need not pick identical sequences (two lanes that insert into the same hash slot
 * race, benignly): every symbol of both alphabets gets a count of at least one, so every symbol has a code word.
 * The block header writes the 316 code lengths one by one with a code-length code built the same way (no repeat
 * symbols: ~140 bytes per chunk, 0.5 % of a compressed 64 KiB chunk of text).
 */
#pragma once

#include "deflate/deflate_encode.hip.h"

namespace deflate {

constexpr uint32_t kNumLL = 286, kNumD = 30, kNumCL = 19;
constexpr uint32_t kCodeMaxBits = 15; /* longest code word of the format (the code-length code: 7) */
constexpr uint32_t kDynLaneLits = 8; /* literal run a lane writes by itself: 8 x 15 + 48 bits x 64 lanes fit the staging area */

/* code word (bit-reversed, ready to be ORed in) | length << 16 */
struct CodeTables
{
  uint32_t* ll; /* LDS: kNumLL entries; the counts while they are taken */
  uint32_t* d;  /* LDS: kNumD entries */
};
constexpr uint32_t kCodeLds = (kNumLL + kNumD + 4) * 4;

/* ---- step 1: counting sink ---- */
struct CountEmitter
{
  static constexpr bool kStream = true;
  static constexpr uint32_t kReach = 32768;

  static __device__ __forceinline__ void count_match(const CodeTables& t, uint32_t match_len, uint32_t offset)
  {
    /* wave-uniform arguments; lane 0 counts */
    uint32_t k, extra;
    const uint32_t dsym = distance_symbol(offset, k, extra);
    uint32_t pieces = 0;
    while (match_len != 0) {
      const uint32_t piece = next_piece(match_len);
      if (wave::lane_id() == 0) {
        atomicAdd(t.ll + length_symbol(piece, k, extra), 1u);
      }
      match_len -= piece;
      ++pieces;
    }
    if (wave::lane_id() == 0 && pieces != 0) {
      atomicAdd(t.d + dsym, pieces);
    }
  }

  static __device__ __forceinline__ void one(CodeTables& t, const uint8_t* lit, uint32_t lit_len, uint32_t offset, uint32_t match_len)
  {
    const uint32_t lane = (uint32_t)wave::lane_id();
    for (uint32_t base = lane; base < lit_len; base += 64) {
      atomicAdd(t.ll + lit[base], 1u);
    }
    count_match(t, match_len, offset);
  }

  static __device__ __forceinline__ void window(
      CodeTables& t, const uint8_t* __restrict__ src, bool sel, uint32_t lit_from, uint32_t lit_len, uint32_t match_len,
      uint32_t offset, uint64_t before8, bool before_ok, uint32_t run)
  {
    /* the same split as the emitting sink's: short sequences per lane, the others one after the other */
    const bool alone = sel && match_len <= kMaxMatch && lit_len <= kDynLaneLits && (before_ok || lit_len == 0);
    uint64_t lits = 0;
    if (alone && lit_len != 0) {
      lits = before8 >> (8 * (8 - run));
    }
    for (uint32_t i = 0; wave::ballot(alone && i < lit_len) != 0; ++i) {
      if (alone && i < lit_len) {
        atomicAdd(t.ll + ((uint32_t)(lits >> (8 * i)) & 0xffu), 1u);
      }
    }
    if (alone) {
      uint32_t k, extra;
      atomicAdd(t.ll + length_symbol(match_len, k, extra), 1u);
      atomicAdd(t.d + distance_symbol(offset, k, extra), 1u);
    }
    uint64_t rest = wave::ballot(sel && !alone);
    while (rest) {
      const uint32_t j = wave::ctz64(rest);
      rest &= rest - 1;
      one(t, src + wave::read_lane(lit_from, j), wave::read_lane(lit_len, j), wave::read_lane(offset, j),
          wave::read_lane(match_len, j));
    }
  }
};

/* ---- step 2: code construction ---- */

/* Scratch of one construction (LDS; the match finder's hash table lends it between the two runs). n <= 286. */
struct BuildScratch
{
  uint32_t* weight; /* [2 n]: the leaves in ascending order of count, then the internal nodes as they are made */
  uint16_t* parent; /* [2 n] */
  uint16_t* order;  /* [n]: symbol of leaf i */
  uint8_t* lens;    /* [n]: code length per SYMBOL (the result) */
};
constexpr uint32_t kBuildLds = 2 * kNumLL * 4 + 2 * kNumLL * 2 + kNumLL * 2 + ((kNumLL + 15) & ~15u);

__device__ __forceinline__ BuildScratch carve_scratch(uint8_t* lds)
{
  BuildScratch b;
  b.weight = (uint32_t*)lds;
  b.parent = (uint16_t*)(lds + 2 * kNumLL * 4);
  b.order = (uint16_t*)(lds + 2 * kNumLL * 4 + 2 * kNumLL * 2);
  b.lens = lds + 2 * kNumLL * 4 + 2 * kNumLL * 2 + kNumLL * 2;
  return b;
}

/*
 * Code lengths of a Huffman code for counts[0, n) (every count >= 1), none longer than max_bits; the canonical code
 * words go to out[s] = reversed code | length << 16. n >= 2.
 */
__device__ __forceinline__ void build_code_words(const uint32_t* counts, uint32_t n, uint32_t max_bits, const BuildScratch& b, uint32_t* out)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  /* leaves in ascending order of (count, symbol): every symbol finds its rank */
  for (uint32_t s = lane; s < n; s += 64) {
    const uint32_t f = counts[s];
    uint32_t rank = 0;
    for (uint32_t t = 0; t < n; ++t) {
      const uint32_t g = counts[t];
      rank += (g < f || (g == f && t < s)) ? 1u : 0u;
    }
    b.order[rank] = (uint16_t)s;
    b.weight[rank] = f;
  }
  wave::sync();
  /* the two-queue merge: the two lightest of {next leaf, next unmerged internal node}, n - 1 times (wave-uniform) */
  {
    uint32_t leaf = 0, inner = n, made = n;
    for (uint32_t k = 0; k + 1 < n; ++k) {
      uint32_t pick[2];
#pragma unroll
      for (uint32_t j = 0; j < 2; ++j) {
        const bool have_leaf = leaf < n, have_inner = inner < made;
        const uint32_t wl = have_leaf ? wave::uniform(b.weight[leaf]) : 0u;
        const uint32_t wi = have_inner ? wave::uniform(b.weight[inner]) : 0u;
        const bool take_leaf = have_leaf && (!have_inner || wl <= wi);
        pick[j] = take_leaf ? leaf : inner;
        leaf += take_leaf ? 1u : 0u;
        inner += take_leaf ? 0u : 1u;
      }
      const uint32_t w = wave::uniform(b.weight[pick[0]]) + wave::uniform(b.weight[pick[1]]);
      b.weight[made] = w; /* every lane writes the same words */
      b.parent[pick[0]] = (uint16_t)made;
      b.parent[pick[1]] = (uint16_t)made;
      wave::sync();
      ++made;
    }
  }
  const uint32_t root = 2 * n - 2;
  /* depth of every leaf, capped; counts per length */
  uint32_t count[kCodeMaxBits + 1];
#pragma unroll
  for (uint32_t l = 0; l <= kCodeMaxBits; ++l) {
    count[l] = 0;
  }
  for (uint32_t base = 0; base < n; base += 64) {
    uint32_t depth = 0;
    if (base + lane < n) {
      uint32_t at = base + lane;
      while (at != root) {
        at = b.parent[at];
        ++depth;
      }
    }
    depth = depth > max_bits ? max_bits : depth; /* too deep: cut, the code is repaired below */
#pragma unroll
    for (uint32_t l = 1; l <= kCodeMaxBits; ++l) {
      count[l] += wave::popc64(wave::ballot(depth == l));
    }
  }
  /* Leaves cut to max_bits over-subscribe the code. In units of 2^-max_bits the code space used is
   * K = sum count[l] << (max_bits - l) and must come to exactly 1 << max_bits: while it is too much, a leaf of the
   * deepest populated length below max_bits moves one level down (halving what it takes); what that overshoots is
   * given back by moving leaves one level up, deepest first. */
  {
    const uint32_t full = 1u << max_bits;
    uint32_t used = 0;
#pragma unroll
    for (uint32_t l = 1; l <= kCodeMaxBits; ++l) {
      used += l <= max_bits ? count[l] << (max_bits - l) : 0u;
    }
    while (used > full) {
      uint32_t bits = 0;
#pragma unroll
      for (uint32_t l = kCodeMaxBits - 1; l >= 1; --l) {
        if (bits == 0 && l < max_bits && count[l] != 0) {
          bits = l;
        }
      }
#pragma unroll
      for (uint32_t l = 1; l <= kCodeMaxBits; ++l) {
        count[l] += l == bits + 1 ? 1u : l == bits ? ~0u : 0u;
      }
      used -= 1u << (max_bits - bits - 1);
    }
    while (used < full) {
      const uint32_t room = full - used;
      uint32_t level = 0;
#pragma unroll
      for (uint32_t l = kCodeMaxBits; l >= 2; --l) {
        if (level == 0 && l <= max_bits && count[l] != 0 && (1u << (max_bits - l)) <= room) {
          level = l;
        }
      }
#pragma unroll
      for (uint32_t l = 1; l <= kCodeMaxBits; ++l) {
        count[l] += l + 1 == level ? 1u : l == level ? ~0u : 0u;
      }
      used += 1u << (max_bits - level);
    }
  }
  /* lengths by rank: the rarest symbols get the longest code words */
  uint32_t upto[kCodeMaxBits + 2]; /* leaves with rank < upto[l] have a length >= l */
  upto[kCodeMaxBits + 1] = 0;
#pragma unroll
  for (uint32_t l = kCodeMaxBits; l >= 1; --l) {
    upto[l] = upto[l + 1] + count[l];
  }
  for (uint32_t r = lane; r < n; r += 64) {
    uint32_t len = 1;
#pragma unroll
    for (uint32_t l = 2; l <= kCodeMaxBits; ++l) {
      len = r < upto[l] ? l : len;
    }
    b.lens[b.order[r]] = (uint8_t)len;
  }
  wave::sync();
  /* canonical code words: per length consecutive, in symbol order */
  uint32_t next[kCodeMaxBits + 1];
  next[0] = 0;
  {
    uint32_t code = 0;
#pragma unroll
    for (uint32_t l = 1; l <= kCodeMaxBits; ++l) {
      code = (code + (l == 1 ? 0u : count[l - 1])) << 1;
      next[l] = code;
    }
  }
  for (uint32_t base = 0; base < n; base += 64) {
    const uint32_t len = base + lane < n ? b.lens[base + lane] : 0u;
#pragma unroll
    for (uint32_t l = 1; l <= kCodeMaxBits; ++l) {
      const uint64_t same = wave::ballot(len == l);
      if (len == l) {
        out[base + lane] = rev(next[l] + wave::prefix_popc(same), l) | (l << 16);
      }
      next[l] += wave::popc64(same);
    }
  }
  wave::sync();
}

/* ---- step 3: emitting sink ---- */
struct DynSink
{
  BitSink bits;
  CodeTables codes;
};

/* up to 64 bits at bit position p, per lane */
__device__ __forceinline__ void put64(const BitSink& s, uint32_t p, uint64_t code)
{
  const uint32_t at = (p - s.base) >> 5;
  const uint32_t sh = p & 31u;
  const uint32_t w0 = (uint32_t)code << sh;
  const uint64_t rest = sh ? code >> (32 - sh) : code >> 32; /* what did not fit the first dword */
  if (w0 != 0) {
    wave::lds_or(s.stage + at, w0);
  }
  if ((uint32_t)rest != 0) {
    wave::lds_or(s.stage + at + 1, (uint32_t)rest);
  }
  if ((uint32_t)(rest >> 32) != 0) {
    wave::lds_or(s.stage + at + 2, (uint32_t)(rest >> 32));
  }
}

/* <length, distance> with the chunk's codes: at most 15 + 5 + 15 + 13 = 48 bits */
__device__ __forceinline__ uint64_t dynamic_pair(const CodeTables& t, uint32_t mlen, uint32_t dist, uint32_t& n)
{
  uint32_t k, extra, k2, extra2;
  const uint32_t e = t.ll[length_symbol(mlen, k, extra)];
  const uint32_t e2 = t.d[distance_symbol(dist, k2, extra2)];
  const uint32_t l1 = e >> 16, l2 = e2 >> 16;
  uint64_t bits = e & 0xffffu;
  bits |= (uint64_t)extra << l1;
  bits |= (uint64_t)(e2 & 0xffffu) << (l1 + k);
  bits |= (uint64_t)extra2 << (l1 + k + l2);
  n = l1 + k + l2 + k2;
  return bits;
}

struct DynEmitter
{
  static constexpr bool kStream = true;
  static constexpr uint32_t kReach = 32768;

  static __device__ __forceinline__ void one(DynSink& s, const uint8_t* lit, uint32_t lit_len, uint32_t offset, uint32_t match_len)
  {
    const uint32_t lane = (uint32_t)wave::lane_id();
    BitSink& o = s.bits;
    for (uint32_t base = 0; base < lit_len; base += 64) {
      uint32_t n = 0, code = 0;
      if (base + lane < lit_len) {
        const uint32_t e = s.codes.ll[lit[base + lane]];
        code = e & 0xffffu;
        n = e >> 16;
      }
      const uint32_t incl = wave::scan_add_inclusive(n);
      if (n != 0) {
        put(o, o.bits + incl - n, code, n);
      }
      o.bits += wave::read_lane(incl, 63);
      flush(o);
    }
    while (match_len != 0) {
      const uint32_t piece = next_piece(match_len);
      uint32_t n;
      const uint64_t code = dynamic_pair(s.codes, piece, offset, n);
      if (lane == 0) {
        put64(o, o.bits, code);
      }
      o.bits += n;
      match_len -= piece;
      if (o.bits - o.base > 8 * (kStageBytes - 64)) {
        flush(o);
      }
    }
    flush(o);
  }

  static __device__ __forceinline__ void window(
      DynSink& s, const uint8_t* __restrict__ src, bool sel, uint32_t lit_from, uint32_t lit_len, uint32_t match_len,
      uint32_t offset, uint64_t before8, bool before_ok, uint32_t run)
  {
    const uint32_t lane = (uint32_t)wave::lane_id();
    BitSink& o = s.bits;
    const bool alone = sel && match_len <= kMaxMatch && lit_len <= kDynLaneLits && (before_ok || lit_len == 0);
    const uint64_t hard = wave::ballot(sel && !alone);
    const uint32_t first_hard = hard ? wave::ctz64(hard) : 64u;
    const bool mine = alone && lane < first_hard;
    uint64_t lits = 0;
    if (mine && lit_len != 0) {
      lits = before8 >> (8 * (8 - run));
    }
    /* size: the literals' code lengths, one lookup each, and the pair */
    uint32_t lit_bits = 0;
    for (uint32_t i = 0; wave::ballot(mine && i < lit_len) != 0; ++i) {
      if (mine && i < lit_len) {
        lit_bits += s.codes.ll[(uint32_t)(lits >> (8 * i)) & 0xffu] >> 16;
      }
    }
    uint32_t pair_bits = 0;
    uint64_t pair_code = 0;
    if (mine) {
      pair_code = dynamic_pair(s.codes, match_len, offset, pair_bits);
    }
    const uint32_t size = mine ? lit_bits + pair_bits : 0u;
    const uint32_t incl = wave::scan_add_inclusive(size);
    uint32_t p = o.bits + incl - size;
    /* literals, two code words (at most 30 bits) per LDS update */
    for (uint32_t i = 0; wave::ballot(mine && i < lit_len) != 0; i += 2) {
      if (mine && i < lit_len) {
        const uint32_t e0 = s.codes.ll[(uint32_t)(lits >> (8 * i)) & 0xffu];
        uint32_t word = e0 & 0xffffu, filled = e0 >> 16;
        if (i + 1 < lit_len) {
          const uint32_t e1 = s.codes.ll[(uint32_t)(lits >> (8 * (i + 1))) & 0xffu];
          word |= (e1 & 0xffffu) << filled;
          filled += e1 >> 16;
        }
        put(o, p, word, filled);
        p += filled;
      }
    }
    if (mine) {
      put64(o, p, pair_code);
    }
    o.bits += wave::read_lane(incl, 63);
    flush(o);
    uint64_t rest = first_hard < 64 ? wave::ballot(sel) & (~0ull << first_hard) : 0ull;
    while (rest) {
      const uint32_t j = wave::ctz64(rest);
      rest &= rest - 1;
      one(s, src + wave::read_lane(lit_from, j), wave::read_lane(lit_len, j), wave::read_lane(offset, j),
          wave::read_lane(match_len, j));
    }
  }
};

/* LDS of one wave: the fixed-code compressor's + the two code tables (the construction scratch lies in the hash table) */
constexpr uint32_t kDynLdsPerWave = kEncLdsPerWave + ((kCodeLds + 15) & ~15u);
static_assert(kBuildLds + (kNumCL + kNumLL + kNumD + 32) * 4 <= 2 * lzm::kTableU16, "the construction borrows the hash table's LDS");

/* Compress src[0, n) into dst (capacity >= max_compressed_size(n)) with per-chunk codes. Returns the size. */
__device__ __forceinline__ uint32_t encode_chunk_dynamic(const uint8_t* __restrict__ src, uint32_t n, uint8_t* dst, uint8_t* lds)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  if (n < 1024) { /* the header alone would outweigh what better codes save */
    return encode_chunk(src, n, dst, lds);
  }
  uint16_t* table = (uint16_t*)lds;
  uint8_t* image = lds + 2 * lzm::kTableU16;
  uint8_t* stage = image + lzm::kStageBytes;
  CodeTables t;
  t.ll = (uint32_t*)(lds + kEncLdsPerWave);
  t.d = t.ll + kNumLL;
  /* 1. count (every symbol starts at one; the end-of-block symbol is used once) */
  for (uint32_t i = lane; i < kNumLL + kNumD; i += 64) {
    t.ll[i] = 1;
  }
  wave::sync();
  (void)lzm::encode_chunk<CountEmitter, 1, CodeTables>(src, n, dst, table, image, n - 4, n, true, &t);
  wave::sync();
  /* 2. build: scratch, then the code lengths of both alphabets in one array for the header */
  uint8_t* scratch = lds; /* the hash table is cleared again by the second run */
  const BuildScratch b = carve_scratch(scratch);
  uint32_t* cl_counts = (uint32_t*)(scratch + kBuildLds);           /* [19] */
  uint32_t* cl_codes = cl_counts + kNumCL;                          /* [19] */
  uint8_t* all_lens = (uint8_t*)(cl_codes + kNumCL + 1);            /* [316] */
  build_code_words(t.ll, kNumLL, kCodeMaxBits, b, t.ll);
  for (uint32_t i = lane; i < kNumLL; i += 64) {
    all_lens[i] = b.lens[i];
  }
  wave::sync();
  build_code_words(t.d, kNumD, kCodeMaxBits, b, t.d);
  for (uint32_t i = lane; i < kNumD; i += 64) {
    all_lens[kNumLL + i] = b.lens[i];
  }
  for (uint32_t i = lane; i < kNumCL; i += 64) {
    cl_counts[i] = 1;
  }
  wave::sync();
  for (uint32_t i = lane; i < kNumLL + kNumD; i += 64) {
    atomicAdd(cl_counts + all_lens[i], 1u);
  }
  wave::sync();
  build_code_words(cl_counts, kNumCL, 7, b, cl_codes);
  /* 3. header: BFINAL = 1, BTYPE = 10, HLIT = 29, HDIST = 29, HCLEN = 15; the 19 code-length code lengths in the
   *    order of the format; the 316 code lengths, one code-length symbol each */
  DynSink s;
  s.codes = t;
  /* the staging area must not be the scratch: it lies behind the table, and is cleared here */
  sink_init(s.bits, dst, stage);
  BitSink& o = s.bits;
  if (lane == 0) {
    put(o, 0, 1u | (2u << 1) | (29u << 3) | (29u << 8) | (15u << 13), 17);
  }
  o.bits = 17;
  {
    constexpr uint8_t kOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint32_t mine = 0;
    if (lane < 19) {
      uint32_t sym = 0;
#pragma unroll
      for (uint32_t i = 0; i < 19; ++i) {
        sym = i == lane ? kOrder[i] : sym;
      }
      mine = cl_codes[sym] >> 16;
      put(o, o.bits + 3 * lane, mine, 3);
    }
    o.bits += 3 * 19;
    flush(o);
  }
  for (uint32_t base = 0; base < kNumLL + kNumD; base += 64) {
    uint32_t nb = 0, code = 0;
    if (base + lane < kNumLL + kNumD) {
      const uint32_t e = cl_codes[all_lens[base + lane]];
      code = e & 0xffffu;
      nb = e >> 16;
    }
    const uint32_t incl = wave::scan_add_inclusive(nb);
    if (nb != 0) {
      put(o, o.bits + incl - nb, code, nb);
    }
    o.bits += wave::read_lane(incl, 63);
    flush(o);
  }
  /* the second run clears and uses the hash table: nothing of the scratch is needed any more */
  wave::sync();
  (void)lzm::encode_chunk<DynEmitter, 1, DynSink>(src, n, dst, table, image, n - 4, n, true, &s);
  {
    const uint32_t e = t.ll[256]; /* end of block */
    if (lane == 0) {
      put(o, o.bits, e & 0xffffu, e >> 16);
    }
    o.bits += e >> 16;
  }
  const uint32_t size = finish(o);
  if (size > stored_size(n)) {
    wave::sync();
    return encode_stored(src, n, dst);
  }
  return size;
}

} // namespace deflate
