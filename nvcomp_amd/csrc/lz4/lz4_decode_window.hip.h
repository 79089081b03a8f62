/*
 * lz4/lz4_decode_window.hip.h -- LZ4 block decoder on the LDS-staged executor
 * (common/lz_window.hip.h). Same entry-point contract as lz4_decode.hip.h; this
 * is the default nvcompBatchedLZ4DecompressAsync path.
 *
 * Token chase: the next 256 stream positions are examined in parallel -- lane l
 * of register j assumes a token starts at position wb + 64 j + l and computes
 * the distance to the token after it (reading the at most two length-extension
 * bytes it needs from the input ring). The real chain is then followed with one
 * v_readlane per sequence. Tokens with a literal run of 525 bytes or more or a match of
 * 1 549 or more (a third / a seventh length-extension byte) take a scalar slow path.
 */
#pragma once

#include "common/lz_window.hip.h"
#include "common/lz_index.hip.h"

namespace lz4w {

/* Delta stored for a position whose next token cannot be derived in the parallel
 * pass (chunk sizes are < 2^28, so position + kUnknown never looks like a position). */
constexpr uint32_t kUnknown = 1u << 28;


/* The 8 (4) stream bytes at virtual position p -- those of them that are resident, the others are whatever the ring
 * holds there: three (two) aligned dword reads (a misaligned ds_read_b32 is served lane by lane) and a funnel shift. */
template <class R>
__device__ __forceinline__ uint64_t ring_bytes8(const R& r, uint32_t p)
{
  const uint32_t m = R::kMask;
  const uint32_t a0 = p & ~3u;
  const uint32_t d0 = *(const uint32_t*)(r.ring + (a0 & m));
  const uint32_t d1 = *(const uint32_t*)(r.ring + ((a0 + 4) & m));
  const uint32_t d2 = *(const uint32_t*)(r.ring + ((a0 + 8) & m));
  const uint32_t lo = wave::align_bytes(d1, d0, p & 3u);
  const uint32_t hi = wave::align_bytes(d2, d1, p & 3u);
  return ((uint64_t)hi << 32) | lo;
}
template <class R>
__device__ __forceinline__ uint32_t ring_bytes4(const R& r, uint32_t p)
{
  const uint32_t m = R::kMask;
  const uint32_t a0 = p & ~3u;
  const uint32_t d0 = *(const uint32_t*)(r.ring + (a0 & m));
  const uint32_t d1 = *(const uint32_t*)(r.ring + ((a0 + 4) & m));
  return wave::align_bytes(d1, d0, p & 3u);
}

/* A length field behind a nibble of 15, its bytes in `f` from bit 0 on (six of them, zeros above): how many of them are
 * 255 (0 .. 6; 6 = the field goes on behind what f holds) */
__device__ __forceinline__ uint32_t leading_255(uint64_t f)
{
  return (uint32_t)__builtin_ctzll(~f) >> 3; /* bits 48-63 of ~f are set */
}

/* Distance from a (speculative) token at virtual position p to the next token, with every bound tested: the windows at
 * the edges of the chunk and of the resident stream, and the positions of an interior window whose lengths the
 * straight-line DeltaFn::fast() gave up on. Up to two extension bytes of the literal length (runs to 524 bytes) and SIX
 * of the match length (to 1 548 bytes): a sorted key column compressed by liblz4 -- the reference's published shape --
 * is all matches of 170 .. 680 bytes, and with one extension byte every one of its tokens left the chase through the
 * scalar slow path, one enumeration per token (profiles/archive/r03_pmc_mortgage.json: 155 scalar instructions per sequence).
 * Longer fields -> kUnknown -> chase_slow_next(). */
template <class R>
__device__ __forceinline__ uint32_t token_delta(const R& r, uint32_t p)
{
  const uint8_t* ring = r.ring;
  const uint32_t m = R::kMask;
  const uint32_t t = ring[p & m];
  const uint32_t e1 = ring[(p + 1) & m];
  const uint32_t e1b = ring[(p + 2) & m];
  const uint32_t lit_code = t >> 4;
  const bool lit_ext = lit_code == 15;
  const bool lit_ext2 = lit_ext && e1 == 255;
  const uint32_t lit = lit_code + (lit_ext ? e1 : 0u) + (lit_ext2 ? e1b : 0u);
  const uint32_t lit_end = p + 1 + (lit_ext ? 1u : 0u) + (lit_ext2 ? 1u : 0u) + lit;
  const bool ends = lit_end >= r.vend; /* literals reach the end of the chunk: the chase stops here */
  const uint32_t mpos = lit_end + 2;   /* where a match-length extension byte would sit */
  const bool m_ext = (t & 15u) == 15u;
  const uint32_t n255 = leading_255(ring_bytes8(r, mpos) & 0xffffffffffffull);
  const uint32_t mfield = m_ext ? n255 + 1 : 0u; /* bytes of the match length field: trusted below only when resident */
  const uint32_t delta = ends ? lit_end - p : mpos + mfield - p;
  const bool unknown = p < r.lo || p + 3 > r.hi || p >= r.vend || (lit_ext2 && e1b == 255)
                       || (!ends && m_ext && (n255 >= 6 || mpos + n255 >= r.hi));
  return unknown ? kUnknown : delta;
}


/* The rest of a length field whose bytes so far were all 255, from virtual position pos on: 64 bytes per step (a byte
 * per lane, one ballot for the first byte that is not 255). Adds the bytes to `sum` (saturating: a corrupt stream may claim
 * anything) and returns the position behind the field, or vend + 1 when the chunk ends first. Whole wave, uniform
 * arguments. A chunk that is ONE literal run or ONE match -- incompressible data, zeros -- spells its length in 257 such
 * bytes, and walking them one dependent LDS read at a time (twice: once for the chase, once for the parser) took a third
 * of such a chunk's time. */
template <class R>
__device__ __forceinline__ uint32_t scan_length_bytes(const R& r, uint32_t pos, uint32_t& sum)
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  for (;;) {
    const uint32_t p = pos + lane;
    const bool inside = p < r.vend;
    const uint32_t b = inside ? lzw::in_byte(r, p) : 0u;
    const uint64_t stop = wave::ballot(b != 255u || !inside);
    if (stop) {
      const uint32_t n = wave::ctz64(stop);
      if (pos + n >= r.vend) {
        return r.vend + 1;
      }
      const uint32_t add = 255u * n + wave::read_lane(b, n);
      sum = sum + add < 0x7fffff00u ? sum + add : 0x7fffff00u;
      return pos + n + 1;
    }
    sum = sum + 255u * 64u < 0x7fffff00u ? sum + 255u * 64u : 0x7fffff00u;
    pos += 64;
  }
}

/* Scalar walk over one token with multi-byte length extensions. */
template <class R>
__device__ __forceinline__ uint32_t chase_slow_next(const R& r, uint32_t q)
{
  const uint32_t vend = r.vend;
  const uint32_t t = lzw::in_byte_uniform(r, q);
  uint32_t pos = q + 1;
  uint32_t lit = t >> 4;
  if (lit == 15) {
    pos = scan_length_bytes(r, pos, lit);
    if (pos > vend) {
      return vend + 1;
    }
  }
  if (lit >= vend - pos) {
    return vend + 1;
  }
  pos += lit + 2;
  if ((t & 15u) == 15u) {
    uint32_t ignored = 0;
    pos = scan_length_bytes(r, pos, ignored);
    if (pos > vend) {
      return vend + 1;
    }
  }
  return pos;
}


struct DeltaFn
{
  /* a delta looks at most this far past its position: token, 2 length bytes, 15 + 254 literals, offset, 8 bytes of length */
  static constexpr uint32_t kReach = 288;
  template <class R>
  __device__ __forceinline__ uint32_t operator()(const R& r, uint32_t p) const { return token_delta(r, p); }
  /* interior window: `w` = the stream bytes from p on (token in bits 0-7, the byte behind it in 8-15) */
  template <class R>
  __device__ __forceinline__ uint32_t fast(const R& r, uint32_t p, uint64_t w) const
  {
    /* straight-line on purpose (no ||, no early exit): the byte a match-length extension would use is read whatever
     * the token says, so that the four positions of a lane compile to one instruction stream without exec-mask
     * juggling -- the scalar unit is as busy as the vector unit in this kernel */
    const uint32_t t = (uint32_t)w & 0xffu;
    const uint32_t e1 = (uint32_t)(w >> 8) & 0xffu;
    const uint32_t lit_code = t >> 4;
    const uint32_t lit_ext = lit_code == 15 ? 1u : 0u;
    const uint32_t d0 = 3 + lit_code + (lit_ext ? e1 + 1u : 0u); /* to the byte a match-length extension would use */
    const uint32_t m_ext = (t & 15u) == 15u ? 1u : 0u;
    const uint32_t e2 = r.ring[(p + d0) & R::kMask];
    const uint32_t unknown = (lit_ext & (e1 == 255 ? 1u : 0u)) | (m_ext & (e2 == 255 ? 1u : 0u));
    return unknown ? kUnknown : d0 + m_ext;
  }
  /* the positions fast() gave up on, once more (lzw::chase_build, under a branch of the wave): a match length with two
   * to six extension bytes -- 274 .. 1 548 bytes, every sequence of a sorted key column -- is resolved here; everything
   * longer stays with the scalar walk */
  static constexpr bool kSecondChance = true;
  template <class R>
  __device__ __forceinline__ uint32_t second(const R& r, uint32_t p, uint64_t w) const
  {
    const uint32_t t = (uint32_t)w & 0xffu;
    const uint32_t e1 = (uint32_t)(w >> 8) & 0xffu;
    const uint32_t lit_code = t >> 4;
    const bool lit_ext = lit_code == 15;
    const uint32_t d0 = 3 + lit_code + (lit_ext ? e1 + 1u : 0u);
    const uint32_t n255 = leading_255(ring_bytes8(r, p + d0) & 0xffffffffffffull);
    const bool ok = !(lit_ext && e1 == 255) && (t & 15u) == 15u && n255 < 6;
    return ok ? d0 + n255 + 1 : kUnknown;
  }
};
struct SlowFn
{
  template <class R>
  __device__ __forceinline__ uint32_t operator()(const R& r, uint32_t p) const { return chase_slow_next(r, p); }
};

/* Lane-parallel field decode of the sequence whose token is at virtual position p (the general parser: any length, any
 * residency, the chunk's last sequence). A lane reads the first byte of a length field itself; a field that goes on
 * behind a 255 is finished by the whole wave, one such lane after the other (scan_length_bytes). */
template <class R>
__device__ __forceinline__ void parse(const R& r, uint32_t p, bool active, lz::Seq& s, bool& bad)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  const uint32_t vend = r.vend;
  s.lit_src = 0;
  s.lit_len = 0;
  s.match_off = 0;
  s.match_len = 0;
  bad = false;
  uint32_t t = 0, pos = 0, lit = 0;
  bool more = false;
  if (active) {
    t = lzw::in_byte(r, p);
    pos = p + 1;
    lit = t >> 4;
    if (lit == 15) {
      if (pos >= vend) {
        bad = true;
      } else {
        const uint32_t b = lzw::in_byte(r, pos++);
        lit += b;
        more = b == 255;
      }
    }
  }
  for (uint64_t m = wave::ballot(more); m; m &= m - 1) {
    const uint32_t j = wave::ctz64(m);
    uint32_t sum = 0;
    const uint32_t np = scan_length_bytes(r, wave::read_lane(pos, j), sum);
    if (lane == j) {
      bad = np > vend;
      pos = np;
      lit += sum;
    }
  }
  bool has_match = false;
  uint32_t mlen = 0;
  more = false;
  if (active && !bad) {
    if (lit > vend - pos) {
      bad = true;
    } else {
      s.lit_src = pos;
      s.lit_len = lit;
      pos += lit;
      if (pos != vend) { /* pos == vend: the last sequence, literals only */
        if (vend - pos < 2) {
          bad = true;
        } else {
          s.match_off = lzw::in_byte(r, pos) | (lzw::in_byte(r, pos + 1) << 8);
          pos += 2;
          has_match = true;
          mlen = t & 15u;
          if (mlen == 15) {
            if (pos >= vend) {
              bad = true;
            } else {
              const uint32_t b = lzw::in_byte(r, pos++);
              mlen += b;
              more = b == 255;
            }
          }
        }
      }
    }
  }
  for (uint64_t m = wave::ballot(more); m; m &= m - 1) {
    const uint32_t j = wave::ctz64(m);
    uint32_t sum = 0;
    const uint32_t np = scan_length_bytes(r, wave::read_lane(pos, j), sum);
    if (lane == j) {
      bad = np > vend;
      pos = np;
      mlen += sum;
    }
  }
  if (has_match && !bad) {
    if (mlen > 0x40000000u || pos >= vend) { /* a token must follow every match */
      bad = true;
    } else {
      s.match_len = mlen + 4;
    }
  }
  if (bad) {
    s.lit_src = 0;
    s.lit_len = 0;
    s.match_off = 0;
    s.match_len = 0;
  }
}

/* The parser of a batch, lanes [from, to) (to >= from). A lane whose fields are resident and inside the chunk -- the
 * position of its offset tells: everything it looks at lies within 8 bytes of it -- runs straight-line code: one
 * extension byte of the literal length, up to six of the match length (to 1 548 bytes; 255 * (the number of leading 255s)
 * + the byte behind them). The other lanes -- the last sequences of a chunk, a literal run of 270 bytes or more, a
 * longer match -- go through parse() together afterwards. (Until the middle of round 3 ONE such lane sent the whole
 * batch to parse(), which finishes a field with a second extension byte one lane after the other: every batch of a
 * sorted key column -- matches of 170 .. 680 bytes -- and the last 40 sequences of every such chunk.) */
template <class R>
__device__ __forceinline__ void parse_batch(const R& r, uint32_t p, uint32_t from, uint32_t to, lz::Seq& s, bool& bad)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  const bool active = lane - from < to - from;
  const uint32_t lim = r.hi < r.vend ? r.hi : r.vend;
  const uint64_t w = ring_bytes8(r, p); /* a lane outside [from, to) reads somewhere inside the ring: harmless */
  const uint32_t t = (uint32_t)w & 0xffu;
  const uint32_t e1 = (uint32_t)(w >> 8) & 0xffu;
  const uint32_t code = t >> 4;
  const bool lit_ext = code == 15;
  const uint32_t lit = code + (lit_ext ? e1 : 0u);
  const uint32_t lit_src = p + 1 + (lit_ext ? 1u : 0u);
  const uint32_t q = lit_src + lit; /* the offset */
  const uint64_t x = ring_bytes8(r, q);
  const uint32_t mcode = t & 15u;
  const uint64_t f = x >> 16; /* the match length field, if there is one */
  const uint32_t n255 = leading_255(f);
  const uint32_t mext = 255u * n255 + ((uint32_t)(f >> (8 * n255)) & 0xffu);
  /* not for this path: a second extension byte of the literal length (code 15 followed by 255 = bits 4-15 of w all
   * set), a seventh of the match length, fields that are not resident or reach the end of the chunk (q + 8 < vend also
   * says that a token follows the match) */
  const bool lit2 = ((uint32_t)w & 0xfff0u) == 0xfff0u;
  const bool straight = !lit2 && n255 < 6 && p >= r.lo && q + 8 < lim;
  const bool mine = active && straight;
  s.lit_src = mine ? lit_src : 0;
  s.lit_len = mine ? lit : 0;
  s.match_off = mine ? ((uint32_t)x & 0xffffu) : 0;
  s.match_len = mine ? mcode + 4 + (mcode == 15 ? mext : 0u) : 0;
  bad = false;
  const uint64_t rest = wave::ballot(active && !straight);
  if (rest) {
    lz::Seq g;
    bool gbad;
    parse(r, p, wave::lane_in(rest), g, gbad);
    if (wave::lane_in(rest)) {
      s = g;
      bad = gbad;
    }
  }
}

/* What the token index (common/lz_index.hip.h) needs to know of the format: the size of ONE sequence in the stream, from
 * the 64 bytes a lane holds in its LDS block. Straight-line like DeltaFn::fast: one extension byte of the literal length,
 * up to two of the match length (matches to 528 bytes); anything longer is "cannot tell" -- the index ends there and the
 * chase above takes over. The bytes behind the literals are read whatever the token says (a read behind the block lands in
 * the wave's own LDS and is not used). */
struct IndexFormat
{
  static __device__ __forceinline__ uint32_t first_token(const uint8_t*, uint32_t) { return 0; }
  /* the token at block offset o (o + 1 < kBlock): true = its successor starts at next_o (which may lie behind the block:
   * the walk refills there); false = the block does not hold what it takes (ok: a refill at the token will; !ok: never) */
  static __device__ __forceinline__ bool step(const uint8_t* blk, uint32_t o, uint32_t& next_o, bool& ok)
  {
    const uint32_t t = blk[o], e1 = blk[o + 1];
    const uint32_t code = t >> 4;
    const bool c15 = code == 15;
    const uint32_t d0 = 3 + code + (c15 ? e1 + 1u : 0u); /* to the byte a match-length extension would use */
    const uint32_t o2 = o + d0;
    const uint32_t e2 = blk[o2], e3 = blk[o2 + 1];
    const bool m15 = (t & 15u) == 15u;
    const bool fits = !m15 | (o2 + 1 < lzx::kBlock); /* only a long match looks at the bytes behind the offset */
    const bool m2 = m15 & fits & (e2 == 255);
    next_o = o2 + (m15 ? (m2 ? 2u : 1u) : 0u);
    ok = !((c15 & (e1 == 255)) | (m2 & (e3 == 255)));
    return ok & fits;
  }
};

/* What the workgroup-per-chunk decoder (common/lz_team.hip.h) needs to know of the format. */
struct TeamFrontEnd
{
  static constexpr uint32_t kPositions = 192; /* a sequence is at least 3 bytes (token + offset): 64 tokens at most */
  static constexpr bool kEmptyIsError = false; /* an empty block decodes to nothing */
  static constexpr uint32_t kFewLongMatchesRatio = 16; /* common/lz_team.hip.h: such chunks go to the two-wave decoder */
  using Delta = DeltaFn;
  using Slow = SlowFn;
  /* an LZ4 block does not say what it decodes to: the caller's capacity decides (pass tight capacities for the team path) */
  static __device__ __forceinline__ uint32_t declared_length(const uint8_t*, uint32_t) { return ~0u; }
  template <class R>
  static __device__ __forceinline__ bool begin(const R& r, uint32_t out_cap, uint32_t& q, uint32_t& limit, uint32_t&)
  {
    q = r.vbeg;
    limit = out_cap;
    return true;
  }
  static __device__ __forceinline__ bool finish_ok(uint32_t, uint32_t, uint32_t, uint32_t) { return true; }
  template <class R>
  static __device__ __forceinline__ void parse_batch(const R& r, uint32_t p, uint32_t from, uint32_t to, lz::Seq& s, bool& bad)
  {
    lz4w::parse_batch(r, p, from, to, s, bad);
  }
};

/* Decode one chunk with the calling wave; `lds` is this wave's kLdsPerWave bytes. */
/* ABLATE (profiling builds only, results are wrong by construction): 1 = stop after the
 * token chase, 2 = after the parse, 0 = the real decoder. */
/* RUNS: the loop tries lzw::execute_run_batch (sorted keys, typed columns). Its mere presence in the loop costs every
 * other kind of data 1.5-2 % (block placement and register allocation of the batch executor around it: gpurun r6u, r6v --
 * the same with the attempts gated off at run time), so the kernels instantiate the loop twice and pick per chunk by what
 * the stream shrank to (decode_one: runs only pay from 8 x on). */
template <bool CHECKED, int ABLATE = 0, bool RUNS = false>
__device__ __forceinline__ uint32_t decode_chunk(
    const uint8_t* __restrict__ in, uint32_t in_len, uint8_t* out, uint32_t out_cap, uint8_t* lds, uint32_t& err,
    uint8_t* index_scratch = nullptr)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  err = lz::kErrNone;
  if (in_len == 0) {
    return 0;
  }
  /* where the sequences start: the token index (common/lz_index.hip.h) when the caller's temp buffer has room for it --
   * a prefix of the chunk's tokens; the chase below takes over where it ends (at once without an index). The index is
   * built in the LDS that the ring, the window and the jump tables use afterwards. */
  static_assert(lzx::kLdsBytes <= lzw::kLdsPerWave, "the index is built in the wave's own LDS");
  LZW_T(9);
  lzx::Index ix = lzx::build<IndexFormat>(in, in_len, lds, index_scratch);
  LZW_T(15); /* the token index */
#ifdef NVCOMP_LZX_BUILD_ONLY /* profiling builds only: the index is built and thrown away (what the walk costs in place) */
  ix.lanes = 0, ix.ahead_n = 0, ix.resume = 0;
#endif
  lzw::InRing ir;
  lzw::OutWindow ow;
  lzw::in_init(ir, in, in_len, lds + lzw::kOutLds);
  lzw::out_init(ow, out, lds);
  lzw::Chase c;
  lzw::chase_init(c, ir.vbeg + ix.resume, lds + lzw::kOutLds + lzw::kInLds);
  uint32_t op = 0;
  uint32_t seqpos = 0;
  uint32_t count = 0; /* sequences recorded in seqpos lanes [0, count) and not yet executed */
  /* A batch ends at 1 KiB of output, so on data with long matches (runs, sorted key columns: 200-400 bytes per
   * sequence) a round executes only a handful of the 64 sequences a chase delivers: chasing again for the rest every
   * round was 5 000 cycles per SEQUENCE on the reference's own published shape (profiles/archive/r02_mortgage_like.json).
   * The chase runs only when fewer than kRefillBelow token positions are left; on text a round takes all 64 and
   * every round refills, as before. NVCOMP_LZ4W_KEEP_PARSED = 1 also keeps the parsed FIELDS in registers across
   * rounds (four more live registers); 0 parses the positions in hand again every round (two LDS round trips). */
#ifndef NVCOMP_LZ4W_KEEP_PARSED
#define NVCOMP_LZ4W_KEEP_PARSED 1 /* measured: headline 498 vs 500 GB/s (noise), mortgage-like column 545 vs 442 */
#endif
  constexpr uint32_t kRefillBelow = 24;
  lzw::RunGate gate = lzw::kRunGateInit;
  lz::Seq s;
  s.lit_src = 0;
  s.lit_len = 0;
  s.match_off = 0;
  s.match_len = 0;
  for (;;) {
    const bool indexed = lzx::more(ix);
    if (count == 0 && !indexed && c.q >= ir.vend) {
      break;
    }
    uint32_t before = NVCOMP_LZ4W_KEEP_PARSED ? count : 0u; /* lanes [before, count) are parsed this round */
    const bool refill = count < kRefillBelow && (indexed || c.q < ir.vend);
    if (refill && indexed) {
      /* the next token positions out of the index */
      LZW_T(10);
      count = lzx::read(ix, seqpos, count, ir.vbeg);
      if (count == 0) {
        continue; /* (the lists that were left held nothing: the chase takes over) */
      }
      const uint32_t oldest = wave::read_lane(seqpos, 0);
      lzw::in_ensure(ir, oldest, (oldest & ~(lzw::kInBlock - 1)) + 3 * lzw::kInBlock);
      LZW_T(0);
    } else if (refill) {
      /* keep the stream resident from the oldest unexecuted token to well past the chase */
      const uint32_t oldest = count ? wave::read_lane(seqpos, 0) : c.q;
      LZW_T(10);
      lzw::in_ensure(ir, oldest, (c.q & ~(lzw::kInBlock - 1)) + 3 * lzw::kInBlock);
      LZW_T(0);
      count = lzw::chase_tokens(c, ir, seqpos, count, DeltaFn(), SlowFn());
      if (ABLATE == 1) {
        op += wave::reduce_add(lane < count ? seqpos : 0u) & 1u;
        count = 0;
        continue;
      }
    }
    if (refill || !NVCOMP_LZ4W_KEEP_PARSED) {
      lz::Seq fresh;
      bool bad;
      parse_batch(ir, seqpos, before, count, fresh, bad);
      LZW_T(3);
      if (lane >= before) {
        s = fresh;
      }
      if (ABLATE == 2) {
        op += wave::reduce_add(s.lit_len + s.match_len + s.match_off) & 1u;
        count = 0;
        continue;
      }
      if (wave::ballot(bad)) {
        err |= lz::kErrInput;
        return 0;
      }
    }
    bool big = false;
    /* runs (sorted keys, typed columns) are executed 60 at a time, straight to the output: lzw::execute_run_batch */
    static_assert(!NVCOMP_LZW_RUNS || lzw::kRunFits, "the run executor needs 2 176 bytes of window LDS");
    uint32_t take = 0;
    if (RUNS && lzw::run_gate_open(gate)) {
      bool misfit;
      take = lzw::execute_run_batch<CHECKED>(ir, ow, out_cap, op, count, s, misfit);
      gate = wave::uniform(lzw::run_gate_tried(gate, take, misfit));
    }
    if (take == 0) {
      auto settle = [&ix]() { lzx::settle(ix); };
      take = lzw::execute_window_batch<CHECKED, false, decltype(settle), NVCOMP_LZW_LAZY_FLUSH && !RUNS>(ir, ow, out_cap, op, count, s, err, big, settle);
      if (CHECKED && err) {
        return 0;
      }
      if (RUNS) {
        gate = wave::uniform(lzw::run_gate_window_took(gate, take, count));
      }
    }
    if (big) {
      /* the first sequence in hand has a long literal run or a long match, or is larger than a batch: straight to HBM */
      if (!lzw::stream_sequence<CHECKED>(ir, ow, out_cap, op, wave::read_lane(s.lit_src, 0), wave::read_lane(s.lit_len, 0),
                                         wave::read_lane(s.match_off, 0), wave::read_lane(s.match_len, 0), err)) {
        return 0;
      }
      take = 1;
    }
    /* drop the executed sequences, keep the rest for the next round */
    if (take < count) {
      const uint32_t from = (lane + take) & 63u;
      seqpos = wave::shuffle(seqpos, from);
#if NVCOMP_LZ4W_KEEP_PARSED
      lzw::drop_front(s, take, count);
#endif
    }
    count -= take;
  }
  lzw::out_flush_all(ow, op);
  return op;
}

/* ---- two waves per chunk (small batches) -----------------------------------------------------------------------
 *
 * One wave per chunk leaves the card under-filled below ~7 000 chunks, and a 64 KiB chunk takes a wave ~0.75 ms
 * however idle the CU is: the wave's own dependent chain -- chase, parse, far loads, copy rounds, flush -- is what
 * takes the time (profiles/archive/r02_decode_phases.json). For small batches the chain is cut in two: wave 0 of a 128-thread
 * workgroup (the PRODUCER) runs the token chase and the parse and hands batches of parsed sequences to wave 1 (the
 * CONSUMER), which executes them; the two overlap, a chunk takes about as long as its slower half. Hand-over is a
 * two-slot queue in LDS with one flag word per slot (wave::lds_store_release / lds_load_acquire). The producer never
 * depends on anything the consumer does except a free slot; each wave keeps its own ring over the compressed stream
 * (the consumer's serves the literal copies), so nothing else is shared. Same bytes as decode_chunk.
 */
namespace pair {

using namespace lzw::pair;

__device__ __forceinline__ void produce(const uint8_t* __restrict__ in, uint32_t in_len, uint8_t* lds)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  const Shared sh = shared_at(lds);
  lzw::InRing ir;
  lzw::in_init(ir, in, in_len, lds + lzw::kOutLds + lzw::kInLds);
  lzw::Chase c;
  lzw::chase_init(c, ir.vbeg, lds + lzw::kOutLds + 2 * lzw::kInLds);
  uint32_t k = 0;
  for (;;) {
    const bool last = c.q >= ir.vend;
    uint32_t count = 0;
    lz::Seq s;
    s.lit_src = 0, s.lit_len = 0, s.match_off = 0, s.match_len = 0;
    bool bad = false;
    if (!last) {
      lzw::in_ensure(ir, c.q, (c.q & ~(lzw::kInBlock - 1)) + 3 * lzw::kInBlock);
      uint32_t seqpos = 0;
      count = lzw::chase_tokens(c, ir, seqpos, 0, DeltaFn(), SlowFn());
      parse_batch(ir, seqpos, 0, count, s, bad);
    }
    const uint32_t flags = (last ? kFlagLast : 0u) | (wave::ballot(bad) ? kFlagBad : 0u);
    while (poll(sh.state + k) != 0) {
      if (poll(sh.abort) != 0) {
        return;
      }
      wave::nap();
    }
    uint32_t* f = (uint32_t*)(sh.slot(k) + 16);
    f[lane] = s.lit_src;
    f[64 + lane] = s.lit_len;
    f[128 + lane] = s.match_off;
    f[192 + lane] = s.match_len;
    if (lane == 0) {
      ((uint32_t*)sh.slot(k))[0] = count;
      ((uint32_t*)sh.slot(k))[1] = flags;
    }
    wave::sync();
    if (lane == 0) {
      wave::lds_store_release(sh.state + k, 1u);
    }
    if (flags) {
      return; /* the end of the chunk, or a malformed token: nothing follows */
    }
    k ^= 1;
  }
}

template <bool CHECKED, bool RUNS = false>
__device__ __forceinline__ uint32_t consume(
    const uint8_t* __restrict__ in, uint32_t in_len, uint8_t* out, uint32_t out_cap, uint8_t* lds, uint32_t& err)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  const Shared sh = shared_at(lds);
  lzw::InRing ir;
  lzw::OutWindow ow;
  lzw::in_init(ir, in, in_len, lds + lzw::kOutLds);
  lzw::out_init(ow, out, lds);
  uint32_t op = 0;
  uint32_t count = 0;
  uint32_t k = 0;
  lzw::RunGate gate = lzw::kRunGateInit;
  lz::Seq s;
  s.lit_src = 0, s.lit_len = 0, s.match_off = 0, s.match_len = 0;
  for (;;) {
    if (count == 0) {
      while (poll(sh.state + k) != 1) {
        wave::nap();
      }
      const uint32_t* f = (const uint32_t*)(sh.slot(k) + 16);
      s.lit_src = f[lane];
      s.lit_len = f[64 + lane];
      s.match_off = f[128 + lane];
      s.match_len = f[192 + lane];
      const uint32_t n = wave::read_lane(((const uint32_t*)sh.slot(k))[0], 0);
      const uint32_t flags = wave::read_lane(((const uint32_t*)sh.slot(k))[1], 0);
      wave::sync();
      if (lane == 0) {
        wave::lds_store_release(sh.state + k, 0u);
      }
      k ^= 1;
      if (flags & kFlagBad) {
        err |= lz::kErrInput;
        return 0;
      }
      if (flags & kFlagLast) {
        break;
      }
      count = n;
      if (count == 0) {
        continue;
      }
    }
    /* the literal copies read this wave's own ring */
    {
      const uint32_t oldest = wave::read_lane(s.lit_src, 0);
      const uint32_t newest = wave::read_lane(s.lit_src, count - 1);
      lzw::in_ensure(ir, oldest, (newest & ~(lzw::kInBlock - 1)) + 2 * lzw::kInBlock);
    }
    bool big = false;
    uint32_t take = 0;
    if (RUNS && lzw::run_gate_open(gate)) {
      bool misfit;
      take = lzw::execute_run_batch<CHECKED>(ir, ow, out_cap, op, count, s, misfit);
      gate = wave::uniform(lzw::run_gate_tried(gate, take, misfit));
    }
    if (take == 0) {
      take = lzw::execute_window_batch<CHECKED>(ir, ow, out_cap, op, count, s, err, big);
      if (RUNS) {
        gate = wave::uniform(lzw::run_gate_window_took(gate, take, count));
      }
    }
    if (CHECKED && err) {
      if (lane == 0) {
        wave::lds_store_release(sh.abort, 1u);
      }
      return 0;
    }
    if (big) {
      /* the first sequence in hand has a long literal run or a long match, or is larger than a batch: straight to HBM */
      if (!lzw::stream_sequence<CHECKED>(ir, ow, out_cap, op, wave::read_lane(s.lit_src, 0), wave::read_lane(s.lit_len, 0),
                                         wave::read_lane(s.match_off, 0), wave::read_lane(s.match_len, 0), err)) {
        if (lane == 0) {
          wave::lds_store_release(sh.abort, 1u);
        }
        return 0;
      }
      take = 1;
    }
    if (take < count) {
      lzw::drop_front(s, take, count);
    }
    count -= take;
  }
  lzw::out_flush_all(ow, op);
  return op;
}

} // namespace pair

} // namespace lz4w
