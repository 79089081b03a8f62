/*
 * snappy/snappy_decode_window.hip.h -- Snappy raw-format decoder on the
 * LDS-staged executor (common/lz_window.hip.h); the default
 * nvcompBatchedSnappyDecompressAsync path. Element chase as in
 * lz4_decode_window.hip.h: 256 speculative tag positions per reload, one
 * v_readlane per element; literals with a multi-byte length field are resolved
 * in the speculative pass as well (the length bytes sit right behind the tag).
 */
#pragma once

#include "common/lz_window.hip.h"

namespace snappyw {

/* Delta stored for a position whose next token cannot be derived in the parallel
 * pass (chunk sizes are < 2^28, so position + kUnknown never looks like a position). */
constexpr uint32_t kUnknown = 1u << 28;

/* A token of the chase is one copy element, or a literal element together with the copy element that
 * follows it (when the literal element is at most kFuseMax bytes long and a copy does follow): the
 * executor takes "literal run, then match" per lane, so fusing the two halves the sequences of the common
 * literal-copy alternation. Chase, scalar fallback and parse apply the same rule. */
constexpr uint32_t kFuseMax = 256;

__device__ __forceinline__ uint32_t copy_size(uint32_t kind) /* kind 1..3 */
{
  return kind == 3 ? 5u : kind + 1;
}


/* Distance from a (speculative) tag at virtual position p to the next tag; kUnknown = unknown.
 * Branch-free: the tag and the four bytes behind it (a literal's length field) are fetched
 * together; the ring's 16-byte mirror covers a dword that starts before the ring's end. */
template <class R>
__device__ __forceinline__ uint32_t tag_delta(const R& r, uint32_t p)
{
  const uint8_t* ring = r.ring;
  const uint32_t m = R::kMask;
  const uint32_t t = ring[p & m];
  /* the 4 bytes behind the tag from two aligned dwords (a misaligned ds_read_b32 costs 16x) */
  const uint32_t fa = (p + 1) & ~3u;
  const uint32_t w = wave::align_bytes(*(const uint32_t*)(ring + ((fa + 4) & m)), *(const uint32_t*)(ring + (fa & m)), (p + 1) & 3u);
  const uint32_t kind = t & 3u;
  const uint32_t code = t >> 2;
  const uint32_t nb = code >= 60 ? code - 59 : 0u; /* bytes of an explicit literal length */
  const uint32_t ext = nb == 4 ? w : (w & ((1u << (8 * nb)) - 1u));
  const uint32_t lit_len = (nb ? ext : code); /* length - 1 */
  const uint32_t lit_delta = 1 + nb + lit_len + 1;
  const uint32_t copy_delta = kind == 3 ? 5u : kind + 1;
  bool unknown = p < r.lo || p + 5 > r.hi || p >= r.vend || (kind == 0 && nb && ext >= 0x7fffff00u);
  uint32_t delta = kind == 0 ? lit_delta : copy_delta;
  if (!unknown && kind == 0 && lit_delta <= kFuseMax) {
    const uint32_t p2 = p + lit_delta;
    if (p2 < r.vend) {
      if (p2 >= r.hi) {
        unknown = true;
      } else {
        const uint32_t k2 = ring[p2 & m] & 3u;
        delta += k2 ? copy_size(k2) : 0u;
      }
    }
  }
  return unknown ? kUnknown : delta;
}


/* Scalar fallback (tag not resolvable from the ring). */
template <class R>
__device__ __forceinline__ uint32_t chase_slow_next(const R& r, uint32_t q)
{
  const uint32_t vend = r.vend;
  const uint32_t t = lzw::in_byte_uniform(r, q);
  const uint32_t kind = t & 3u;
  if (kind != 0) {
    return q + copy_size(kind);
  }
  uint32_t len = t >> 2;
  uint32_t pos = q + 1;
  if (len >= 60) {
    const uint32_t nb = len - 59;
    if (vend - pos < nb) {
      return vend + 1;
    }
    len = 0;
    for (uint32_t i = 0; i < nb; ++i) {
      len |= lzw::in_byte_uniform(r, pos + i) << (8 * i);
    }
    pos += nb;
  }
  if (len >= vend - pos) {
    return vend + 1;
  }
  const uint32_t p2 = pos + len + 1;
  if (p2 - q <= kFuseMax && p2 < vend) {
    const uint32_t k2 = lzw::in_byte_uniform(r, p2) & 3u;
    return p2 + (k2 ? copy_size(k2) : 0u);
  }
  return p2;
}


struct DeltaFn
{
  static constexpr uint32_t kReach = 8 + kFuseMax + 8; /* tag + length field, and the tag behind a fusable literal */
  template <class R>
  __device__ __forceinline__ uint32_t operator()(const R& r, uint32_t p) const { return tag_delta(r, p); }
  /* interior window: `w` = the stream bytes from p on (tag in bits 0-7). A literal element with explicit length bytes
   * (more than 60 bytes: one literal element in a thousand on the mix) is left to the scalar walk: the speculative pass
   * runs for four positions per lane per window, and resolving the 1-4 length bytes there cost more than twice as many
   * instructions per position as the rest of the rule (35 vs 15). The size of a copy element by kind -- 0, 2, 3, 5 --
   * is a nibble table in a constant. */
  template <class R>
  __device__ __forceinline__ uint32_t fast(const R& r, uint32_t p, uint64_t w) const
  {
    const uint32_t t = (uint32_t)w & 0xffu;
    const uint32_t kind = t & 3u;
    const uint32_t code = t >> 2;
    const uint32_t lit_delta = code + 2; /* tag + (code + 1) bytes */
    const uint32_t k2 = r.ring[(p + lit_delta) & R::kMask] & 3u; /* harmless when not a literal */
    const uint32_t lit_total = code >= 60 ? kUnknown : lit_delta + ((0x5320u >> (4 * k2)) & 15u); /* kFuseMax >= 62: always fused */
    return kind == 0 ? lit_total : (0x5320u >> (4 * kind)) & 15u;
  }
  /* no second look at the positions fast() gave up on: one speculative position in 64 carries a literal tag with length
   * bytes, so nearly every window would take the branch */
  static constexpr bool kSecondChance = false;
  template <class R>
  __device__ __forceinline__ uint32_t second(const R&, uint32_t, uint64_t) const { return kUnknown; }
};
struct SlowFn
{
  template <class R>
  __device__ __forceinline__ uint32_t operator()(const R& r, uint32_t p) const { return chase_slow_next(r, p); }
};

template <class R>
__device__ __forceinline__ void parse(const R& r, uint32_t p, bool active, lz::Seq& s, bool& bad)
{
  s.lit_src = 0;
  s.lit_len = 0;
  s.match_off = 0;
  s.match_len = 0;
  bad = false;
  if (!active) {
    return;
  }
  const uint32_t vend = r.vend;
  uint32_t t = lzw::in_byte(r, p);
  uint32_t kind = t & 3u;
  uint32_t pos = p + 1;
  if (kind == 0) {
    uint32_t len = t >> 2;
    if (len >= 60) {
      const uint32_t nb = len - 59;
      if (vend - pos < nb) {
        bad = true;
        return;
      }
      len = 0;
      for (uint32_t i = 0; i < nb; ++i) {
        len |= lzw::in_byte(r, pos + i) << (8 * i);
      }
      pos += nb;
    }
    if (len >= vend - pos) {
      bad = true;
      return;
    }
    s.lit_src = pos;
    s.lit_len = len + 1;
    pos += len + 1;
    /* the copy element behind a short literal element belongs to the same token (kFuseMax rule) */
    if (pos - p > kFuseMax || pos >= vend) {
      return;
    }
    t = lzw::in_byte(r, pos);
    kind = t & 3u;
    if (kind == 0) {
      return;
    }
    ++pos;
  }
  const uint32_t need = kind == 3 ? 4u : kind;
  if (vend - pos < need) {
    bad = true;
    return;
  }
  if (kind == 1) {
    s.match_len = 4 + ((t >> 2) & 7u);
    s.match_off = ((t >> 5) << 8) | lzw::in_byte(r, pos);
  } else if (kind == 2) {
    s.match_len = 1 + (t >> 2);
    s.match_off = lzw::in_byte(r, pos) | (lzw::in_byte(r, pos + 1) << 8);
  } else {
    s.match_len = 1 + (t >> 2);
    s.match_off = lzw::in_byte(r, pos) | (lzw::in_byte(r, pos + 1) << 8) | (lzw::in_byte(r, pos + 2) << 16)
                  | (lzw::in_byte(r, pos + 3) << 24);
  }
  if (s.match_off == 0) {
    bad = true;
  }
}

/* The 8 stream bytes at virtual position p (resident, p + 11 below the end of residency): aligned dword reads and a
 * funnel shift (a misaligned ds_read_b32 is served lane by lane). */
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

/* parse() for the common batch, lanes [from, to) (to > from): literal elements of at most 60 bytes (no length bytes).
 * As in lz4_decode_window.hip.h, residency and the chunk's end are checked once for the wave from the first and the last
 * token position: a token of the fast kind spans at most kFastSpan stream bytes (tag, 60 literals, the fused copy's tag
 * and offset, and the tag behind it), so under the precondition every lane's fields are resident and inside the chunk
 * and the lanes run straight-line code. Returns false (wave-uniform) when the general parser must do the batch. */
constexpr uint32_t kFastSpan = 80;

template <class R>
__device__ __forceinline__ bool parse_fast(
    const R& r, uint32_t p, uint32_t from, uint32_t to, lz::Seq& s, bool& bad)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  const uint32_t first = wave::read_lane(p, from);
  const uint32_t last = wave::read_lane(p, to - 1);
  const uint32_t lim = r.hi < r.vend ? r.hi : r.vend;
  if (first < r.lo || last + kFastSpan > lim) {
    return false;
  }
  const bool active = lane - from < to - from;
  const uint64_t w = ring_bytes8(r, p); /* a lane outside [from, to) reads somewhere inside the ring: harmless */
  const uint32_t t = (uint32_t)w & 0xffu;
  const bool is_lit = (t & 3u) == 0;
  const uint32_t code = t >> 2;
  const uint32_t q = p + 2 + code; /* a literal element without length bytes ends here */
  /* the copy element: the token itself, or the tag behind the (always fusable: kFuseMax >= 62) literal */
  const uint64_t y = ring_bytes8(r, q);
  const uint64_t c = is_lit ? y : w;
  const uint32_t tag = (uint32_t)c & 0xffu;
  const uint32_t k = tag & 3u;
  if (wave::ballot(active && is_lit && code >= 60)) {
    return false;
  }
  const bool has_copy = k != 0; /* a literal element behind a literal element is a token of its own */
  const uint32_t b1 = (uint32_t)(c >> 8) & 0xffu;
  const uint32_t b12 = (uint32_t)(c >> 8) & 0xffffu;
  const uint32_t b1234 = (uint32_t)(c >> 8);
  const uint32_t off = k == 1 ? (((tag >> 5) << 8) | b1) : k == 2 ? b12 : b1234;
  const uint32_t mlen = k == 1 ? 4 + ((tag >> 2) & 7u) : 1 + (tag >> 2);
  s.lit_src = active && is_lit ? p + 1 : 0;
  s.lit_len = active && is_lit ? code + 1 : 0;
  s.match_off = active && has_copy ? off : 0;
  s.match_len = active && has_copy ? mlen : 0;
  bad = active && has_copy && off == 0;
  return true;
}

/* The varint32 preamble (uncompressed length) at the start of the stream, whose first block must be resident:
 * q = position of the first element. False: malformed. */
template <class R>
__device__ __forceinline__ bool read_preamble(const R& ir, uint32_t& q, uint32_t& total)
{
  q = ir.vbeg;
  total = 0;
  for (uint32_t shift = 0; shift <= 28 && q < ir.vend; shift += 7) {
    const uint32_t b = lzw::in_byte_uniform(ir, q);
    ++q;
    total |= (b & 127u) << shift;
    if (!(b & 128u)) {
      return !(shift == 28 && b > 15);
    }
  }
  return false;
}

/* A copy that continues the copy before it (same offset, nothing in between) is the same match going on: the
 * compressor cuts matches into 64-byte elements, so runs and periodic columns arrive as long trains of them.
 * The first lane of a train takes the whole length, the others become empty sequences. Returns the mask of the
 * continuing lanes. */
__device__ __forceinline__ uint64_t merge_trains(lz::Seq& s, uint32_t count)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  /* lanes >= count hold empty sequences, lane 0's neighbour reads as one: no lane tests */
  const uint32_t prev_off = wave::prev_lane(s.match_off);
  const uint32_t prev_len = wave::prev_lane(s.match_len);
  const bool cont = s.lit_len == 0 && s.match_len != 0 && prev_len != 0 && prev_off == s.match_off;
  const uint64_t train = wave::ballot(cont);
  if (train) {
    const uint32_t incl = wave::scan_add_inclusive(lane < count ? s.match_len : 0u);
    const uint64_t above = lane < 63 ? train >> (lane + 1) : 0ull;
    const uint32_t followers = wave::ctz64(~above); /* consecutive continuing lanes right after this one */
    const uint32_t end_incl = wave::shuffle(incl, (lane + followers) & 63u);
    if (cont) {
      s.match_len = 0;
      s.match_off = 0;
    } else if (followers) {
      s.match_len += end_incl - incl;
    }
  }
  return train;
}

/* What the workgroup-per-chunk decoder (common/lz_team.hip.h) needs to know of the format. */
struct TeamFrontEnd
{
  static constexpr uint32_t kPositions = 128; /* a token is at least 2 bytes (a copy element with a one-byte offset) */
  static constexpr bool kEmptyIsError = true; /* no preamble */
  static constexpr uint32_t kFewLongMatchesRatio = 12; /* common/lz_team.hip.h: such chunks go to the two-wave decoder */
  using Delta = DeltaFn;
  using Slow = SlowFn;
  /* What the stream says it decodes to (its varint32 preamble), read straight from memory by every lane alike; ~0u when
   * there is no complete preamble. The team decoder keeps a chunk on chip when THIS fits its buffer, whatever capacity the
   * caller passed (a rounded-up max_uncompressed_chunk_bytes, say: ADVICE r4). */
  static __device__ __forceinline__ uint32_t declared_length(const uint8_t* in, uint32_t in_len)
  {
    uint32_t v = 0;
    for (uint32_t i = 0; i < 5 && i < in_len; ++i) {
      const uint32_t b = in[i];
      v |= (b & 127u) << (7 * i);
      if (!(b & 128u)) {
        return (i == 4 && b > 15u) ? ~0u : v;
      }
    }
    return ~0u;
  }
  /* the preamble: first element, and the length the elements must produce exactly */
  template <class R>
  static __device__ __forceinline__ bool begin(const R& r, uint32_t out_cap, uint32_t& q, uint32_t& limit, uint32_t& err)
  {
    uint32_t total;
    if (!read_preamble(r, q, total)) {
      err |= lz::kErrInput;
      return false;
    }
    if (total > out_cap) {
      err |= lz::kErrOutput;
      return false;
    }
    limit = total;
    return true;
  }
  template <class R>
  static __device__ __forceinline__ void parse_batch(const R& r, uint32_t p, uint32_t from, uint32_t to, lz::Seq& s, bool& bad)
  {
    if (to <= from || !parse_fast(r, p, from, to, s, bad)) {
      const uint32_t lane = (uint32_t)wave::lane_id();
      parse(r, p, lane - from < to - from, s, bad);
    }
    (void)merge_trains(s, to);
  }
  static __device__ __forceinline__ bool finish_ok(uint32_t op, uint32_t q, uint32_t vend, uint32_t limit)
  {
    return op == limit && q == vend;
  }
};

/* RUNS: lz4_decode_window.hip.h: decode_chunk -- the instance of the loop that tries the run executor, for the chunks that
 * shrank 8 x (snappy_api.hip: decode_one). A refill's 64 elements are only ~9 runs here, 2-4 KB, which does not earn the
 * executor's fixed cost back (int32 column 1 308 -> 1 210 GB/s with every batch tried as it came, gpurun r6t): this loop
 * squeezes the empty followers of the merged trains out and tops the sequences in hand up first (below). */
template <bool CHECKED, bool RUNS = false>
__device__ __forceinline__ uint32_t decode_chunk(
    const uint8_t* __restrict__ in, uint32_t in_len, uint8_t* out, uint32_t out_cap, uint8_t* lds, uint32_t& err)
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
  lzw::in_ensure(ir, ir.vbeg, ir.vbeg + lzw::kInBlock);
  uint32_t q, total;
  if (!read_preamble(ir, q, total)) {
    err = lz::kErrInput;
    return 0;
  }
  if (CHECKED && total > out_cap) {
    err = lz::kErrOutput;
    return 0;
  }
  const uint32_t limit = CHECKED ? total : out_cap;
  lzw::Chase c;
  lzw::chase_init(c, q, lds + lzw::kOutLds + lzw::kInLds);
  uint32_t op = 0;
  uint32_t seqpos = 0;
  uint32_t count = 0;
  /* parsed elements stay in registers until they are executed; chase and parse run only when few are left
   * (lz4_decode_window.hip.h: decode_chunk) */
  constexpr uint32_t kRefillBelow = 24;
  lzw::RunGate gate = lzw::kRunGateInit;
  lz::Seq s;
  s.lit_src = 0;
  s.lit_len = 0;
  s.match_off = 0;
  s.match_len = 0;
  for (;;) {
    if (count == 0 && c.q >= ir.vend) {
      break;
    }
    if (count < kRefillBelow && c.q < ir.vend) {
      const uint32_t oldest = count ? wave::read_lane(seqpos, 0) : c.q;
      LZW_T(10);
      lzw::in_ensure(ir, oldest, (c.q & ~(lzw::kInBlock - 1)) + 3 * lzw::kInBlock);
      LZW_T(0);
      const uint32_t before = count;
      count = lzw::chase_tokens(c, ir, seqpos, count, DeltaFn(), SlowFn());
      lz::Seq fresh;
      bool bad;
      LZ_STAT("sn_refills", 1);
      LZ_STAT("sn_new_tokens", count - before);
      if (count <= before || !parse_fast(ir, seqpos, before, count, fresh, bad)) {
        LZ_STAT("sn_parse_general", 1);
        parse(ir, seqpos, lane >= before && lane < count, fresh, bad);
      }
      if (lane >= before) {
        s = fresh;
      }
      if (wave::ballot(bad)) {
        err |= lz::kErrInput;
        return 0;
      }
      LZW_T(3);
    }
    uint64_t train = merge_trains(s, count);
    if (RUNS && lzw::run_gate_open(gate)) {
      /* The run executor takes 60 sequences a batch, and a refill's 64 elements of a typed column are ~9 runs -- a literal
       * element and a train of 64-byte copies each, the followers of a train EMPTY sequences behind its head: the empty
       * ones are squeezed out and more elements taken in, until 48 sequences are in hand (or the stream, or the ring, ends). */
      for (uint32_t topup = 0; topup < 12; ++topup) {
        const bool live = lane < count && (s.lit_len | s.match_len) != 0;
        const uint64_t lm = wave::ballot(live);
        const uint32_t n_live = wave::popc64(lm);
        if (n_live != count) {
          const uint32_t below = wave::prefix_popc(lm);
          const uint32_t to = live ? below : n_live + (lane - below); /* a permutation: the others, empty, behind the live ones */
          seqpos = wave::permute_to(live ? seqpos : 0u, to);
          s.lit_src = wave::permute_to(live ? s.lit_src : 0u, to);
          s.lit_len = wave::permute_to(live ? s.lit_len : 0u, to);
          s.match_off = wave::permute_to(live ? s.match_off : 0u, to);
          s.match_len = wave::permute_to(live ? s.match_len : 0u, to);
          count = n_live;
          train = 0;
          LZ_STAT("sn_compactions", 1);
        }
        if (count >= 48 || c.q >= ir.vend) {
          break;
        }
        const uint32_t oldest = count ? wave::read_lane(seqpos, 0) : c.q;
        if (c.q + lzw::kChaseWin + DeltaFn::kReach - (oldest & ~(lzw::kInBlock - 1)) > lzw::kInRing) {
          break; /* the elements in hand hold the ring: the next chase window would reach past what can be resident */
        }
        lzw::in_ensure(ir, oldest, (c.q & ~(lzw::kInBlock - 1)) + 3 * lzw::kInBlock);
        const uint32_t before = count;
        count = lzw::chase_tokens(c, ir, seqpos, count, DeltaFn(), SlowFn());
        if (count <= before) {
          break;
        }
        lz::Seq fresh;
        bool bad;
        if (!parse_fast(ir, seqpos, before, count, fresh, bad)) {
          parse(ir, seqpos, lane >= before && lane < count, fresh, bad);
        }
        if (lane >= before) {
          s = fresh;
        }
        if (wave::ballot(bad)) {
          err |= lz::kErrInput;
          return 0;
        }
        train = merge_trains(s, count);
      }
    }
    LZW_T(15); /* copy trains merged */
    LZ_STAT("sn_rounds", 1);
    LZ_STAT("sn_rounds_with_train", train ? 1 : 0);
    bool big = false;
    /* runs (typed columns: a literal element, then copies of period 1 .. 16) straight to the output: lzw::execute_run_batch */
    static_assert(!NVCOMP_LZW_RUNS || lzw::kRunFits, "the run executor needs 2 176 bytes of window LDS");
    uint32_t take = 0;
    if (RUNS && lzw::run_gate_open(gate)) {
      bool misfit;
      take = lzw::execute_run_batch<CHECKED>(ir, ow, limit, op, count, s, misfit);
      gate = wave::uniform(lzw::run_gate_tried(gate, take, misfit));
    }
    if (take == 0) {
      take = lzw::execute_window_batch<CHECKED, false, lzw::NoHook, NVCOMP_LZW_LAZY_FLUSH && !RUNS>(ir, ow, limit, op, count, s, err, big);
      if (CHECKED && err) {
        return 0;
      }
      if (RUNS) {
        gate = wave::uniform(lzw::run_gate_window_took(gate, take, count));
      }
    }
    if (big) {
      /* the first sequence in hand has a long literal run or a long match, or is larger than a batch: straight to HBM */
      if (!lzw::stream_sequence<CHECKED>(ir, ow, limit, op, wave::read_lane(s.lit_src, 0), wave::read_lane(s.lit_len, 0),
                                         wave::read_lane(s.match_off, 0), wave::read_lane(s.match_len, 0), err)) {
        return 0;
      }
      take = 1 + wave::ctz64(~(train >> 1)); /* sequence 0 and the empty sequences of its train */
    }
    if (take < count) {
      const uint32_t from = (lane + take) & 63u;
      seqpos = wave::shuffle(seqpos, from);
      lzw::drop_front(s, take, count);
    }
    count -= take;
  }
  if (CHECKED && (op != total || c.q != ir.vend)) {
    err |= lz::kErrInput;
    return 0;
  }
  lzw::out_flush_all(ow, op);
  return op;
}

/* ---- two waves per chunk (small batches): lz4_decode_window.hip.h has the description ---- */
namespace pair {

using namespace lzw::pair;

template <bool CHECKED>
__device__ __forceinline__ void produce(const uint8_t* __restrict__ in, uint32_t in_len, uint8_t* lds)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  const Shared sh = shared_at(lds);
  lzw::InRing ir;
  lzw::in_init(ir, in, in_len, lds + lzw::kOutLds + lzw::kInLds);
  lzw::in_ensure(ir, ir.vbeg, ir.vbeg + lzw::kInBlock);
  uint32_t q, total;
  const bool preamble_ok = read_preamble(ir, q, total);
  lzw::Chase c;
  lzw::chase_init(c, q, lds + lzw::kOutLds + 2 * lzw::kInLds);
  uint32_t k = 0;
  for (;;) {
    const bool last = !preamble_ok || c.q >= ir.vend;
    uint32_t count = 0;
    lz::Seq s;
    s.lit_src = 0, s.lit_len = 0, s.match_off = 0, s.match_len = 0;
    bool bad = !preamble_ok || (last && CHECKED && c.q != ir.vend); /* the elements must end exactly at the end */
    if (!last) {
      lzw::in_ensure(ir, c.q, (c.q & ~(lzw::kInBlock - 1)) + 3 * lzw::kInBlock);
      uint32_t seqpos = 0;
      count = lzw::chase_tokens(c, ir, seqpos, 0, DeltaFn(), SlowFn());
      if (count == 0 || !parse_fast(ir, seqpos, 0, count, s, bad)) {
        parse(ir, seqpos, lane < count, s, bad);
      }
      (void)merge_trains(s, count);
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
      return;
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
  lzw::in_ensure(ir, ir.vbeg, ir.vbeg + lzw::kInBlock);
  uint32_t q, total;
  const bool preamble_ok = read_preamble(ir, q, total);
  if (!preamble_ok || (CHECKED && total > out_cap)) {
    err = preamble_ok ? lz::kErrOutput : lz::kErrInput;
    if (lane == 0) {
      wave::lds_store_release(sh.abort, 1u);
    }
    return 0;
  }
  const uint32_t limit = CHECKED ? total : out_cap;
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
    {
      /* the literal copies read this wave's own ring; copy elements carry no stream position */
      const bool has_lit = lane < count && s.lit_len != 0;
      if (wave::ballot(has_lit)) {
        const uint32_t hi = wave::reduce_max(has_lit ? s.lit_src : 0u);
        const uint32_t lo = ~wave::reduce_max(has_lit ? ~s.lit_src : 0u);
        lzw::in_ensure(ir, lo, (hi & ~(lzw::kInBlock - 1)) + 2 * lzw::kInBlock);
      }
    }
    bool big = false;
    uint32_t take = 0;
    if (RUNS && lzw::run_gate_open(gate)) {
      bool misfit;
      take = lzw::execute_run_batch<CHECKED>(ir, ow, limit, op, count, s, misfit);
      gate = wave::uniform(lzw::run_gate_tried(gate, take, misfit));
    }
    if (take == 0) {
      take = lzw::execute_window_batch<CHECKED>(ir, ow, limit, op, count, s, err, big);
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
      if (!lzw::stream_sequence<CHECKED>(ir, ow, limit, op, wave::read_lane(s.lit_src, 0), wave::read_lane(s.lit_len, 0),
                                         wave::read_lane(s.match_off, 0), wave::read_lane(s.match_len, 0), err)) {
        if (lane == 0) {
          wave::lds_store_release(sh.abort, 1u);
        }
        return 0;
      }
      /* sequence 0 and the empty sequences (the followers of its copy train) right behind it */
      const uint64_t empty = wave::ballot(lane < count && s.lit_len + s.match_len == 0);
      take = 1 + wave::ctz64(~(empty >> 1));
      take = take < count ? take : count;
    }
    if (take < count) {
      lzw::drop_front(s, take, count);
    }
    count -= take;
  }
  if (CHECKED && op != total) {
    err |= lz::kErrInput;
    return 0;
  }
  lzw::out_flush_all(ow, op);
  return op;
}

} // namespace pair

} // namespace snappyw
