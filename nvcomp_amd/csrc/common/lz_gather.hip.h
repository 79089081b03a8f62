/*
 * common/lz_gather.hip.h -- byte-gather batch executor of the LZ decoders (LZ4 and Snappy).
 *
 * The executor of round 1 (lzw::execute_window_batch) copied every literal run and every match with its own
 * lane: clamped dword moves, a 2/4/8-step ladder per copy kind, a separate aligned path for far data, rounds for
 * the matches that depend on each other -- about 290 vector instructions per batch of 64 sequences on a kernel
 * that is bound by vector-instruction issue (profiles/r01_final_pmc.json: VALU busy 70-80 %). Here the data
 * movement is turned around. A batch's output is a run of at most kCap bytes; LANE j OWNS THE 16 OUTPUT BYTES
 * j*16 .. j*16+15 of that run (a 16-byte aligned block of the output address) and GATHERS them:
 *
 *   - a sequence is two SEGMENTS of output bytes, its literals and its match; all bytes of a segment come from
 *     consecutive LDS addresses, i.e. source address = own window address + one constant `delta` per segment:
 *       literals        -> the compressed-stream ring,
 *       near match      -> the output window itself (delta = -offset),
 *       far match       -> a 32-byte staging slot per sequence, filled from HBM with two 16-byte loads,
 *       anything odd    -> copied into place by the whole wave first, then delta = 0 ("identity");
 *   - the sequence lanes publish their segments: one bit per segment START in a 1024-bit map (ds_or), the delta
 *     in a table indexed by the segment's rank (ballot + mbcnt);
 *   - a byte's segment is the number of start bits at or before it: one popcount per byte on the lane's 16 map
 *     bits plus an exclusive scan of the lanes' bit counts; then one table read for the delta and one byte read
 *     for the data -- no divergence, no per-length ladders, no alignment cases;
 *   - the 16 bytes are written to the window as one aligned ds_write_b128 and to HBM as one aligned
 *     global_store_dwordx4 straight from the registers.
 *
 * Matches whose source lies inside the batch's own output are not special: the gather is simply repeated (the
 * addresses stay in registers: 16 byte reads + the write) until a pass changes nothing; a pass is idempotent
 * for every byte whose source is final. When no dependent match reads another dependent match's output, ONE
 * repeat is known to suffice and the verifying pass is skipped. A self-overlapping match longer than
 * 4 x offset (runs) would need length / offset passes: it ends its batch and is expanded by the whole wave
 * (lzw::lds_match_copy), as is a match straddling the edge of the window.
 */
#pragma once

#include "common/lz_window.hip.h"

namespace lzg {

constexpr uint32_t kBytesPerLane = 16;
constexpr uint32_t kCap = 64 * kBytesPerLane; /* output bytes one batch covers, head bytes included */
constexpr uint32_t kFarSlot = 32;             /* bytes of far-match data staged per sequence */
constexpr uint32_t kMaxRank = 2 * 64 + 2;     /* identity + 128 segments + end marker */

/* Scratch in LDS (per wave, 16-byte aligned): far staging | start-bit map (one dword per lane) | delta table */
constexpr uint32_t kStageBytes = 64 * kFarSlot;
constexpr uint32_t kMapBytes = 64 * 4;
constexpr uint32_t kDeltaBytes = ((kMaxRank + 3) & ~3u) * 4;
constexpr uint32_t kScratch = kStageBytes + kMapBytes + kDeltaBytes;

static_assert(!NVCOMP_LZ_GATHER || lzw::kBatchMax == kCap, "the window is sized for batches of kCap bytes");

struct Scratch
{
  uint8_t* stage;
  uint32_t* map;
  int32_t* delta;
};

__device__ __forceinline__ Scratch scratch_at(uint8_t* lds)
{
  Scratch s;
  s.stage = lds;
  s.map = (uint32_t*)(lds + kStageBytes);
  s.delta = (int32_t*)(lds + kStageBytes + kMapBytes);
  return s;
}

/* Once per chunk: an empty map, rank 0 = identity. */
__device__ __forceinline__ void scratch_init(uint8_t* lds)
{
  const Scratch s = scratch_at(lds);
  s.map[wave::lane_id()] = 0;
  if (wave::lane_id() == 0) {
    s.delta[0] = 0;
  }
  wave::sync();
}

/* One gather pass over the lane's 16 bytes: p[b] already carries the segment's delta and the lane's block,
 * the byte's index inside the block is the instruction's immediate offset. */
__device__ __forceinline__ wave::u32x4 gather16(const uint8_t* const (&p)[16])
{
  uint32_t d[4];
#pragma unroll
  for (uint32_t q = 0; q < 4; ++q) {
    const uint32_t b0 = p[4 * q + 0][4 * q + 0];
    const uint32_t b1 = p[4 * q + 1][4 * q + 1];
    const uint32_t b2 = p[4 * q + 2][4 * q + 2];
    const uint32_t b3 = p[4 * q + 3][4 * q + 3];
    d[q] = (b0 | (b2 << 16)) | ((b1 | (b3 << 16)) << 8);
  }
  wave::u32x4 r = {d[0], d[1], d[2], d[3]};
  return r;
}

__device__ __forceinline__ bool differs(const wave::u32x4& a, const wave::u32x4& b)
{
  return ((a.x ^ b.x) | (a.y ^ b.y) | (a.z ^ b.z) | (a.w ^ b.w)) != 0;
}

/*
 * Execute the first sequences of a parsed batch. Same contract as lzw::execute_window_batch: lane k owns
 * sequence k (k < n), literal positions are virtual positions of the input ring; consumes as many leading
 * sequences as fit (at least one unless the first alone is larger than a batch: then `big` is set and nothing
 * is consumed), returns their number and adds the bytes produced to op.
 */
template <bool CHECKED>
__device__ __forceinline__ uint32_t execute_gather_batch(
    lzw::InRing& ir, lzw::OutWindow& ow, uint8_t* scratch_lds, uint32_t out_cap, uint32_t& op, uint32_t n,
    const lz::Seq& s, uint32_t& err, bool& big)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  const Scratch sc = scratch_at(scratch_lds);
  const bool active = lane < n;
  const uint32_t lit_len = active ? s.lit_len : 0;
  const uint32_t match_len = active ? s.match_len : 0;
  const uint32_t len = lit_len + match_len;
  const uint32_t incl = wave::scan_add_inclusive(len);
  big = false;

  /* the batch's blocks start at the 16-byte block of the output ADDRESS that op falls into: `head` bytes of
   * that block are older output (they are gathered from themselves) */
  const uint32_t head = (op + ow.align) & 15u;
  const uint32_t room = kCap - head;
  const uint64_t over = wave::ballot(active && incl > room) | (n < 64 ? (~0ull << n) : 0ull);
  uint32_t take = over ? wave::ctz64(over) : 64u;
  if (take == 0) {
    big = true;
    return 0;
  }

  LZW_T(4);
  lzw::out_make_room(ow, op);
  LZW_T(5);

  /* ---- where every sequence's bytes come from ---- */
  const uint32_t lit_dst = op + incl - len;
  const uint32_t match_dst = lit_dst + lit_len;
  const uint32_t match_src = match_dst - s.match_off;
  const bool has_match = match_len != 0;
  const bool near = has_match && match_src >= ow.valid_lo;
  const bool in_hbm = has_match && !near && match_src + match_len <= ow.flushed;
  /* runs (length > 4 x offset: one gather pass per offset's worth of bytes) and matches straddling the window's
   * lower edge are expanded by the whole wave after everything before them: such a sequence ends its batch */
  const bool serial = has_match && ((near && match_len > 4 * s.match_off) || (!near && !in_hbm));
  {
    const uint64_t cut = wave::ballot(serial) & ((take < 64 ? (1ull << take) : 0ull) - 1ull);
    if (cut) {
      /* the empty sequences behind it (Snappy: the followers of a copy train) go with it */
      const uint32_t f = wave::ctz64(cut);
      const uint64_t later = wave::ballot(active && len != 0) & ~((2ull << f) - 1ull);
      const uint32_t next = later ? wave::ctz64(later) : n;
      take = next < take ? next : take;
    }
  }
  const bool mine = lane < take;
  const uint32_t total = wave::read_lane(incl, take - 1);

  if (CHECKED) {
    const bool bad_out = mine && ((uint64_t)op + incl > out_cap);
    const bool bad_off = mine && has_match && (s.match_off == 0 || s.match_off > match_dst);
    const uint64_t any_out = wave::ballot(bad_out);
    const uint64_t any_off = wave::ballot(bad_off);
    if (any_out | any_off) {
      err |= (any_out ? lz::kErrOutput : 0u) | (any_off ? lz::kErrOffset : 0u);
      return 0;
    }
  }
  LZ_STAT("batches", 1);
  LZ_STAT("seqs", take);
  LZ_STAT("bytes", total);

  const bool my_match = mine && has_match;
  const bool my_serial = my_match && serial; /* at most one: the last non-empty sequence taken */
  /* far data: two 16-byte loads cover a match of up to kFarSlot bytes; they must stay inside the chunk's buffer */
  const bool far_stage = my_match && in_hbm && match_len <= kFarSlot && (uint64_t)match_src + kFarSlot <= out_cap;
  const bool far_coop = my_match && in_hbm && !far_stage;
  wave::u32x4 far0 = {0, 0, 0, 0}, far1 = {0, 0, 0, 0};
  if (far_stage) {
    const uint8_t* src = ow.out + match_src;
    far0 = wave::gload_u32x4(src);
    if (match_len > 16) {
      far1 = wave::gload_u32x4(src + 16);
    }
  }

  /* literals come straight out of the ring when the run is resident and does not cross the ring's end
   * (16 mirrored bytes behind it) */
  const bool my_lit = mine && lit_len != 0;
  const uint32_t ring_at = s.lit_src & (lzw::kInRing - 1);
  const bool lit_ring = my_lit && lzw::in_resident(ir, s.lit_src, s.lit_src + lit_len)
                        && ring_at + lit_len <= lzw::kInRing + 16;
  const bool lit_coop = my_lit && !lit_ring;

  /* ---- the uncommon sources are put in place by the whole wave (identity segments afterwards) ---- */
  {
    uint64_t pending = wave::ballot(lit_coop);
    LZ_STAT("lit_coop", wave::popc64(pending));
    while (pending) {
      const uint32_t j = wave::ctz64(pending);
      pending &= pending - 1;
      const uint32_t jsrc = wave::read_lane(s.lit_src, j);
      const uint32_t jlen = wave::read_lane(lit_len, j);
      const uint32_t jdst = wave::read_lane(lit_dst, j);
      lzw::copy_to_lds(lzw::out_at(ow, jdst), ir.base + jsrc, jlen);
    }
    pending = wave::ballot(far_coop);
    LZ_STAT("match_far_coop", wave::popc64(pending));
    while (pending) {
      const uint32_t j = wave::ctz64(pending);
      pending &= pending - 1;
      const uint32_t jsrc = wave::read_lane(match_src, j);
      const uint32_t jlen = wave::read_lane(match_len, j);
      const uint32_t jdst = wave::read_lane(match_dst, j);
      lzw::copy_to_lds(lzw::out_at(ow, jdst), ow.out + jsrc, jlen);
    }
  }

  /* ---- publish the segments: start bits and deltas ---- */
  const uint32_t a0 = (op + ow.align) & ~15u; /* address-congruent coordinate of block 0 */
  uint8_t* const blocks = ow.win + (a0 - ow.wbase); /* window address of batch byte 0 (wbase is a multiple of 16) */
  const uint32_t lit_at = head + incl - len;        /* batch byte index of the literals / the match */
  const uint32_t match_at = lit_at + lit_len;
  const uint64_t lit_mask = wave::ballot(my_lit);
  const uint64_t match_mask = wave::ballot(my_match);
  const uint32_t lit_rank = 1 + wave::prefix_popc(lit_mask) + wave::prefix_popc(match_mask);
  const uint32_t match_rank = lit_rank + (my_lit ? 1u : 0u);
  const uint32_t end_rank = 1 + wave::popc64(lit_mask) + wave::popc64(match_mask);
  if (my_lit) {
    sc.delta[lit_rank] = lit_ring ? (int32_t)((ir.ring + ring_at) - (blocks + lit_at)) : 0;
    wave::lds_or(sc.map + (lit_at >> 4), 1u << (lit_at & 15u));
  }
  if (my_match) {
    const int32_t d = far_stage ? (int32_t)((sc.stage + lane * kFarSlot) - (blocks + match_at))
                      : near && !serial ? -(int32_t)s.match_off
                                        : 0;
    sc.delta[match_rank] = d;
    wave::lds_or(sc.map + (match_at >> 4), 1u << (match_at & 15u));
  }
  const uint32_t end_at = head + total;
  if (lane == take - 1 && end_at < kCap) { /* what lies behind the batch is gathered from itself */
    sc.delta[end_rank] = 0;
    wave::lds_or(sc.map + (end_at >> 4), 1u << (end_at & 15u));
  }
  wave::sync();

  /* ---- every byte finds its segment ---- */
  const uint32_t bits = sc.map[lane];
  sc.map[lane] = 0; /* ready for the next batch */
  const uint32_t cnt = (uint32_t)__builtin_popcount(bits);
  const uint32_t before = wave::scan_add_inclusive(cnt) - cnt;
  const uint8_t* src[16];
  {
    const uint8_t* mine16 = blocks + lane * kBytesPerLane;
#pragma unroll
    for (uint32_t b = 0; b < 16; ++b) {
      const uint32_t rank = before + (uint32_t)__builtin_popcount(bits & ((2u << b) - 1u));
      src[b] = mine16 + sc.delta[rank];
    }
  }
  LZW_T(6);

  /* ---- far data into its slots ---- */
  if (far_stage) {
    *(wave::u32x4*)(sc.stage + lane * kFarSlot) = far0;
    if (match_len > 16) {
      *(wave::u32x4*)(sc.stage + lane * kFarSlot + 16) = far1;
    }
  }
  LZ_STAT("match_far_lanes", wave::popc64(wave::ballot(far_stage)));
  wave::sync();
  LZW_T(7);

  /* ---- gather; repeat while matches read bytes of this batch that were not final yet ---- */
  wave::u32x4 g = gather16(src);
  wave::sync(); /* every lane has read before any lane writes */
  *(wave::u32x4*)(blocks + lane * kBytesPerLane) = g;
  wave::sync();
  {
    /* bytes below op are final wherever they sit */
    const bool dep = my_match && near && !serial && match_src + match_len > op;
    const uint64_t dep_mask = wave::ballot(dep);
    LZ_STAT("match_dep_lanes", wave::popc64(dep_mask));
    if (dep_mask) {
      /* one repeat is enough when no dependent match reads behind the first dependent match's output */
      const uint32_t first_dst = wave::read_lane(match_dst, wave::ctz64(dep_mask));
      const bool chained = wave::ballot(dep && match_src + match_len > first_dst) != 0;
      bool again = true;
      while (again) {
        const wave::u32x4 g2 = gather16(src);
        again = chained && wave::ballot(differs(g, g2)) != 0;
        g = g2;
        wave::sync();
        *(wave::u32x4*)(blocks + lane * kBytesPerLane) = g;
        wave::sync();
        LZ_STAT("gather_repeats", 1);
      }
    }
  }
  LZW_T(8);

  /* ---- the run / straddling match that ended the batch ---- */
  if (const uint64_t serial_mask = wave::ballot(my_serial)) {
    const uint32_t f = wave::ctz64(serial_mask); /* the last non-empty sequence of the batch */
    const uint32_t hw = wave::read_lane(match_dst, f);
    const uint32_t foff = wave::read_lane(s.match_off, f);
    const uint32_t flen = wave::read_lane(match_len, f);
    const uint32_t fsrc = hw - foff;
    LZ_STAT("match_serial", 1);
    if (fsrc >= ow.valid_lo) {
      lzw::lds_match_copy(lzw::out_at(ow, hw), foff, flen);
    } else {
      /* bytes below valid_lo come from HBM (flushed before the window let go of them), the rest is in the window */
      const uint32_t n_hbm = fsrc + flen <= ow.valid_lo ? flen : ow.valid_lo - fsrc;
      lzw::copy_to_lds(lzw::out_at(ow, hw), ow.out + fsrc, n_hbm);
      wave::sync();
      if (n_hbm < flen) {
        lzw::lds_match_copy(lzw::out_at(ow, hw + n_hbm), foff, flen - n_hbm);
      }
    }
    wave::sync();
    g = *(const wave::u32x4*)(blocks + lane * kBytesPerLane);
  }

  /* ---- whole 16-byte blocks go to HBM from the registers; the last partial block waits for the next batch ---- */
  {
    const uint32_t a_end = (op + total + ow.align) & ~15u;
    uint8_t* gout = ow.out - ow.align; /* gout + a == out + position */
    const uint32_t a = a0 + lane * kBytesPerLane;
    if (a0 < ow.align) {
      /* first block of a chunk whose output pointer is not 16-byte aligned: its leading bytes are not ours */
      if (a_end > a0) {
        if (lane >= ow.align && lane < 16) {
          wave::gstore_u8(gout + a0 + lane, blocks[lane]);
        }
      }
      if (lane != 0 && a < a_end) {
        wave::gstore_u32x4_aligned(gout + a, g);
      }
    } else if (a < a_end) {
      wave::gstore_u32x4_aligned(gout + a, g);
    }
    if (a_end > ow.flushed + ow.align) {
      ow.flushed = a_end - ow.align;
    }
  }
  LZW_T(9);
  wave::sync(); /* later far reads of this wave must see the stored bytes */
  op += total;
  return take;
}

constexpr uint32_t kLdsPerWave = lzw::kLdsPerWave + (NVCOMP_LZ_GATHER ? kScratch : 0);
constexpr uint32_t kLdsPerWaveIndexed = lzw::kLdsPerWaveIndexed + (NVCOMP_LZ_GATHER ? kScratch : 0);

/* `ow.scratch` must point at kScratch bytes of this wave's LDS (ignored by the round-1 executor). */
__device__ __forceinline__ void attach_scratch(lzw::OutWindow& ow, uint8_t* lds)
{
  ow.scratch = lds;
#if NVCOMP_LZ_GATHER
  scratch_init(lds);
#endif
}

/* After a sequence was streamed HBM -> HBM behind the window's back (the callers' `big` path), restart the window at
 * op. The gather executor stores whole 16-byte blocks, so the bytes of op's block that are already out must be in
 * the window again: they are read back (at most 15 bytes). */
__device__ __forceinline__ void restart_window(lzw::OutWindow& ow, uint32_t op)
{
  ow.wbase = op & ~15u;
  ow.valid_lo = op;
  ow.flushed = op;
#if NVCOMP_LZ_GATHER
  const uint32_t head = (op + ow.align) & 15u;
  const uint32_t from = op - (head < op ? head : op);
  const uint32_t pos = from + (uint32_t)wave::lane_id();
  if (pos < op) {
    *lzw::out_at(ow, pos) = (uint8_t)wave::gload_u8(ow.out + pos);
  }
  ow.valid_lo = from;
  wave::sync();
#endif
}

template <bool CHECKED>
__device__ __forceinline__ uint32_t execute_batch(
    lzw::InRing& ir, lzw::OutWindow& ow, uint32_t out_cap, uint32_t& op, uint32_t n, const lz::Seq& s, uint32_t& err,
    bool& big)
{
#if NVCOMP_LZ_GATHER
  return execute_gather_batch<CHECKED>(ir, ow, ow.scratch, out_cap, op, n, s, err, big);
#else
  return lzw::execute_window_batch<CHECKED>(ir, ow, out_cap, op, n, s, err, big);
#endif
}

} // namespace lzg
