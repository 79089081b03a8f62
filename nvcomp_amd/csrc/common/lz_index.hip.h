/*
 * common/lz_index.hip.h -- the token index of a chunk: where its sequences start, found by 64 serial walks instead of a
 * speculation over every stream position (round 6).
 *
 * WHY. Measured in round 6 (scripts/microbench/valu_issue.hip, profiles/r06_valu_issue_*.jsonl): a wave64 vector
 * instruction of the decoders' mix holds its SIMD for 4 cycles, the window decoder is at 0.93-0.95 of vector issue, and
 * folding its far-match loads onto resident lines moves it by 4 % (profiles/r06_far_ablate.jsonl): the only lever is the
 * instruction count. The token chase of common/lz_window.hip.h (chase_build + chase_tokens) is 30 % of a wave's time: it
 * computes "the distance to the next token" for EVERY stream position -- 256 positions for the ~54 tokens among them --
 * and builds five jump tables over them.
 *
 * WHAT. A token chain is serial, but it synchronises itself: a walk that starts at a wrong position falls onto the
 * true chain after a few dozen bytes (the literals it misreads as tokens are short). So the chunk's stream is cut into
 * 64 segments, lane k walks segment k -- ONE token per step and lane, ~20 instructions, no speculation per position --
 * after a run-in of kRunIn bytes in front of it, and the segments are joined EXACTLY:
 *   entry  E_k = the first token position >= S_k on lane k's chain,   exit X_k = the first one >= S_{k+1};
 *   lane 0 starts on the chunk's first token, so its chain is the true one; lane k's is true from E_k on iff
 *   E_k == X_{k-1} (induction over k). A lane whose entry is not its predecessor's exit walks again from that exit
 *   (a few rounds at most); what still does not join, a token the walk's straight-line step cannot size (length
 *   fields of three and more bytes, a long literal run in front of a long match), and the last bytes of the chunk are
 *   left to the classic chase: the index is a PREFIX of the chunk's tokens plus the position where the chase takes
 *   over. Nothing is trusted that was not walked from a position known to be a token.
 * A lane writes the positions it passes (16 bits each, from its segment's start on) into ITS OWN list in the caller's
 * temp buffer (kLaneCap entries per lane, this wave's slice); the decoder's batches read them segment after segment
 * (Reader below): one or two coalesced loads where they used to build and enumerate jump tables.
 *
 * HOW THE BYTES GET THERE. Every lane holds a block of 64 stream bytes in LDS (stride 68: conflict-free byte reads). All
 * lanes refill TOGETHER ("phase"): four 16-byte loads per lane, one wait for the whole wave, then every lane walks until
 * its next token needs a byte outside its block -- no memory operation inside the walk's loop. The first version of this
 * file let a lane refill by itself whenever it ran out: with 64 lanes some lane ran out in every step, and every step cost
 * a loaded memory round trip (4 200 cycles a step on the mix, the decoder fell from 665 to 440 GB/s: gpurun r6e).
 */
#pragma once

#include "common/lz_common.hip.h"

#ifndef NVCOMP_LZX_RUNIN
#define NVCOMP_LZX_RUNIN 256
#endif
#ifndef NVCOMP_LZX_FIX_ROUNDS
#define NVCOMP_LZX_FIX_ROUNDS 3
#endif

namespace lzx {

constexpr uint32_t kRunIn = NVCOMP_LZX_RUNIN;          /* bytes walked in front of a segment to fall onto the true chain */
constexpr uint32_t kFixRounds = NVCOMP_LZX_FIX_ROUNDS; /* re-walks of lanes whose entry is not their predecessor's exit */
constexpr uint32_t kBlock = 64;                        /* stream bytes a lane holds in LDS */
constexpr uint32_t kStride = 68;                       /* LDS bytes between two lanes' blocks: 17 dwords, odd -> no bank shared */
constexpr uint32_t kLdsBytes = 64 * kStride + 512;     /* of the wave's LDS, free while the index is built (+ what a
                                                        * speculative read behind the last lane's block may touch) */
constexpr uint32_t kTail = kBlock + 8;                 /* the last bytes of a chunk are never indexed (no load passes its end) */
constexpr uint32_t kMinStream = 2048;                  /* shorter streams are not worth an index */
constexpr uint32_t kMaxStream = 0xffffu;               /* positions are stored in 16 bits */
constexpr uint32_t kLaneCap = 344; /* (a multiple of 4: positions are stored four at a time) */
constexpr uint32_t kLaneCapUnused = 0;                     /* positions a lane's list holds: a segment is at most 1 023 bytes, a
                                                        * sequence at least 3 (Snappy: 2 -- a lane that fills its list stops) */
constexpr size_t kScratchPerWave = 2 * 64 * (size_t)kLaneCap;
constexpr uint32_t kNone = 0xffffffffu;

/* What the decoder keeps of an index: the lists (this wave's slice of the temp buffer), how many lanes' lists are valid,
 * every lane's count (lane k: n of list k), the reader's cursor, and the stream position (chunk offset) where the
 * classic chase takes over. */
struct Index
{
  const uint16_t* lists;
  uint32_t lanes;  /* lists 0 .. lanes - 1 are the chunk's first tokens, in order */
  uint32_t n;      /* PER LANE: entries of list `lane` (0 for lane >= lanes) */
  uint32_t seg;    /* reader: the list being read ... */
  uint32_t at;     /* ... and the next entry in it */
  uint32_t a0, a1; /* PER LANE: the entries behind the cursor, requested a round ahead (peek) */
  uint32_t n0;     /* ... of which the first n0 lanes' come from list `seg` (a0), the others from the next one (a1) */
  uint32_t ahead_n;
  uint32_t resume;
};

/* One walk of all lanes, in phases. Format::step(blk, o, next_o, ok) looks at the token at block offset o (o + 1 < kBlock):
 * true = its successor starts at block offset next_o (possibly behind the block); false = the block does not hold what it
 * takes to tell (ok: a block that starts at the token will; !ok: nothing will -- the walk stops there).
 * Lane state: p (chunk offset of the next token). Tokens at p >= from are recorded in `list` (at most kLaneCap).
 * The loop over a block's tokens is written for the scalar unit as much as for the vector unit: one exit condition, a
 * straight-line body, no min / max bookkeeping (the entry is read back from the list afterwards). */
template <class Format>
__device__ __forceinline__ void walk(
    const uint8_t* in, uint32_t in_len, uint8_t* blk, bool take_part, uint32_t start, uint32_t from, uint32_t limit,
    uint16_t* list, uint32_t& entry, uint32_t& n, uint32_t& exit_pos, bool& stopped)
{
  uint32_t p = start;
  uint32_t cnt = 0;
  uint32_t w0 = 0, w1 = 0; /* the last four positions recorded, the newest on top */
  bool stop = false;
  bool active = take_part && p < limit;
  while (wave::ballot(active) != 0) {
    /* ---- refill: every active lane's block starts at its next token (p + kBlock <= in_len: p < limit <= in_len - kTail) ---- */
    if (active) {
      const wave::u32x4 x0 = wave::gload_u32x4(in + p), x1 = wave::gload_u32x4(in + p + 16);
      const wave::u32x4 x2 = wave::gload_u32x4(in + p + 32), x3 = wave::gload_u32x4(in + p + 48);
      uint32_t* w = (uint32_t*)blk;
      w[0] = x0.x, w[1] = x0.y, w[2] = x0.z, w[3] = x0.w;
      w[4] = x1.x, w[5] = x1.y, w[6] = x1.z, w[7] = x1.w;
      w[8] = x2.x, w[9] = x2.y, w[10] = x2.z, w[11] = x2.w;
      w[12] = x3.x, w[13] = x3.y, w[14] = x3.z, w[15] = x3.w;
    }
    wave::sync_wave();
    /* ---- walk inside the block ---- */
    if (active) {
      const uint32_t bpos = p;
      /* block offsets from which tokens are recorded / at which the lane leaves the block */
      const uint32_t from_o = from > bpos ? from - bpos : 0u;
      const uint32_t end_o = limit - bpos < kBlock - 1 ? limit - bpos : kBlock - 1;
      uint32_t o = 0;
      bool go, ok;
      do {
        uint32_t next_o;
        go = Format::step(blk, o, next_o, ok) & (cnt < kLaneCap);
        if (go & (o >= from_o)) {
          /* four positions travel in a 64-bit shift register and leave with ONE 8-byte store (a 2-byte store per token
           * and lane was 200 M scattered requests a launch: the index alone took 2.4 ms per 32 768 chunks, gpurun r6h) */
          w0 = wave::align_bits(w1, w0, 16);
          w1 = wave::align_bits(bpos + o, w1, 16);
          cnt += 1;
          if ((cnt & 3u) == 0) {
            uint32_t* q = (uint32_t*)(list + cnt - 4);
            q[0] = w0, q[1] = w1;
          }
        }
        o = go ? next_o : o;
      } while (go & (o < end_o));
      /* the lane stops for good at a token that nothing will size, at one that a block of its own did not, at a full list */
      stop = (!go) & (!ok | (o == 0) | (cnt >= kLaneCap));
      p = bpos + o;
      active = !stop & (p < limit);
    }
  }
  /* what did not fill a group of four: moved down to the group's start (the list has room for whole groups) */
  if ((cnt & 3u) != 0) {
    const uint64_t w = (((uint64_t)w1 << 32) | w0) >> (16 * (4 - (cnt & 3u)));
    uint32_t* q = (uint32_t*)(list + (cnt & ~3u));
    q[0] = (uint32_t)w, q[1] = (uint32_t)(w >> 32);
  }
  n = cnt;
  stopped = stop;
  exit_pos = p;
  /* the entry: the first token at or behind `from` -- the list's first entry; without one the exit itself, unless the walk
   * stopped in front of `from` (a run-in that met a token it cannot size: this lane knows nothing) */
  uint32_t first = p;
  if (cnt != 0) {
    first = wave::gload_u16(list);
  }
  entry = cnt != 0 ? first : (stop && p < from ? kNone : p);
}

__device__ __forceinline__ void peek(Index& ix);

/* Build the index of one chunk with the calling wave. `lds`: kLdsBytes of the wave's LDS (free at this point); `scratch`:
 * the wave's slice of the temp buffer (kScratchPerWave bytes, 2-byte aligned). Returns an empty index (no lanes, resume 0)
 * for streams it is not made for. */
template <class Format>
__device__ __forceinline__ Index build(const uint8_t* in, uint32_t in_len, uint8_t* lds, uint8_t* scratch)
{
  Index ix;
  ix.lists = (const uint16_t*)scratch;
  ix.lanes = 0;
  ix.n = 0;
  ix.seg = 0;
  ix.at = 0;
  ix.a0 = 0;
  ix.a1 = 0;
  ix.n0 = 0;
  ix.ahead_n = 0;
  ix.resume = 0;
  if (scratch == nullptr || in_len < kMinStream || in_len > kMaxStream) {
    return ix;
  }
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  uint8_t* blk = lds + kStride * lane;
  uint16_t* list = (uint16_t*)scratch + kLaneCap * lane;
  const uint32_t region = in_len - kTail;     /* tokens at [first, region) are indexed: >= 1 976 bytes */
  const uint32_t seg = region >> 6;           /* 30 .. 1 022 */
  const uint32_t s_lo = seg * lane;
  const uint32_t s_hi = lane == 63 ? region : s_lo + seg;
  const uint32_t first = Format::first_token(in, in_len); /* uniform: 0 for LZ4, behind the preamble for Snappy */
  if (first >= seg) {
    return ix;
  }

  /* ---- the walk: from the run-in to the segment's end ---- */
  uint32_t entry, n, exit_pos;
  bool stopped;
  {
    const uint32_t back = s_lo > kRunIn ? s_lo - kRunIn : 0u;
    const uint32_t start = back > first ? back : first; /* lanes whose run-in reaches the chunk's start walk the true chain */
#ifndef NVCOMP_LZX_NO_TOUCH
    /* every lane's stretch of the stream is asked for at once (a load per 128-byte line, the data is not looked at): the
     * phases of the walk then wait for the L2, not for HBM -- one long round trip per chunk instead of one per phase
     * (the decoder with an index was 26 % fewer vector instructions and 21 % MORE time: waiting, gpurun r6pmc) */
    {
      uint32_t acc = 0;
      for (uint32_t at = start & ~127u; at < s_hi + kBlock; at += 128) {
        acc |= wave::gload_u8(in + (at < in_len ? at : in_len - 1));
      }
      wave::touch(acc);
    }
#endif
    walk<Format>(in, in_len, blk, true, start, s_lo, s_hi, list, entry, n, exit_pos, stopped);
  }
  /* ---- join: a lane's entry must be its predecessor's exit ---- */
  uint64_t bad = 0;
  for (uint32_t round = 0;; ++round) {
    const uint32_t px = wave::prev_lane(exit_pos);
    const uint32_t pstop = wave::prev_lane(stopped ? 1u : 0u);
    const uint32_t want = lane == 0 ? first : px; /* lane 0: the chunk's first token (its own walk started there) */
    /* behind a lane that stopped nothing is known: such lanes are cut off below, they need no second walk */
    bad = wave::ballot(lane != 0 && pstop == 0 && entry != want);
    if (bad == 0 || round == kFixRounds) {
      break;
    }
    /* again from the predecessor's exit (>= s_lo by construction), writing the list anew; past the segment's end:
     * nothing to walk */
    const bool again = wave::lane_in(bad);
    const bool passes = again && want >= s_hi;
    uint32_t e2, n2, x2;
    bool st2;
    walk<Format>(in, in_len, blk, again && !passes, want, want, s_hi, list, e2, n2, x2, st2);
    if (again) {
      entry = passes ? want : e2;
      n = passes ? 0u : n2;
      exit_pos = passes ? want : x2;
      stopped = passes ? false : st2;
    }
  }
  /* ---- the prefix that joined: lanes [0, K) ---- */
  const uint64_t stop_mask = wave::ballot(stopped);
  uint32_t K = bad ? wave::ctz64(bad) : 64u;
  if (stop_mask) {
    const uint32_t s = wave::ctz64(stop_mask) + 1; /* the lane that stopped still contributes what it counted */
    K = s < K ? s : K;
  }
  if (K == 0 || wave::ballot(lane < K && n != 0) == 0) {
    return ix; /* nothing joined, or no token in what did (a chunk that opens with a run the walk cannot size) */
  }
  wave::sync(); /* the decoder's loads of the lists are this wave's own: served behind the stores above */
  ix.lanes = K;
  ix.n = lane < K ? n : 0u;
  ix.resume = wave::read_lane(exit_pos, K - 1);
  peek(ix);
  return ix;
}

/* The reader. A batch's positions are loaded ONE ROUND AHEAD (`a0` / `a1`: the next entries from the cursor on, out of
 * at most two lists -- lane j < n0 holds entry j of the first, lane n0 + i entry i of the second): the load's round trip
 * -- thousands of cycles on a card whose memory system is busy with far matches; waited for in place it was a fifth of the
 * decoder's time (gpurun r6p) -- passes behind a whole batch's execution. Two registers, no loop: a value merged in a loop
 * is waited for at the merge (the first version's `v = loaded ? load : v` put an s_waitcnt vmcnt(0) right behind the load).
 * The decoder calls settle() where it waits for its far matches anyway: the registers are then plain values, and the next
 * round's read() does not wait behind the stores of the batch's flush (vmcnt counts in issue order). */
__device__ __forceinline__ void peek(Index& ix)
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  uint32_t seg = ix.seg, at = ix.at;
  /* lists without entries behind the cursor are passed over (a long sequence covers a whole segment) */
  uint32_t have = 0;
  while (seg < ix.lanes && (have = wave::read_lane(ix.n, seg) - at) == 0) {
    seg += 1;
    at = 0;
  }
  ix.seg = seg;
  ix.at = at;
  ix.n0 = 0;
  ix.ahead_n = 0;
  if (seg >= ix.lanes) {
    return;
  }
  const uint32_t n0 = have < 64 ? have : 64u;
  if (lane < n0) {
    ix.a0 = wave::gload_u16(ix.lists + seg * kLaneCap + at + lane);
  }
  uint32_t n1 = 0;
  if (n0 < 64 && seg + 1 < ix.lanes) {
    const uint32_t have1 = wave::read_lane(ix.n, seg + 1);
    n1 = have1 < 64 - n0 ? have1 : 64 - n0;
    if (lane - n0 < n1) {
      ix.a1 = wave::gload_u16(ix.lists + (seg + 1) * kLaneCap + (lane - n0));
    }
  }
  ix.n0 = n0;
  ix.ahead_n = n0 + n1;
}

/* Are there entries left? (uniform) */
__device__ __forceinline__ bool more(const Index& ix)
{
  return ix.ahead_n != 0;
}

/* Wait for the entries requested by peek() -- called where the decoder waits for vector memory anyway. */
__device__ __forceinline__ void settle(Index& ix)
{
  wave::touch(ix.a0);
  wave::touch(ix.a1);
}

/* The next token positions, into seqpos lanes [count, 64): chunk offsets + `bias`. Returns the new count. */
__device__ __forceinline__ uint32_t read(Index& ix, uint32_t& seqpos, uint32_t count, uint32_t bias)
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  const uint32_t room = 64 - count;
  const uint32_t take = ix.ahead_n < room ? ix.ahead_n : room;
  const uint32_t mine = lane < ix.n0 ? ix.a0 : ix.a1;
  const uint32_t v = wave::shuffle(mine, (lane - count) & 63u);
  if (lane - count < take) {
    seqpos = bias + v;
  }
  /* the cursor moves behind what was taken (at most into the second list) ... */
  if (take < ix.n0) {
    ix.at += take;
  } else {
    const uint32_t into = take - ix.n0;
    const bool whole = ix.n0 == wave::read_lane(ix.n, ix.seg) - ix.at; /* (always, unless the first list had more than 64 left) */
    if (whole) {
      ix.seg += 1;
      ix.at = into;
    } else {
      ix.at += take;
    }
  }
  peek(ix); /* ... and the entries behind it are requested for the next round */
  return count + take;
}

} // namespace lzx
