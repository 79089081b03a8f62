/*
 * common/lz_team.hip.h -- a WORKGROUP per chunk: the LZ77 decoder for batches that cannot fill the card with one wave per
 * chunk (nvcompBatched{LZ4,Snappy}DecompressAsync below lzl::kTeamMaxBatch chunks; reference call sites
 * benchmarks/benchmark_template_chunked.cuh:519-530, benchmarks/benchmark_lz4_synth.cpp:64-72 -- 1 ... 8 192 chunks --,
 * doc/Benchmarks.md:88-95 -- 5 021 chunks).
 *
 * Why: a wave needs ~0.4 ms for a 64 KiB chunk of text-like data whatever the load -- ~1 000 dependent instructions per batch
 * of 64 sequences, measured the same with every match served from LDS (profiles/r04_feasibility.json) -- so a batch of a few
 * thousand chunks leaves most of the card idle for most of that time. Here eight waves share ONE chunk:
 *
 *   buffer   the chunk's whole OUTPUT (up to 64 KiB) lives in LDS, and so does its whole compressed STREAM, right-aligned
 *            behind it in the same buffer ("in place": a decoder's write position never passes its read position by more
 *            than the format's overhead, kMargin covers it; checked every step, a stream that breaks it -- or a chunk
 *            that does not fit -- is decoded by waves 0 and 1 with the two-wave decoder). Every match is an LDS -> LDS copy:
 *            no far-match gathers, HBM traffic = stream in + output out.
 *   step     per step the eight waves look at eight consecutive windows of kPositions stream positions, one each:
 *              1. every wave builds the jump tables of its window (lzw::chase_build: they do not depend on where the
 *                 token chain enters the window) and tabulates the window's EXIT for a chain that enters at each of its
 *                 first 16 offsets (a lane-parallel descent of the tables);
 *              2. a wave finds its entry by walking the exit tables of the windows in front of it (exact for entries at
 *                 those offsets, a guess -- offset 0's exit: chains merge within a few tokens -- for the rare entry further
 *                 in), enumerates and parses its own tokens; a window that was entered by something else than the true
 *                 exit of the window in front ends the step there (the waves behind it repeat their work in the next step);
 *              3. a prefix sum over the waves' output sizes places every sequence; literals are copied;
 *              4. matches are copied as soon as their source bytes are final: per-granule counters of pending match
 *                 destinations (16 bytes a granule) tell, the oldest pending match of the step is always free to go.
 *            Three workgroup barriers a step (the tables of step s + 1 are built in front of the barrier that ends step s:
 *            whoever finishes its matches early has something to do).
 *
 * kPositions is the format's: the largest window that cannot hold more than 64 tokens (LZ4: 192 = 64 x 3 bytes), so a
 * wave's window is one batch, one sequence per lane.
 */
#pragma once

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "common/lz_team.hip.h: the workgroup-per-chunk decoders hold a chunk's output and stream in 79-91 KB of static LDS: gfx950 (160 KB per CU) only"
#endif

#include "common/lz_window.hip.h"

namespace lzt {

#ifdef LZT_TRACE /* host emulation only: where a team is */
#define LZT_TR(...) do { if (wave::lane_id() == 0) { fprintf(stderr, __VA_ARGS__); } } while (0)
#else
#define LZT_TR(...) ((void)0)
#endif

#ifndef NVCOMP_LZT_MARGIN
#define NVCOMP_LZT_MARGIN 2560
#endif
constexpr uint32_t kMaxOut = 65536;              /* the largest chunk (output capacity) a team takes */
constexpr uint32_t kMargin = NVCOMP_LZT_MARGIN;  /* room between the end of the output and the end of the stream */
constexpr uint32_t kBuf = kMaxOut + kMargin + 32; /* + the two 16-byte alignment slacks (output front, stream back) */
constexpr uint32_t kBufLds = kBuf + 32;          /* zeroes behind the stream: field reads may run past its end */
constexpr uint32_t kMaxIn = kMaxOut + 512;       /* compressed bytes a team takes (LZ4's bound is 65 809) */
constexpr uint32_t kGran = 16;                   /* bytes per readiness granule */
constexpr uint32_t kTrack = 16384;               /* output bytes of a step the granule counters cover */
constexpr uint32_t kGranules = kTrack / kGran;
constexpr uint32_t kCntLds = kGranules + 16;     /* one byte a granule (+ what a 4-counter read may touch behind the last) */
constexpr uint32_t kCtlWords = 352;
constexpr uint32_t kEntries = 16;               /* entry offsets of a window whose exits are tabulated */
constexpr uint32_t kMaxWaves = 16;
constexpr uint32_t kSpinLimit = 1u << 18;       /* polls of a waiting wave (tens of milliseconds) before the chunk is given up */

/* A team of WAVES waves (8: two teams per CU, 16: one -- a batch of at most one chunk per CU): threads and LDS. */
template <uint32_t WAVES>
struct Geo
{
  static_assert(WAVES == 8 || WAVES == 16, "teams of 8 or 16 waves");
  static constexpr uint32_t kWaves = WAVES;
  static constexpr uint32_t kThreads = 64 * WAVES;
  static constexpr uint32_t kLds = kBufLds + WAVES * lzw::kChaseLds + kCntLds + 4 * kCtlWords;
  static_assert(WAVES * lzw::kChaseLds >= lzw::pair::kLdsPerChunk, "the fallback decoder's scratch is the table area");
  static_assert(kLds <= (WAVES == 8 ? 81920u : 163840u), "two workgroups of 8 waves per CU, or one of 16");
};

constexpr uint32_t kUnknownExit = 0xffffffffu;

/* control words in LDS */
enum : uint32_t {
  kCtlErr = 0,
  kCtlFallback = 1,
  kCtlTicket = 2,
  kCtlExit = 4,   /* [kMaxWaves] true exit of window w */
  kCtlBytes = 20, /* [kMaxWaves] output bytes of window w's sequences */
  kCtlProg = 36,  /* [kMaxWaves] everything of slot w below this output position is final */
  kCtlBad = 52,   /* [kMaxWaves] parse error */
  kCtlUsed = 68,  /* [kMaxWaves] the entry window w was enumerated from */
  kCtlSpec = 84,  /* [kMaxWaves][kEntries] exit of window w for a chain that enters it at offset 0 .. kEntries - 1 */
};

/* The whole chunk's stream, staged in LDS: ring[v] is the byte at virtual position v (v = chunk offset + (chunk & 15)),
 * the same interface as lzw::InRing without the wrap. */
struct Stream
{
  static constexpr uint32_t kMask = 0xffffffffu;
  const uint8_t* base; /* chunk pointer rounded down to 16 bytes */
  uint8_t* ring;
  uint32_t vbeg, vend;
  uint32_t lo, hi; /* resident range: everything (hi reaches into the zeroes behind the stream) */
};

struct Team
{
  uint8_t* buf;  /* kBufLds: output position p at buf[p + oa] */
  uint8_t* tab;  /* this wave's jump tables */
  uint32_t* cnt; /* granule counters, a byte each */
  uint32_t* ctl;
  uint8_t* out;  /* the chunk's output in HBM */
  uint32_t oa;   /* out & 15 */
  uint32_t w;    /* wave of the team */
  Stream st;
};

__device__ __forceinline__ uint32_t ctl_read(const Team& t, uint32_t i)
{
  return wave::uniform(wave::lds_load_acquire(t.ctl + i));
}

/* all lanes of the team: chunk -> LDS, 16 bytes per lane and step, bytes outside the chunk never fetched (read as zero) */
template <uint32_t kThreads>
__device__ __forceinline__ void stage_stream(const Stream& st, uint32_t tid)
{
  const uint32_t nblocks = ((st.vend + 15u) >> 4) + 1u; /* + one block of zeroes */
  for (uint32_t b = tid; b < nblocks; b += kThreads) {
    const uint32_t v = 16u * b;
    wave::u32x4 x = {0, 0, 0, 0};
    if (v >= st.vbeg && v + 16 <= st.vend) {
      x = wave::gload_u32x4_aligned(st.base + v);
    } else if (v + 16 > st.vbeg && v < st.vend) {
      uint32_t wd[4] = {0, 0, 0, 0};
#pragma unroll
      for (uint32_t j = 0; j < 16; ++j) {
        if (v + j >= st.vbeg && v + j < st.vend) {
          wd[j >> 2] |= wave::gload_u8(st.base + v + j) << (8 * (j & 3));
        }
      }
      x.x = wd[0], x.y = wd[1], x.z = wd[2], x.w = wd[3];
    }
    *(wave::u32x4*)(st.ring + v) = x;
  }
}

/* all lanes of the team: output positions [from, to) -> HBM; 16-byte aligned lane stores, bytes around them */
template <uint32_t kThreads>
__device__ __forceinline__ void flush(const Team& t, uint32_t from, uint32_t to, uint32_t tid)
{
  if (to <= from) {
    return;
  }
  const uint32_t a_from = from + t.oa, a_to = to + t.oa; /* buffer indices = address-congruent coordinates */
  uint32_t body_lo = (a_from + 15u) & ~15u;
  uint32_t body_hi = a_to & ~15u;
  if (body_lo > body_hi) {
    body_lo = a_to;
    body_hi = a_to;
  }
  uint8_t* gout = t.out - t.oa;
  if (tid < body_lo - a_from) {
    wave::gstore_u8(gout + a_from + tid, t.buf[a_from + tid]);
  }
  for (uint32_t a = body_lo + 16u * tid; a < body_hi; a += 16u * kThreads) {
    wave::gstore_u32x4_aligned(gout + a, *(const wave::u32x4*)(t.buf + a));
  }
  if (tid < a_to - body_hi) {
    wave::gstore_u8(gout + body_hi + tid, t.buf[body_hi + tid]);
  }
}

/* The token behind the last one of a window's chain (window offset `last`), from the deltas chase_build kept. */
template <class Slow>
__device__ __forceinline__ uint32_t exit_of(const lzw::Chase& c, const Stream& st, uint32_t last, bool speculative, Slow slow)
{
  const uint32_t pair = wave::read_lane((last & 2u) ? c.nx23 : c.nx01, last >> 2);
  const uint32_t d = (last & 1u) ? pair >> 16 : pair & 0xffffu;
  if (d != lzw::kNxUnknown) {
    return c.wb + last + d;
  }
  return speculative ? kUnknownExit : slow(st, c.wb + last);
}

/* The exit of the window for a chain that enters it at offset `start` (per lane): descend the jump tables to the chain's
 * last token -- at most 64 tokens: four jumps of 16, then 8, 4, 2, 1 -- and leave through that token's own delta. */
__device__ __forceinline__ uint32_t exit_from(const lzw::Chase& c, uint32_t start)
{
  uint32_t pos = start;
  const uint32_t top = (lzw::kChaseLevels - 1) * lzw::kChaseWin;
  bool go = true;
#pragma unroll
  for (uint32_t it = 0; it < 4; ++it) {
    const uint32_t a = c.tab[top + pos];
    go = go && a != 255u;
    pos += go ? a : 0u;
  }
#pragma unroll
  for (int32_t i = (int32_t)lzw::kChaseLevels - 2; i >= 0; --i) {
    const uint32_t a = c.tab[(uint32_t)i * lzw::kChaseWin + pos];
    pos += a == 255u ? 0u : a;
  }
  const uint32_t p01 = wave::shuffle(c.nx01, pos >> 2), p23 = wave::shuffle(c.nx23, pos >> 2);
  const uint32_t pair = (pos & 2u) ? p23 : p01;
  const uint32_t d = (pos & 1u) ? pair >> 16 : pair & 0xffffu;
  return d == lzw::kNxUnknown ? kUnknownExit : c.wb + pos + d;
}

/* Lane n: the n-th token of the chain that enters the window at offset pos0 (lzw::chase_tokens' enumeration); returns the
 * number of tokens (>= 1) and the window offset of the last one. */
__device__ __forceinline__ uint32_t enumerate(const lzw::Chase& c, uint32_t pos0, uint32_t& seqpos, uint32_t& last)
{
  const uint32_t idx = (uint32_t)wave::fresh_lane_id();
  const uint32_t top = (lzw::kChaseLevels - 1) * lzw::kChaseWin;
  const uint32_t t1 = c.tab[top + pos0];
  const uint32_t t2 = c.tab[top + ((pos0 + t1) & (lzw::kChaseWin - 1))];
  const bool upper = (idx & 32u) != 0;
  uint32_t pos = upper ? (pos0 + t1 + t2) & (lzw::kChaseWin - 1) : pos0;
  uint32_t worst = upper ? (t1 > t2 ? t1 : t2) : 0u;
#pragma unroll
  for (uint32_t i = 0; i < lzw::kChaseLevels; ++i) {
    const uint32_t a = c.tab[i * lzw::kChaseWin + pos];
    const uint32_t adv = a & (uint32_t)(-(int32_t)((idx >> i) & 1u));
    worst = adv > worst ? adv : worst;
    pos = (pos + adv) & (lzw::kChaseWin - 1);
  }
  const uint32_t count = wave::popc64(wave::ballot(worst != 255u));
  seqpos = idx < count ? c.wb + pos : 0u;
  last = wave::read_lane(pos, count - 1);
  return count;
}

/* granule counters: a byte each, four to a word */
__device__ __forceinline__ void cnt_adjust(uint32_t* cnt, uint32_t ga, uint32_t n, bool add)
{
  /* n = 1..3 consecutive granules from ga on: two word updates, the second of them often by zero (no branch on it) */
  const uint64_t inc = (uint64_t)(0x010101u >> (8u * (3u - n))) << (8u * (ga & 3u));
  const uint32_t lo = (uint32_t)inc, hi = (uint32_t)(inc >> 32);
  if (add) {
    wave::lds_add(cnt + (ga >> 2), lo);
    wave::lds_add(cnt + (ga >> 2) + 1, hi);
  } else {
    wave::lds_sub(cnt + (ga >> 2), lo);
    wave::lds_sub(cnt + (ga >> 2) + 1, hi);
  }
}

/* the whole wave: granules [ga, gb] of one long match */
__device__ __forceinline__ void cnt_adjust_range(uint32_t* cnt, uint32_t ga, uint32_t gb, bool add)
{
  const uint32_t lane = (uint32_t)wave::fresh_lane_id();
  for (uint32_t g = ga + lane; g <= gb; g += 64) {
    const uint32_t v = 1u << (8u * (g & 3u));
    if (add) {
      wave::lds_add(cnt + (g >> 2), v);
    } else {
      wave::lds_sub(cnt + (g >> 2), v);
    }
  }
}

/* the four counters from granule g on */
__device__ __forceinline__ uint32_t cnt_read4(const uint32_t* cnt, uint32_t g)
{
  const uint32_t a = cnt[g >> 2], b = cnt[(g >> 2) + 1];
  return wave::align_bytes(b, a, g & 3u);
}

/* A chunk that is not a team's (more than 64 KiB of capacity or stream, or a stream that breaks the in-place invariant):
 * waves 0 and 1 decode it as producer and consumer (lz4_decode_window.hip.h: pair) in the table area, the others wait.
 * `fallback(role, ...)`: role 0 produces, role 1 consumes and returns the bytes produced. */
template <class Fallback>
__device__ __forceinline__ uint32_t run_fallback(
    const Team& t, uint8_t* scratch, const uint8_t* in, uint32_t in_len, uint8_t* out, uint32_t out_cap, uint32_t& err, Fallback fallback)
{
  if (threadIdx.x < 4) {
    ((uint32_t*)(scratch + lzw::pair::kLdsPerChunk - lzw::pair::kCtrlBytes))[threadIdx.x] = 0; /* both slots empty, no abort */
  }
  __syncthreads();
  if (t.w < 2) {
    uint32_t e = lz::kErrNone;
    const uint32_t produced = fallback(t.w, in, in_len, out, out_cap, scratch, e);
    if (t.w == 1 && wave::lane_id() == 0) {
      t.ctl[kCtlErr] = e;
      t.ctl[kCtlTicket] = produced;
    }
  }
  __syncthreads();
  err = ctl_read(t, kCtlErr);
  const uint32_t produced = ctl_read(t, kCtlTicket);
  __syncthreads();
  return produced;
}

/*
 * Decode one chunk with the calling workgroup (64 x WAVES lanes, all of them call). `lds` = Geo<WAVES>::kLds bytes, 16-byte aligned.
 * FrontEnd supplies the format: kPositions, DeltaFn / SlowFn (the chase's distance functions) and parse_batch().
 * Returns the bytes produced; err != 0 on a malformed chunk. Always validates (a team never writes outside its buffer).
 */
template <class FrontEnd, uint32_t WAVES, class Fallback>
__device__ __forceinline__ uint32_t decode_chunk(
    const uint8_t* __restrict__ in, uint32_t in_len, uint8_t* out, uint32_t out_cap, uint8_t* lds, uint32_t& err, Fallback fallback)
{
  constexpr uint32_t kWaves = WAVES, kThreads = 64 * WAVES;
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = (uint32_t)wave::lane_id();
  Team t;
  t.buf = lds;
  t.w = wave::uniform(tid >> 6);
  t.tab = lds + kBufLds + t.w * lzw::kChaseLds;
  t.cnt = (uint32_t*)(lds + kBufLds + kWaves * lzw::kChaseLds);
  t.ctl = (uint32_t*)(lds + kBufLds + kWaves * lzw::kChaseLds + kCntLds);
  t.out = out;
  t.oa = (uint32_t)((uintptr_t)out & 15u);
  err = lz::kErrNone;
  if (in_len == 0) {
    err = FrontEnd::kEmptyIsError ? lz::kErrInput : lz::kErrNone;
    return 0;
  }
  uint8_t* fb_scratch = lds + kBufLds;
  if (out_cap > kMaxOut && in_len <= kMaxIn) {
    /* a generous capacity: what counts is what the stream decodes to, where it says so (Snappy's preamble) */
    const uint32_t declared = wave::uniform(FrontEnd::declared_length(in, in_len));
    if (declared <= kMaxOut) {
      out_cap = kMaxOut;
    }
  }
  /* A chunk that shrank sixteen times or more is a few hundred long matches (runs, sorted or typed columns), most of them
   * overlapping their own source: each takes the team a step of its own with its three barriers, and two waves as
   * producer and consumer decode such a chunk 20-27 % sooner (gpurun r5n, 256 / 512 chunks: sorted-key column 0.188 /
   * 0.206 ms against 0.158 / 0.157, int32 column 0.239 against 0.175; text the other way round, 0.143 against 0.256).
   * The ratio from which on that holds is the format's (FrontEnd::kFewLongMatchesRatio): a Snappy copy element carries at
   * most 64 bytes, so its streams of runs and columns stop at ratios of 13-21 where LZ4's reach 37-245. */
  const uint32_t declared_out = wave::uniform(FrontEnd::declared_length(in, in_len));
  const uint32_t expect_out = declared_out != ~0u && declared_out < out_cap ? declared_out : out_cap;
  const bool few_long_matches = (uint64_t)in_len * FrontEnd::kFewLongMatchesRatio <= expect_out;
  if (out_cap > kMaxOut || in_len > kMaxIn || few_long_matches) {
    /* not a team's chunk: the two-waves-per-chunk decoder, by waves 0 and 1 */
    return run_fallback(t, fb_scratch, in, in_len, out, out_cap, err, fallback);
  }
  const uint32_t ia = (uint32_t)((uintptr_t)in & 15u);
  t.st.base = in - ia;
  t.st.vbeg = ia;
  t.st.vend = ia + in_len;
  const uint32_t sb = (kBuf - t.st.vend) & ~15u; /* the stream ends at the end of the buffer, 16-byte congruent with HBM */
  t.st.ring = lds + sb;
  t.st.lo = 0;
  t.st.hi = t.st.vend + 16;
  const Stream& st = t.st;
  if (tid < kCtlWords) {
    t.ctl[tid] = 0;
  }
  for (uint32_t i = tid; i < kCntLds / 4; i += kThreads) {
    t.cnt[i] = 0; /* (every match that registers here leaves again: the counters are zero between steps) */
  }
  stage_stream<kThreads>(st, tid);
  __syncthreads();
  LZW_T(0);

  lzw::Chase c;
  c.tab = t.tab;
  c.nx01 = 0, c.nx23 = 0, c.wb = 0, c.q = 0;
  uint32_t q = st.vbeg; /* the next token */
  uint32_t limit = out_cap; /* the most the chunk may produce (Snappy: exactly what its preamble says) */
  if (!FrontEnd::begin(st, out_cap, q, limit, err)) {
    return 0; /* (uniform: every wave read the same preamble) */
  }
  uint32_t op = 0;      /* output produced (all of it final) */
  uint32_t flushed = 0; /* output written to HBM */
  bool give_up = false; /* in-place invariant broken: waves 0 and 1 redo the chunk as producer and consumer */
  typename FrontEnd::Delta delta;
  typename FrontEnd::Slow slow;
  /* ---- 1. tables of this wave's window of the step that starts at q_, speculated exit ---- */
  auto build_step = [&](uint32_t q_) {
    const uint32_t wb_ = q_ + FrontEnd::kPositions * t.w;
    if (wb_ < st.vend) {
      c.q = wb_;
      lzw::chase_build(c, st, delta, FrontEnd::kPositions);
      if (t.w + 1 < kWaves) { /* (nobody enters behind the last window) */
        const uint32_t x = exit_from(c, lane & (kEntries - 1));
        if (lane < kEntries) {
          t.ctl[kCtlSpec + kEntries * t.w + lane] = x;
        }
      }
    } else if (lane < kEntries) {
      t.ctl[kCtlSpec + kEntries * t.w + lane] = kUnknownExit;
    }
  };
  if (q < st.vend) {
    build_step(q);
  }
  LZW_T(1);
  __syncthreads();
  LZW_T(2);
  while (q < st.vend) {
    LZT_TR("w%u step q=%u op=%u\n", t.w, q, op);
    const uint32_t wb = q + FrontEnd::kPositions * t.w;
    const bool have_window = wb < st.vend;
    /* ---- 2. the entry of this wave's window, through the tabulated exits of the windows in front; enumerate, parse, sizes.
     * An entry at one of a window's first kEntries offsets has its exit in the table -- exact; one further in (a token
     * with a long literal run reached over from the window before) borrows offset 0's exit: chains merge within a few
     * tokens, and step 3 compares what every window was entered by with the true exit of the window in front. (One exit
     * per window, speculated from offset 0, was wrong for 2.2 % of the windows of text: a step of 16 windows in four
     * was cut short.) */
    uint32_t entry = q;
    if (t.w != 0) {
      /* the whole table in registers (four words a lane), the walk with v_readlane: a dependent LDS round trip per window
       * in front made the last wave of sixteen wait 3 000 cycles for its entry */
      const wave::u32x4 sp4 = *(const wave::u32x4*)(t.ctl + kCtlSpec + 4 * lane);
      for (uint32_t j = 0; j < t.w && entry != kUnknownExit; ++j) {
        const uint32_t off = entry - (q + FrontEnd::kPositions * j);
        if (off < FrontEnd::kPositions) { /* (else the chain jumps over window j altogether) */
          const uint32_t o = off < kEntries ? off : 0u;
          const uint32_t from = (kEntries * j + o) >> 2;
          const uint32_t x0 = wave::read_lane(sp4.x, from), x1 = wave::read_lane(sp4.y, from);
          const uint32_t x2 = wave::read_lane(sp4.z, from), x3 = wave::read_lane(sp4.w, from);
          entry = (o & 2u) ? ((o & 1u) ? x3 : x2) : ((o & 1u) ? x1 : x0);
        }
      }
    }
    uint32_t n = 0;
    uint32_t my_exit = entry;
    uint32_t seqpos = 0;
    lz::Seq s;
    s.lit_src = 0, s.lit_len = 0, s.match_off = 0, s.match_len = 0;
    bool bad = false;
    if (have_window && entry != kUnknownExit && entry - wb < FrontEnd::kPositions && entry < st.vend) {
      uint32_t last;
      n = enumerate(c, entry - wb, seqpos, last);
      my_exit = exit_of(c, st, last, false, slow);
      FrontEnd::parse_batch(st, seqpos, 0, n, s, bad);
    }
    const uint32_t len = s.lit_len + s.match_len;
    const uint32_t incl = wave::scan_add_inclusive(len);
    const uint32_t total = wave::read_lane(incl, 63);
    const uint64_t bad_lanes = wave::ballot(bad);
    LZW_T(3);
    LZT_TR("w%u B entry=%u n=%u exit=%u total=%u\n", t.w, entry, n, my_exit, total);
    if (lane == 0) {
      t.ctl[kCtlUsed + t.w] = entry;
      t.ctl[kCtlExit + t.w] = my_exit;
      t.ctl[kCtlBytes + t.w] = total;
      t.ctl[kCtlBad + t.w] = bad_lanes ? 1u : 0u;
    }
    __syncthreads();
    LZW_T(4);
    /* ---- 3. which windows stand, where their output goes; literals; pending matches registered ---- */
    /* lane j < kWaves looks at window j: it stands when every window in front of it stands and left through the exit
     * that window j was entered by */
    uint32_t sp = 0, ex = 0, by = 0, bd = 0;
    if (lane < kWaves) {
      sp = lane + 1 < kWaves ? t.ctl[kCtlUsed + lane + 1] : 0u; /* what the window behind was entered by */
      ex = t.ctl[kCtlExit + lane];
      by = t.ctl[kCtlBytes + lane];
      bd = t.ctl[kCtlBad + lane];
    }
    /* window j + 1 stands iff window j stands and its true exit is what j + 1 was entered by */
    const uint64_t chain_ok = wave::ballot(lane < kWaves && (lane + 1 == kWaves || ex == sp));
    const uint32_t broken = wave::ctz64(~chain_ok);            /* first window whose exit was not the speculated one */
    const uint32_t standing = broken + 1 < kWaves ? broken + 1 : kWaves; /* windows 0 .. standing - 1 stand */
    const bool stands = t.w < standing;
    const uint32_t by_ok = lane < standing ? by : 0u;
    const uint32_t by_incl = wave::scan_add_inclusive(by_ok);
    const uint32_t slot_end = op + by_incl; /* lane j: the end of slot j (slots behind `standing` are empty) */
    const uint32_t step_bytes = wave::read_lane(by_incl, 63);
    const uint32_t base = op + wave::read_lane(by_incl - by_ok, t.w);
    const uint32_t next_q = wave::read_lane(ex, standing - 1);
    const uint32_t any_bad = wave::ballot(lane < standing && bd != 0) ? 1u : 0u;
    const uint32_t step_end = op + step_bytes;
    if (any_bad || step_end > limit || next_q <= q) { /* (a step always consumes its first token) */
      err |= any_bad || next_q <= q ? lz::kErrInput : lz::kErrOutput;
      break;
    }
    if (t.w == 0) {
      LZ_STAT("team_steps", 1);
      LZ_STAT("team_windows_standing", standing);
      LZ_STAT("team_steps_cut_short", standing < kWaves && next_q < st.vend ? 1 : 0);
      LZ_STAT("team_gave_up", step_end + t.oa > sb + next_q ? 1 : 0);
      LZ_STAT("team_steps_literals_from_hbm", step_end + t.oa > sb + q ? 1 : 0);
    }
    if (step_end + t.oa > sb + next_q) { /* the output would reach into the stream the next step reads */
      give_up = true;
      break;
    }
    LZT_TR("w%u standing=%u base=%u step_end=%u next_q=%u\n", t.w, standing, base, step_end, next_q);
    const bool hazard = step_end + t.oa > sb + q; /* this step's output overwrites stream bytes its own literals come from */
    const uint32_t T0 = op & ~(kGran - 1);
    const uint32_t lit_dst = base + incl - len;
    const uint32_t match_dst = lit_dst + s.lit_len;
    const uint32_t my_lit = stands ? s.lit_len : 0u;
    uint32_t my_match = stands ? s.match_len : 0u;
    const uint32_t match_src = match_dst - s.match_off;
    const uint64_t bad_off = wave::ballot(my_match != 0 && s.match_off - 1 >= match_dst);
    if (bad_off && lane == 0) {
      t.ctl[kCtlErr] = lz::kErrOffset;
    }
    const bool short_match = my_match - 4 <= lzw::kMatchShort - 4;
    const bool lane_class = short_match && s.match_off >= 4;
    const uint64_t lane_class_mask = wave::ballot(lane_class);
    /* what must be final before this match may start: its source, up to its own first byte */
    const uint32_t need_end = match_src + my_match < match_dst ? match_src + my_match : match_dst;
    const uint32_t ga = (match_dst - T0) / kGran;
    uint32_t gb = (match_dst + my_match - 1 - T0) / kGran;
    gb = gb < kGranules ? gb : kGranules - 1;
    bool tracked = my_match != 0 && ga < kGranules;
    LZW_T(5);
    if (!bad_off) {
      /* literals */
      uint8_t* dst = t.buf + t.oa + lit_dst;
      const bool lit_own = my_lit - 1 < lzw::kLitShort;
      if (wave::ballot(lit_own && my_lit >= 4)) {
        const bool lit_lane = lit_own && my_lit >= 4;
        const uint32_t steps = lzw::steps_for(lit_lane, my_lit);
        if (lit_lane) {
          const uint32_t lastd = my_lit - 4;
          uint32_t data[8];
#pragma unroll
          for (uint32_t i = 0; i < 8; ++i) {
            if (i < steps) {
              const uint32_t o = 4 * i < lastd ? 4 * i : lastd;
              data[i] = hazard ? wave::gload_u32(st.base + s.lit_src + o) : lzw::ld32(st.ring + s.lit_src + o);
            }
          }
#pragma unroll
          for (uint32_t i = 0; i < 8; ++i) {
            if (i < steps) {
              const uint32_t o = 4 * i < lastd ? 4 * i : lastd;
              lz::st_u32(dst + o, data[i]);
            }
          }
        }
      }
      if (wave::ballot(lit_own && my_lit < 4)) {
        if (lit_own && my_lit < 4) {
          uint32_t b0, b1 = 0, b2 = 0;
          if (hazard) {
            b0 = wave::gload_u8(st.base + s.lit_src);
            b1 = my_lit > 1 ? wave::gload_u8(st.base + s.lit_src + 1) : 0u;
            b2 = my_lit > 2 ? wave::gload_u8(st.base + s.lit_src + 2) : 0u;
          } else {
            b0 = st.ring[s.lit_src];
            b1 = my_lit > 1 ? st.ring[s.lit_src + 1] : 0u;
            b2 = my_lit > 2 ? st.ring[s.lit_src + 2] : 0u;
          }
          dst[0] = (uint8_t)b0;
          if (my_lit > 1) {
            dst[1] = (uint8_t)b1;
          }
          if (my_lit > 2) {
            dst[2] = (uint8_t)b2;
          }
        }
      }
      for (uint64_t m = wave::ballot(my_lit != 0 && !lit_own); m; m &= m - 1) {
        const uint32_t j = wave::ctz64(m);
        /* long runs come from the chunk in HBM: nothing another wave writes can be in their way */
        lzw::copy_to_lds(t.buf + t.oa + wave::read_lane(lit_dst, j), st.base + wave::read_lane(s.lit_src, j), wave::read_lane(my_lit, j));
      }
      LZW_T(6);
      /* matches whose source lies in front of this step read final bytes: they go now, in front of the barrier, and
       * never enter the counters (three quarters of the matches of text) */
      {
        const bool early = my_match != 0 && need_end <= op;
        const bool go = early && lane_class;
        if (wave::ballot(go)) {
          const uint32_t steps = lzw::steps_for(go, my_match);
          if (go) {
            const uint8_t* src = t.buf + t.oa + match_src;
            uint8_t* d = t.buf + t.oa + match_dst;
            if (steps == 2) {
              lzw::copy_dwords_clamped<2>(d, src, my_match);
            } else if (steps == 4) {
              lzw::copy_dwords_clamped<4>(d, src, my_match);
            } else {
              lzw::copy_dwords_clamped<8>(d, src, my_match);
            }
          }
        }
        for (uint64_t m = wave::ballot(early && !lane_class); m; m &= m - 1) {
          const uint32_t j = wave::ctz64(m);
          lzw::lds_match_copy(t.buf + t.oa + wave::read_lane(match_dst, j), wave::read_lane(s.match_off, j), wave::read_lane(my_match, j));
        }
        my_match = early ? 0u : my_match;
      }
      /* the others are pending: their destination granules */
      tracked = tracked && my_match != 0;
      if (tracked && short_match) {
        cnt_adjust(t.cnt, ga, gb - ga + 1, true);
      }
      for (uint64_t m = wave::ballot(tracked && !short_match); m; m &= m - 1) {
        const uint32_t j = wave::ctz64(m);
        cnt_adjust_range(t.cnt, wave::read_lane(ga, j), wave::read_lane(gb, j), true);
      }
    }
    uint64_t pending = bad_off ? 0ull : wave::ballot(my_match != 0);
    const uint32_t my_end = wave::read_lane(slot_end, t.w);
    const uint32_t first_pending = pending ? wave::read_lane(match_dst, wave::ctz64(pending)) : my_end;
    wave::sync();
    if (lane == 0) {
      wave::lds_store_relaxed(t.ctl + kCtlProg + t.w, first_pending);
    }
    LZT_TR("w%u L pending=%llx\n", t.w, (unsigned long long)pending);
    LZW_T(7);
    __syncthreads();
    LZW_T(8);
    if (ctl_read(t, kCtlErr)) {
      err |= lz::kErrOffset;
      break;
    }
    /* ---- 4. the pending matches, as their sources become final ----
     * A lane-class match (4 .. 32 bytes, distance >= 4) whose source lies inside the counted range asks the counters of
     * its (at most three) source granules: zero -- apart from its own count where its destination starts in the granule
     * its source ends in -- means no pending match writes there any more. Everything else (long matches, periods below
     * 4, sources behind the counted range, granules shared with an unrelated pending match) waits for the step's
     * FRONTIER, the oldest pending match of the first slot that has one: every byte below it is final, and that match
     * itself is always free to go. The frontier costs two LDS round trips and is only worked out when the counters
     * let nothing through. */
    const bool in_track = lane_class && need_end > T0 && need_end <= T0 + kTrack;
    const uint32_t src0 = match_src > T0 ? match_src - T0 : 0u;
    const uint32_t sa = in_track ? src0 / kGran : 0u;
    const uint32_t sbg = in_track ? (need_end - 1 - T0) / kGran : 0u;
    const uint32_t ngran = sbg - sa + 1; /* 1 .. 3 */
    const uint32_t cmask = ngran >= 4 ? 0xffffffffu : (1u << (8u * ngran)) - 1u;
    const uint32_t cself = (tracked && sbg == ga) ? 1u << (8u * (sbg - sa)) : 0u;
    uint32_t spins = 0;
    while (pending) {
      pending = wave::uniform64(pending); /* (it is: spelled out for the compiler) */
      const bool mine = wave::lane_in(pending);
      bool go = mine && in_track && (cnt_read4(t.cnt, sa) & cmask) == cself;
      uint64_t gone = wave::ballot(go);
      /* (looking again a few times before working out the frontier -- two LDS reads and a ballot a look -- was measured
       * SLOWER: 72 -> 66 GB/s at 256 chunks; the polls compete with the waves that do the copying) */
      LZW_T(9);
      if (!gone) {
        const uint32_t f = wave::ctz64(pending);
        const uint32_t hw = wave::read_lane(match_dst, f);
        wave::sync();
        if (lane == 0) {
          wave::lds_store_relaxed(t.ctl + kCtlProg + t.w, hw);
        }
        uint32_t pg = 0xffffffffu;
        if (lane < kWaves) {
          pg = wave::lds_load_relaxed(t.ctl + kCtlProg + lane);
        }
        wave::sync();
        const uint64_t open = wave::ballot(lane < kWaves && pg < slot_end);
        const uint32_t frontier = open ? wave::read_lane(pg, wave::ctz64(open)) : step_end;
        if (!((lane_class_mask >> f) & 1)) {
          /* the wave's oldest pending match is long or has a period below 4: the whole wave copies it once everything
           * in front of it is final */
          const uint32_t f_need = wave::read_lane(need_end, f);
          if (f_need <= frontier || hw == frontier) {
            lzw::lds_match_copy(t.buf + t.oa + hw, wave::read_lane(s.match_off, f), wave::read_lane(my_match, f));
            if (wave::read_lane(tracked ? 1u : 0u, f)) {
              if (wave::read_lane(short_match ? 1u : 0u, f)) {
                if (lane == f) {
                  cnt_adjust(t.cnt, ga, gb - ga + 1, false);
                }
              } else {
                cnt_adjust_range(t.cnt, wave::read_lane(ga, f), wave::read_lane(gb, f), false);
              }
            }
            pending &= ~(1ull << f);
            LZW_T(11);
            continue;
          }
        }
        go = mine && lane_class && (need_end <= frontier || match_dst == frontier);
        gone = wave::ballot(go);
        if (!gone) {
          if (ctl_read(t, kCtlErr)) {
            break;
          }
          /* waiting for another wave's matches. The wait is bounded by construction (the step's oldest pending match is
           * always free to go); the counter turns a logic error into a failed chunk instead of a hung card */
          if (++spins > kSpinLimit) {
            if (lane == 0) {
              t.ctl[kCtlErr] = lz::kErrInput;
            }
            break;
          }
          wave::nap();
          LZW_T(12);
          continue;
        }
      }
      const uint32_t steps = lzw::steps_for(go, my_match);
      if (go) {
        const uint8_t* src = t.buf + t.oa + match_src;
        uint8_t* d = t.buf + t.oa + match_dst;
        if (steps == 2) {
          lzw::copy_dwords_clamped<2>(d, src, my_match);
        } else if (steps == 4) {
          lzw::copy_dwords_clamped<4>(d, src, my_match);
        } else {
          lzw::copy_dwords_clamped<8>(d, src, my_match);
        }
      }
      wave::sync();
      if (go && tracked) {
        cnt_adjust(t.cnt, ga, gb - ga + 1, false);
      }
      pending &= ~gone;
      LZW_T(10);
    }
    wave::sync();
    if (lane == 0) {
      wave::lds_store_relaxed(t.ctl + kCtlProg + t.w, my_end);
    }
    LZT_TR("w%u C\n", t.w);
    LZW_T(9);
    /* a wave that is through with its matches builds the tables of the NEXT step while the others finish theirs (they depend
     * on the stream only, and the stream from next_q on is out of this step's reach: the in-place test above); the barrier
     * that ends this step is the one the next step's enumeration waits for */
    if (next_q < st.vend) {
      build_step(next_q);
    }
    LZW_T(1);
    __syncthreads();
    LZW_T(13);
    if (ctl_read(t, kCtlErr)) { /* a wave gave the step up */
      err |= lz::kErrInput;
      break;
    }
    op = step_end;
    q = next_q;
    {
      /* HBM gets what the step finished: whole 16-byte blocks only, the rest waits */
      const uint32_t upto = ((op + t.oa) & ~15u) - t.oa;
      if ((op + t.oa) >= 16 && upto > flushed) {
        flush<kThreads>(t, flushed, upto, tid);
        flushed = upto;
      }
    }
  }
  /* every wave leaves the loop at the same point (the conditions are uniform across the team) */
  __syncthreads();
  if (give_up) {
    return run_fallback(t, fb_scratch, in, in_len, out, out_cap, err, fallback);
  }
  if (err) {
    return 0;
  }
  if (!FrontEnd::finish_ok(op, q, st.vend, limit)) {
    err |= lz::kErrInput;
    return 0;
  }
  /* (a last token that claims more than the chunk holds was refused by the parser; the chase itself may report
   * "behind the end" for a chunk's regular last sequence) */
  flush<kThreads>(t, flushed, op, tid);
  __syncthreads(); /* the buffer is free again */
  LZW_T(14);
  return op;
}

} // namespace lzt
