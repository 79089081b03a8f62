/*
 * common/lz_index.hip.h -- token indexer of the LZ decoders: ONE LANE PER CHUNK.
 *
 * The token chain of an LZ4 block / Snappy stream is serial per chunk (the position of
 * token k+1 follows from the lengths in token k), which is what a one-wave-per-chunk
 * decoder spends most of its instructions on (profiles/r01_final_pmc.json: 8.7 of 20
 * wave-instructions per sequence went into the speculative 256-position jump tables).
 * Across chunks the chains are independent, so here every LANE walks the chain of its own
 * chunk: one wave advances 64 chains by one token in ~50 instructions, and writes the
 * token positions (u16, "virtual" = relative to the chunk pointer rounded down to 16
 * bytes) to a table in the caller's temp buffer. The decode kernel proper
 * (lz4_decode_window.hip.h: decode_chunk_indexed) then reads 64 positions per batch with
 * one coalesced load instead of chasing them.
 *
 * Per lane, the compressed stream is staged through a 128-byte ring in LDS (dword-
 * interleaved across lanes: slot s of lane l sits at dword 64 s + l, so every access of
 * the wave is bank-conflict free whatever positions the lanes are at) that a 4-deep
 * pipeline of 16-byte global loads keeps ahead of the walk: the load issued in step k is
 * written to the ring in step k + 4, nothing ever waits for memory in the steady state and
 * a lane whose data has not arrived simply sits the step out.
 *
 * Temp layout (lzi::Layout): u32 count[N] | u16 table[N][stride]. count = kNotIndexed
 * marks a chunk this path does not take (longer than 65535 bytes, or more tokens than
 * the row holds); the caller then runs the one-wave-per-chunk chase decoder on it.
 */
#pragma once

#include "common/wave.h"

namespace lzi {

constexpr uint32_t kNotIndexed = 0xffffffffu;
constexpr uint32_t kMaxInput = 65535;   /* token positions are stored as u16 */
constexpr uint32_t kRingDwords = 32;    /* 128 bytes of stream per lane */
constexpr uint32_t kRingBytes = 4 * kRingDwords;
constexpr uint32_t kPipe = 4;           /* 16-byte loads in flight per lane */
constexpr uint32_t kTokSlots = 16;      /* 32 u16 entries per lane awaiting their 16-byte store */
constexpr uint32_t kLdsDwords = 64 * (kRingDwords + kTokSlots);

/* Where the index lives inside the caller's temp buffer. */
struct Layout
{
  uint32_t* counts; /* [batch] */
  uint16_t* table;  /* [batch][stride] */
  uint32_t stride;  /* u16 entries per row, multiple of 8; 0 = temp buffer too small, path unused */
};

inline size_t header_bytes(size_t batch)
{
  return (batch * 4 + 15) & ~(size_t)15;
}

/* Rows hold what the worst stream of the largest eligible chunk can contain: a token every 3 bytes. */
inline uint32_t stride_for(size_t max_compressed_chunk_bytes)
{
  const size_t in = max_compressed_chunk_bytes < kMaxInput ? max_compressed_chunk_bytes : kMaxInput;
  return (uint32_t)((in / 3 + 2 + 7) & ~(size_t)7);
}

inline size_t temp_bytes_for(size_t batch, size_t max_compressed_chunk_bytes)
{
  return 16 + header_bytes(batch) + batch * (size_t)stride_for(max_compressed_chunk_bytes) * 2;
}

/* Carve the layout out of (temp, bytes); stride 0 when the buffer cannot hold a useful table. */
inline Layout carve(void* temp, size_t bytes, size_t batch)
{
  Layout l = {nullptr, nullptr, 0};
  if (temp == nullptr || batch == 0) {
    return l;
  }
  const uintptr_t a = ((uintptr_t)temp + 15) & ~(uintptr_t)15;
  const size_t skip = a - (uintptr_t)temp;
  const size_t head = header_bytes(batch);
  if (bytes < skip + head + batch * 128) {
    return l;
  }
  size_t per = (bytes - skip - head) / batch / 16 * 8; /* u16 entries, multiple of 8 */
  const size_t most = stride_for(kMaxInput);
  l.counts = (uint32_t*)a;
  l.table = (uint16_t*)(a + head);
  l.stride = (uint32_t)(per < most ? per : most);
  return l;
}

/* 128-bit value moved c (0..15) bytes towards lower / higher byte positions, zero filled. */
__device__ __forceinline__ wave::u32x4 bytes_down(wave::u32x4 x, uint32_t c)
{
  uint32_t e0 = x.x, e1 = x.y, e2 = x.z, e3 = x.w;
  if (c & 4) {
    e0 = e1, e1 = e2, e2 = e3, e3 = 0;
  }
  if (c & 8) {
    e0 = e2, e1 = e3, e2 = 0, e3 = 0;
  }
  const uint32_t b = c & 3u;
  wave::u32x4 r = {wave::align_bytes(e1, e0, b), wave::align_bytes(e2, e1, b), wave::align_bytes(e3, e2, b),
                   wave::align_bytes(0u, e3, b)};
  return r;
}
__device__ __forceinline__ wave::u32x4 bytes_up(wave::u32x4 x, uint32_t c)
{
  uint32_t e0 = x.x, e1 = x.y, e2 = x.z, e3 = x.w;
  if (c & 4) {
    e3 = e2, e2 = e1, e1 = e0, e0 = 0;
  }
  if (c & 8) {
    e3 = e1, e2 = e0, e1 = 0, e0 = 0;
  }
  const uint32_t b = c & 3u;
  if (b) {
    const uint32_t s = 4u - b;
    e3 = wave::align_bytes(e3, e2, s);
    e2 = wave::align_bytes(e2, e1, s);
    e1 = wave::align_bytes(e1, e0, s);
    e0 = wave::align_bytes(e0, 0u, s);
  }
  wave::u32x4 r = {e0, e1, e2, e3};
  return r;
}

/* One lane's view of its chunk. Positions are virtual: byte i of the chunk is vbeg + i. */
struct Stream
{
  const uint8_t* base; /* chunk pointer rounded down to 16 */
  uint32_t vbeg;       /* chunk & 15 */
  uint32_t vend;       /* vbeg + length */
  uint32_t lim;        /* 16-byte pieces below lim have been requested */
  uint32_t wlim;       /* ... below wlim are in the ring (contiguous from the last restart) */
};

/* The 16 stream bytes at virtual position a (multiple of 16, piece intersects the chunk, chunk >= 16 bytes):
 * bytes outside the chunk are never read -- a piece cut by the chunk's first / last byte is loaded from the
 * nearest position that lies inside and moved into place (`cut` != 0: adjust() below). */
__device__ __forceinline__ const uint8_t* piece_address(const Stream& s, uint32_t a, int32_t& cut)
{
  const uint32_t lo_cut = a < s.vbeg ? s.vbeg - a : 0u;
  const uint32_t hi_cut = a + 16 > s.vend ? a + 16 - s.vend : 0u;
  cut = (int32_t)lo_cut - (int32_t)hi_cut;
  return s.base + a + cut;
}
__device__ __forceinline__ wave::u32x4 adjust(wave::u32x4 x, int32_t cut)
{
  if (wave::ballot(cut != 0)) { /* wave-uniform: the common step has no cut piece in any lane */
    const wave::u32x4 up = bytes_up(x, (uint32_t)cut & 15u);
    const wave::u32x4 down = bytes_down(x, (uint32_t)(-cut) & 15u);
    x = cut > 0 ? up : cut < 0 ? down : x;
  }
  return x;
}

__device__ __forceinline__ uint32_t ring_slot(uint32_t vpos)
{
  return ((vpos >> 2) & (kRingDwords - 1)) * 64u;
}

/* The 8 stream bytes at p (those below wlim, or everything when the whole chunk is resident). */
__device__ __forceinline__ uint64_t ring_read8(const uint32_t* ring, uint32_t lane, uint32_t p)
{
  const uint32_t d0 = ring[ring_slot(p) + lane];
  const uint32_t d1 = ring[ring_slot(p + 4) + lane];
  const uint32_t d2 = ring[ring_slot(p + 8) + lane];
  const uint32_t lo = wave::align_bytes(d1, d0, p & 3u);
  const uint32_t hi = wave::align_bytes(d2, d1, p & 3u);
  return ((uint64_t)hi << 32) | lo;
}

/* Index of the first byte of w that is not 255 (8 when all are). */
__device__ __forceinline__ uint32_t first_not_255(uint64_t w)
{
  const uint64_t x = ~w;
  return x ? (uint32_t)__builtin_ctzll(x) >> 3 : 8u;
}

/*
 * The walk. `Format` supplies
 *   struct State;                                    per-lane walker state
 *   static void start(State&, const Stream&, ...)    (may consume a preamble: returns first position)
 *   static bool step(State&, uint32_t& p, uint64_t w, uint32_t vend, bool any_slow)
 *       one step at position p over the 8 bytes w: advances p, returns true when a token starts at the old p
 *       (any_slow: some lane of the wave reports in_length_bytes(), the uncommon states).
 * A lane finishes when p >= vend.
 */
template <class Format>
__device__ __forceinline__ void index_chunks(
    const void* const* __restrict__ comp_ptrs,
    const size_t* __restrict__ comp_bytes,
    size_t batch_size,
    Layout lay,
    uint32_t* lds /* kLdsDwords of this wave */)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  const size_t chunk = (size_t)blockIdx.x * 64 + lane;
  const bool have = chunk < batch_size;
  uint32_t* ring = lds;
  uint16_t* tokbuf = (uint16_t*)(lds + 64 * kRingDwords);
  const uint8_t* safe = (const uint8_t*)lay.counts; /* 16 readable bytes for the loads of lanes that request nothing */

  Stream s;
  const uint8_t* in = have ? (const uint8_t*)comp_ptrs[chunk] : nullptr;
  const size_t len64 = have ? comp_bytes[chunk] : 0;
  const bool eligible = have && len64 <= kMaxInput;
  const uint32_t len = eligible ? (uint32_t)len64 : 0u;
  s.vbeg = (uint32_t)((uintptr_t)in & 15u);
  s.base = in - s.vbeg;
  s.vend = s.vbeg + len;
  s.lim = 0;
  s.wlim = 0;
  const uint32_t vend_up = (s.vend + 15u) & ~15u;
  bool running = eligible && len != 0;

  /* chunks shorter than 16 bytes go into the ring whole, byte by byte (no 16-byte load fits inside them) */
  if (wave::ballot(running && len < 16)) {
    const bool tiny = running && len < 16;
    for (uint32_t j = 0; j < 8; ++j) { /* vbeg + len < 31: at most 8 dwords */
      uint32_t v = 0;
      for (uint32_t b = 0; b < 4; ++b) {
        const uint32_t q = 4 * j + b;
        if (tiny && q >= s.vbeg && q < s.vend) {
          v |= wave::gload_u8(s.base + q) << (8 * b);
        }
      }
      if (tiny) {
        ring[ring_slot(4 * j) + lane] = v;
      }
    }
    if (tiny) {
      s.lim = vend_up;
      s.wlim = vend_up;
    }
  }

  typename Format::State st;
  uint32_t p = s.vbeg;
  Format::start(st);
  uint32_t n = 0;       /* tokens found */
  uint32_t flushed = 0; /* tokens already in the table (multiple of 8) */
  bool overflow = false;
  uint16_t* row = lay.table + (have ? chunk : 0) * (size_t)lay.stride;

  wave::u32x4 data[kPipe];
  uint32_t dpos[kPipe];
  int32_t dcut[kPipe];
  bool dval[kPipe];
#pragma unroll
  for (uint32_t i = 0; i < kPipe; ++i) {
    data[i] = wave::u32x4{0, 0, 0, 0};
    dpos[i] = 0;
    dcut[i] = 0;
    dval[i] = false;
  }

  while (wave::ballot(running)) {
#pragma unroll
    for (uint32_t round = 0; round < 4; ++round) {
#pragma unroll
      for (uint32_t i = 0; i < kPipe; ++i) {
        /* (1) the load issued kPipe steps ago lands in the ring (stale ones, from before a restart, are dropped) */
        {
          const wave::u32x4 x = adjust(data[i], dval[i] ? dcut[i] : 0);
          const bool land = dval[i] && dpos[i] == s.wlim;
          if (land) {
            const uint32_t a = dpos[i];
            ring[ring_slot(a) + lane] = x.x;
            ring[ring_slot(a + 4) + lane] = x.y;
            ring[ring_slot(a + 8) + lane] = x.z;
            ring[ring_slot(a + 12) + lane] = x.w;
            s.wlim += 16;
          }
        }
        /* (2) request the next piece while the ring has room; every lane issues the load (from `safe` when it
         * wants nothing), so the number of loads in flight is the same on every path */
        {
          const bool want = running && s.lim < vend_up && s.lim - (p & ~15u) < kRingBytes;
          int32_t cut = 0;
          const uint8_t* addr = safe;
          if (want) {
            addr = piece_address(s, s.lim, cut);
          }
          data[i] = wave::gload_u32x4(addr);
          dpos[i] = s.lim;
          dcut[i] = cut;
          dval[i] = want;
          s.lim += want ? 16u : 0u;
        }
        /* (3) one step of the walk for the lanes whose bytes are there */
        {
          const bool ready = running && (p + 8 <= s.wlim || s.wlim >= vend_up);
          wave::sync(); /* ring writes of (1) before the reads */
          const uint64_t w = ring_read8(ring, lane, p);
          const bool any_slow = wave::ballot(ready && Format::in_length_bytes(st)) != 0;
          if (ready) {
            const uint32_t at = p;
            const bool token = Format::step(st, p, w, s.vend, any_slow);
            const bool fits = n < lay.stride;
            if (token && fits) {
              tokbuf[(((n & 31u) >> 1) * 64u + lane) * 2u + (n & 1u)] = (uint16_t)at;
            }
            n += token && fits ? 1u : 0u;
            overflow = overflow || (token && !fits);
            running = p < s.vend && !overflow;
            const bool restart = p >= s.lim; /* jumped over everything requested (long literal run): restart the stream there */
            s.lim = restart ? p & ~15u : s.lim;
            s.wlim = restart ? s.lim : s.wlim;
          }
        }
      }
    }
    /* every 16 steps: complete groups of 8 positions go to the table as one 16-byte store per lane */
    wave::sync();
#pragma unroll
    for (uint32_t it = 0; it < 2; ++it) {
      if (n - flushed >= 8) {
        const uint32_t* tb = (const uint32_t*)tokbuf;
        const uint32_t slot = (flushed & 31u) >> 1;
        wave::u32x4 v = {tb[(slot + 0) * 64 + lane], tb[(slot + 1) * 64 + lane], tb[(slot + 2) * 64 + lane],
                         tb[(slot + 3) * 64 + lane]};
        wave::gstore_u32x4_aligned((uint8_t*)(row + flushed), v);
        flushed += 8;
      }
    }
  }

  /* the rest: up to 23 positions, the last group padded (rows are multiples of 8 entries long) */
  wave::sync();
#pragma unroll
  for (uint32_t it = 0; it < 3; ++it) {
    if (have && !overflow && flushed < n) {
      const uint32_t* tb = (const uint32_t*)tokbuf;
      const uint32_t slot = (flushed & 31u) >> 1;
      wave::u32x4 v = {tb[(slot + 0) * 64 + lane], tb[(slot + 1) * 64 + lane], tb[(slot + 2) * 64 + lane],
                       tb[(slot + 3) * 64 + lane]};
      wave::gstore_u32x4_aligned((uint8_t*)(row + flushed), v);
      flushed += 8;
    }
  }
  if (have) {
    lay.counts[chunk] = (!eligible || overflow) ? kNotIndexed : n;
  }
}

} // namespace lzi
