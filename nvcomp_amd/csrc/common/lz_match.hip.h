/*
 * common/lz_match.hip.h -- the wave-parallel greedy match finder shared by the
 * LZ4 and Snappy compressors (one wavefront per chunk).
 *
 * Each step looks at 64 consecutive input positions, one per lane:
 *   1. hash the 4-byte word, probe a per-wave LDS hash table (2-byte entries holding
 *      position mod 65536), verify the candidate; repeats inside the window at the
 *      distances typed columns produce (1, 2, 4, 8 bytes) are found by comparing with
 *      the neighbouring lanes' words;
 *   2. every lane with a verified candidate measures its own match, up to kLaneCap bytes;
 *   3. a scalar walk picks the non-overlapping matches greedily in position order
 *      (a match that hit the cap becomes the last of the step and is extended by the
 *      whole wave, 64 bytes per ballot);
 *   4. a DPP prefix sum over the sequence sizes gives every selected lane its output
 *      offset and the lanes emit their sequences together (literal run + match); only
 *      a sequence with a long literal run or a long match is written cooperatively;
 *   5. the positions the step consumed are inserted into the hash table.
 * Greedy, single probe: the ratio class of the CPU "fast" compressors. Match distances
 * are limited to 65535 (both formats' 2-byte offset forms).
 *
 * Emitter concept (all static):
 *   uint32_t seq_size(lit_len, match_len, offset)                 bytes one sequence takes
 *   bool     is_small(lit_len, match_len)                         can a single lane write it?
 *   void     emit_small_header(dst, lit_len, offset, match_len)   per lane; literals are copied by the caller
 *   uint32_t lit_offset(lit_len)                                  where the literal bytes start inside the sequence
 *   uint32_t match(dst, lit, lit_len, offset, match_len)          whole wave, any size; returns bytes written
 *   uint32_t tail(dst, lit, lit_len)                              whole wave: the final literal-only part
 */
#pragma once

#include "common/lz_common.hip.h"

namespace lzm {

#ifndef NVCOMP_LZM_HASH_BITS
#define NVCOMP_LZM_HASH_BITS 12
#endif
#ifndef NVCOMP_LZM_WAVES_PER_SIMD
#define NVCOMP_LZM_WAVES_PER_SIMD 5 /* what the 8 KiB hash table per wave allows (4-wave workgroups, 160 KB LDS per CU) */
#endif
constexpr uint32_t kHashBits = NVCOMP_LZM_HASH_BITS;
/* Entries of the per-wave table. A power of two by default; -DNVCOMP_LZM_HASH_ENTRIES=3072 (a multiple of 128) gives
 * the 6 KiB table that fits 6 waves/SIMD (with -DNVCOMP_LZM_WAVES_PER_SIMD=6) -- an A/B build, not yet measured. */
#ifdef NVCOMP_LZM_HASH_ENTRIES
constexpr uint32_t kHashSize = NVCOMP_LZM_HASH_ENTRIES;
#else
constexpr uint32_t kHashSize = 1u << kHashBits;
#endif
static_assert(kHashSize % 128 == 0 && kHashSize <= 65536, "the table is cleared 64 dwords at a time");
constexpr uint32_t kMinMatch = 4;
constexpr uint32_t kLaneCap = 36; /* per-lane match measurement: 4 + 8 dword compares */

__device__ __forceinline__ uint32_t hash4(uint32_t v)
{
  const uint32_t h = v * 2654435761u;
  if constexpr ((kHashSize & (kHashSize - 1)) == 0) {
    return h >> (32 - kHashBits);
  } else {
    return __umulhi(h, kHashSize); /* multiply-shift range reduction onto [0, kHashSize) */
  }
}

/* Wave-cooperative extension of a match known to be at least `have` bytes long. */
__device__ __forceinline__ uint32_t extend_match(
    const uint8_t* __restrict__ src, uint32_t mpos, uint32_t mcand, uint32_t have, uint32_t match_end)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  uint32_t mlen = have;
  for (;;) {
    const uint32_t p = mpos + mlen + lane;
    const bool same = p < match_end && src[p] == src[mcand + mlen + lane];
    const uint64_t diff = ~wave::ballot(same);
    if (diff != 0) {
      return mlen + wave::ctz64(diff);
    }
    mlen += 64;
  }
}

template <class Emitter>
__device__ __forceinline__ uint32_t encode_chunk(
    const uint8_t* __restrict__ src, uint32_t n, uint8_t* dst, uint16_t* table, uint32_t last_start,
    uint32_t match_end, bool any_match)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  for (uint32_t i = lane; i < kHashSize / 2; i += 64) {
    ((uint32_t*)table)[i] = 0;
  }
  wave::sync();

  uint32_t op = 0;
  uint32_t anchor = 0;
  if (any_match) {
    uint32_t ip = 0;
    while (ip <= last_start) {
      const uint32_t pos = ip + lane;
      const bool eligible = pos <= last_start;
      uint32_t word = 0, cand = 0;
      bool found = false;
      if (eligible) {
        word = lz::ld_u32(src + pos);
        const uint32_t low = table[hash4(word)];
        cand = (pos & ~0xffffu) | low; /* nearest position below pos whose low 16 bits are `low` */
        if (cand >= pos) {
          cand -= 0x10000u;
        }
      }
      if (eligible && cand < pos && pos - cand <= 65535u) { /* cand wraps to a huge value when there is none */
        found = lz::ld_u32(src + cand) == word;
      }
      for (uint32_t d = 1; d <= 8; d *= 2) {
        const uint32_t other = wave::shuffle(word, (lane - d) & 63u);
        if (eligible && !found && lane >= d && other == word) {
          found = true;
          cand = pos - d;
        }
      }
      const uint64_t hits = wave::ballot(found);
      if (hits == 0) {
        /* no match in this window: its 64 positions become literals; remember them */
        if (eligible) {
          table[hash4(word)] = (uint16_t)pos;
        }
        wave::sync();
        ip += 64;
        continue;
      }

      /* ---- long first match (runs, periodic columns): one cooperative probe decides ---- */
      {
        const uint32_t f0 = wave::ctz64(hits);
        const uint32_t mpos = ip + f0;
        const uint32_t mcand = wave::read_lane(cand, f0);
        const uint32_t p = mpos + kMinMatch + lane;
        const bool same = p < match_end && src[p] == src[mcand + kMinMatch + lane];
        const uint64_t diff = ~wave::ballot(same);
        const uint32_t len0 = diff ? kMinMatch + wave::ctz64(diff) : extend_match(src, mpos, mcand, kMinMatch + 64, match_end);
        if (len0 >= kLaneCap) {
          op += Emitter::match(dst + op, src + anchor, mpos - anchor, mpos - mcand, len0);
          const uint32_t next = mpos + len0;
          if (eligible && pos < next) {
            table[hash4(word)] = (uint16_t)pos;
          }
          wave::sync();
          anchor = next;
          ip = next;
          continue;
        }
      }

      /* ---- every hit lane measures its own match (16-byte compares, capped): the candidate side of these
       * loads is scattered, one L1 tag lookup per lane and load, so fewer and wider loads is what counts ---- */
      uint32_t mlen = 0;
      if (found) {
        const uint32_t room = match_end - pos; /* >= 4 for an eligible position */
        uint32_t cap = room < kLaneCap ? room : kLaneCap;
        mlen = kMinMatch;
        while (mlen + 16 <= cap) {
          const wave::u32x4 a = lz::ld_u32x4(src + pos + mlen);
          const wave::u32x4 b = lz::ld_u32x4(src + cand + mlen);
          const uint32_t x0 = a.x ^ b.x, x1 = a.y ^ b.y, x2 = a.z ^ b.z, x3 = a.w ^ b.w;
          if ((x0 | x1 | x2 | x3) != 0) {
            const uint32_t first = x0 ? 0u : x1 ? 1u : x2 ? 2u : 3u;
            const uint32_t x = x0 ? x0 : x1 ? x1 : x2 ? x2 : x3;
            mlen += 4 * first + ((uint32_t)__builtin_ctz(x) >> 3);
            cap = 0; /* stop */
            break;
          }
          mlen += 16;
        }
        while (mlen + 4 <= cap) {
          const uint32_t x = lz::ld_u32(src + pos + mlen) ^ lz::ld_u32(src + cand + mlen);
          if (x != 0) {
            mlen += (uint32_t)__builtin_ctz(x) >> 3;
            cap = 0;
            break;
          }
          mlen += 4;
        }
        while (mlen < cap && src[pos + mlen] == src[cand + mlen]) { /* at most 3 tail bytes */
          ++mlen;
        }
      }

      /* ---- greedy selection in position order (scalar walk over the hit mask) ---- */
      uint32_t prev_end = 0;  /* per selected lane: where its literal run starts */
      uint64_t selected = 0;
      uint32_t cur = 0;       /* window-relative position the next match may start at */
      uint32_t lit_from = anchor;
      uint32_t last = 64;     /* lane of the selected match that hit the cap, if any */
      uint64_t rest = hits;
      while (rest) {
        const uint32_t f = wave::ctz64(rest);
        const uint32_t flen = wave::read_lane(mlen, f);
        prev_end = wave::write_lane(prev_end, lit_from, f);
        selected |= 1ull << f;
        cur = f + flen;
        if (flen >= kLaneCap) { /* the capped match may be much longer: measure it with the whole wave */
          const uint32_t full = extend_match(src, ip + f, wave::read_lane(cand, f), kLaneCap, match_end);
          mlen = wave::write_lane(mlen, full, f);
          cur = f + full;
        }
        lit_from = ip + cur;
        rest = cur < 64 ? (hits & (~0ull << cur)) : 0ull;
      }
      (void)last;
      const uint32_t next_ip = cur > 64 ? ip + cur : ip + 64;

      /* ---- emit the selected sequences ---- */
      const bool sel = (selected >> lane) & 1;
      const uint32_t lit_len = sel ? pos - prev_end : 0;
      const uint32_t my_len = sel ? mlen : 0;
      const uint32_t offset = pos - cand;
      const uint32_t size = sel ? Emitter::seq_size(lit_len, my_len, offset) : 0;
      const uint32_t incl = wave::scan_add_inclusive(size);
      const uint32_t total = wave::read_lane(incl, 63);
      uint8_t* my_dst = dst + op + incl - size;
      const bool small = sel && Emitter::is_small(lit_len, my_len);
      if (small) {
        Emitter::emit_small_header(my_dst, lit_len, offset, my_len);
      }
      /* literal runs of the small sequences: 4-byte steps, offsets clamped to len-4 */
      {
        const uint8_t* ls = src + prev_end;
        uint8_t* ld = my_dst + Emitter::lit_offset(lit_len);
        const bool lit4 = small && lit_len >= 4;
        uint32_t steps = 0;
        if (wave::ballot(lit4)) {
          steps = wave::reduce_max(lit4 ? (lit_len + 3) / 4 : 0u);
        }
        if (lit4) {
          const uint32_t lastoff = lit_len - 4;
          for (uint32_t i = 0; i < steps; ++i) {
            const uint32_t o = 4 * i < lastoff ? 4 * i : lastoff;
            lz::st_u32(ld + o, lz::ld_u32(ls + o));
          }
        }
        if (small && lit_len != 0 && lit_len < 4) {
          ld[0] = ls[0];
          if (lit_len > 1) {
            ld[1] = ls[1];
          }
          if (lit_len > 2) {
            ld[2] = ls[2];
          }
        }
      }
      /* the few sequences a single lane cannot write: long literal run (first of the step,
       * after match-less windows) or long match (last of the step) */
      uint64_t big = wave::ballot(sel && !small);
      while (big) {
        const uint32_t j = wave::ctz64(big);
        big &= big - 1;
        const uint32_t jdst = op + wave::read_lane(incl - size, j);
        const uint32_t jlit = wave::read_lane(prev_end, j);
        const uint32_t jpos = ip + j;
        Emitter::match(dst + jdst, src + jlit, jpos - jlit, wave::read_lane(offset, j), wave::read_lane(my_len, j));
      }
      op += total;

      /* insert only the positions this step consumes; the rest of the window is probed again
       * by the next step and must still see its older candidates */
      if (eligible && pos < next_ip) {
        table[hash4(word)] = (uint16_t)pos;
      }
      wave::sync();
      anchor = lit_from;
      ip = next_ip;
    }
  }
  op += Emitter::tail(dst + op, src + anchor, n - anchor);
  return op;
}

} // namespace lzm
