/*
 * common/lz_match.hip.h -- the wave-parallel greedy match finder shared by the
 * LZ4 and Snappy compressors (one wavefront per chunk).
 *
 * Each step the 64 lanes hash the 4-byte words at 64 consecutive positions and
 * probe a per-wave hash table in LDS (2-byte entries holding position mod
 * 65536); the first lane whose candidate really matches wins, the wave measures
 * the match length cooperatively (64 bytes per ballot), the format's emitter
 * writes one literal-run + match, and the positions the step consumed are
 * inserted into the table. Greedy, single probe: the ratio class of the CPU
 * "fast" compressors. Match distances are limited to 65535 (both formats'
 * 2-byte offset forms).
 */
#pragma once

#include "common/lz_common.hip.h"

namespace lzm {

constexpr uint32_t kHashBits = 12;
constexpr uint32_t kHashSize = 1u << kHashBits;
constexpr uint32_t kMinMatch = 4;

__device__ __forceinline__ uint32_t hash4(uint32_t v)
{
  return (v * 2654435761u) >> (32 - kHashBits);
}

/*
 * Emitter concept:
 *   uint32_t Emitter::match(uint8_t* dst, const uint8_t* lit, uint32_t lit_len, uint32_t offset, uint32_t match_len)
 *   uint32_t Emitter::tail(uint8_t* dst, const uint8_t* lit, uint32_t lit_len)
 * both return the number of bytes written (called by the whole wave).
 *
 * last_start : a match may start at positions <= last_start (needs n >= 4 readable bytes there)
 * match_end  : a match may not extend past this position
 */
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
        /* nearest position below pos whose low 16 bits are `low` */
        cand = (pos & ~0xffffu) | low;
        if (cand >= pos) {
          cand -= 0x10000u;
        }
      }
      if (eligible && cand < pos && pos - cand <= 65535u) { /* cand wraps to a huge value when there is none */
        found = lz::ld_u32(src + cand) == word;
      }
      /* The table only knows positions of earlier steps. Repeats inside the
       * window at the distances typed columns produce (1, 2, 4, 8 bytes) are
       * caught by comparing against the neighbouring lanes' words. */
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
      const uint32_t f = wave::ctz64(hits);
      const uint32_t mpos = ip + f;
      const uint32_t mcand = wave::read_lane(cand, f);
      /* cooperative match-length measurement, 64 bytes per step */
      uint32_t mlen = kMinMatch;
      for (;;) {
        const uint32_t p = mpos + mlen + lane;
        const bool same = p < match_end && src[p] == src[mcand + mlen + lane];
        const uint64_t diff = ~wave::ballot(same);
        if (diff != 0) {
          mlen += wave::ctz64(diff);
          break;
        }
        mlen += 64;
      }
      /* insert only the positions this step consumes; the rest of the window is
       * probed again by the next step and must still see its older candidates */
      if (eligible && pos < mpos + mlen) {
        table[hash4(word)] = (uint16_t)pos;
      }
      wave::sync();
      op += Emitter::match(dst + op, src + anchor, mpos - anchor, mpos - mcand, mlen);
      ip = mpos + mlen;
      anchor = ip;
    }
  }
  op += Emitter::tail(dst + op, src + anchor, n - anchor);
  return op;
}

} // namespace lzm
