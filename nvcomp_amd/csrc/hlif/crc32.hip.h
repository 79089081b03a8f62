/*
 * hlif/crc32.hip.h -- CRC-32 (IEEE 802.3, reflected: what zlib's crc32() and boost::crc_32_type compute,
 * examples/standard_crc_checksum.cu:94-107 of the reference) of one chunk by one wavefront.
 *
 * A CRC is a chain over the bytes, so the chunk is cut for the lanes and the pieces are combined with the algebra of
 * the checksum (the remainder of the message polynomial modulo P is linear in the message):
 *
 *   - the message, padded IN FRONT with zero bytes to a multiple of kTile = 64 kSeg = 2 KiB, is walked tile by tile; lane l
 *     owns the kSeg = 32 bytes [32 l, 32 l + 32) of every tile: two 16-byte loads per lane and tile (kSeg = 64, four loads
 *     that each touch all 64 lines of a 4 KiB tile, is as fast on 64 KiB chunks and 18 % slower on compressed chunks of
 *     any size and alignment; kSeg = 16, fully coalesced, pays a skip per 16 bytes: scripts/probes/crc_bench.hip);
 *   - inside its 64 bytes a lane runs slicing-by-4 (one table lookup per byte, four per dword);
 *   - between two tiles a lane's state skips the 2 016 bytes that belong to the other lanes: appending Z zero bytes to a
 *     raw remainder is the linear map s -> s * x^(8Z) mod P, done as four more table lookups (one per state byte);
 *   - leading zeros do not change a raw remainder, so the front padding costs nothing, and the lane that holds the first
 *     real byte starts from the 0xffffffff every CRC-32 starts from; the padding makes every lane's distance to the END
 *     of the message the same for every chunk length, (63 - l) * kSeg bytes: six conditional multiplications by constants;
 *   - an XOR across the lanes and the final complement give the checksum.
 *
 * The tables (8 KiB: slicing and skipping) and the six constants are computed at COMPILE time (constexpr) and copied to
 * LDS once per workgroup. Rounds 1-2 walked a chunk with ONE THREAD, a dependent table lookup per byte.
 */
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common/wave.h"

namespace crc32w {

constexpr uint32_t kPoly = 0xedb88320u;
#ifndef NVCOMP_CRC_SEG
#define NVCOMP_CRC_SEG 32
#endif
constexpr uint32_t kSegDefault = NVCOMP_CRC_SEG; /* bytes of a tile one lane owns: 16, 32 or 64 (a tile = 64 of them) */

/* a(x) * b(x) mod P(x) in the reflected representation (bit 31 is x^0) */
constexpr uint32_t mulmod(uint32_t a, uint32_t b)
{
  uint32_t p = 0;
  for (uint32_t m = 0x80000000u; m != 0; m >>= 1) {
    if (a & m) {
      p ^= b;
    }
    b = (b & 1u) ? (b >> 1) ^ kPoly : b >> 1;
  }
  return p;
}

/* x^(8 n) mod P: what appending n zero bytes multiplies a raw remainder by */
constexpr uint32_t xpow_bytes(uint64_t n)
{
  uint32_t r = 0x80000000u;    /* x^0 */
  uint32_t sq = 0x00800000u;   /* x^8 */
  for (; n != 0; n >>= 1) {
    if (n & 1u) {
      r = mulmod(r, sq);
    }
    sq = mulmod(sq, sq);
  }
  return r;
}

template <uint32_t kSeg>
struct Tables
{
  static constexpr uint32_t kTile = 64 * kSeg;
  uint32_t slice[4][256]; /* slicing-by-4: slice[k][b] = remainder of byte b followed by k zero bytes */
  uint32_t skip[4][256];  /* skip[k][b] = (b << 8 k) * x^(8 (kTile - kSeg)) mod P */
  uint32_t fin[8];        /* fin[j] = x^(8 * kSeg * 2^j) mod P, j = 0..5 */
  constexpr Tables() : slice(), skip(), fin()
  {
    for (uint32_t b = 0; b < 256; ++b) {
      uint32_t c = b;
      for (int k = 0; k < 8; ++k) {
        c = (c & 1u) ? kPoly ^ (c >> 1) : c >> 1;
      }
      slice[0][b] = c;
    }
    for (uint32_t b = 0; b < 256; ++b) {
      for (uint32_t k = 1; k < 4; ++k) {
        slice[k][b] = (slice[k - 1][b] >> 8) ^ slice[0][slice[k - 1][b] & 0xffu];
      }
    }
    const uint32_t z = xpow_bytes(kTile - kSeg);
    for (uint32_t b = 0; b < 256; ++b) {
      for (uint32_t k = 0; k < 4; ++k) {
        skip[k][b] = mulmod(b << (8 * k), z);
      }
    }
    for (uint32_t j = 0; j < 6; ++j) {
      fin[j] = xpow_bytes((uint64_t)kSeg << j);
    }
  }
};
template <uint32_t kSeg>
__constant__ static const Tables<kSeg> kTables = Tables<kSeg>();

constexpr uint32_t kLdsDwords = 2048; /* slice | skip, per workgroup */

/* Copy the lookup tables to the workgroup's LDS; a __syncthreads() must follow. */
template <uint32_t kSeg = kSegDefault>
__device__ __forceinline__ void load_tables(uint32_t* lds)
{
  const uint32_t* src = &kTables<kSeg>.slice[0][0];
  for (uint32_t i = threadIdx.x; i < kLdsDwords; i += blockDim.x) {
    lds[i] = src[i];
  }
}

__device__ __forceinline__ uint32_t lookup4(const uint32_t* t, uint32_t s)
{
  return t[s & 0xffu] ^ t[256 + ((s >> 8) & 0xffu)] ^ t[512 + ((s >> 16) & 0xffu)] ^ t[768 + (s >> 24)];
}

/* one dword of message (little endian) into the raw state s */
__device__ __forceinline__ uint32_t step_dword(const uint32_t* slice, uint32_t s, uint32_t w)
{
  s ^= w;
  return slice[768 + (s & 0xffu)] ^ slice[512 + ((s >> 8) & 0xffu)] ^ slice[256 + ((s >> 16) & 0xffu)] ^ slice[s >> 24];
}

/* s * k mod P for a wave-uniform constant k: the constant's multiples k x^i come from a scalar shift chain */
__device__ __forceinline__ uint32_t mul_const(uint32_t s, uint32_t k)
{
  uint32_t p = 0;
#pragma unroll
  for (uint32_t i = 0; i < 32; ++i) {
    p ^= (uint32_t)(-(int32_t)((s >> (31 - i)) & 1u)) & k;
    k = (k & 1u) ? (k >> 1) ^ kPoly : k >> 1;
  }
  return p;
}

/* one lane's kSeg bytes of a tile: 16-byte loads, and the slicing steps over them */
template <uint32_t kSeg>
struct Piece
{
  wave::u32x4 w[kSeg / 16];
  __device__ __forceinline__ void load(const uint8_t* q)
  {
#pragma unroll
    for (uint32_t i = 0; i < kSeg / 16; ++i) {
      w[i] = wave::gload_u32x4(q + 16 * i);
    }
  }
  __device__ __forceinline__ uint32_t into(const uint32_t* slice, uint32_t s) const
  {
#pragma unroll
    for (uint32_t i = 0; i < kSeg / 16; ++i) {
      s = step_dword(slice, s, w[i].x), s = step_dword(slice, s, w[i].y);
      s = step_dword(slice, s, w[i].z), s = step_dword(slice, s, w[i].w);
    }
    return s;
  }
};

/* CRC-32 of p[0, n) with the calling wave (n < 2^31); `lds` = the tables (load_tables). The result is wave-uniform.
 * (Issuing the loads of tile t + 1 before the lookups of tile t was measured and changes nothing: the other waves of the
 * CU fill the gaps. scripts/probes/crc_bench.hip: loads alone 6.0 TB/s, lookups alone 9.2, the kernel 4.9.) */
template <uint32_t kSeg = kSegDefault>
__device__ __forceinline__ uint32_t wave_crc32(const uint8_t* p, uint32_t n, const uint32_t* lds)
{
  static_assert(kSeg == 16 || kSeg == 32 || kSeg == 64, "one, two or four 16-byte loads per lane and tile");
  constexpr uint32_t kTile = 64 * kSeg;
  if (n == 0) {
    return 0;
  }
  const uint32_t lane = (uint32_t)wave::lane_id();
  const uint32_t* slice = lds;
  const uint32_t* skip = lds + 1024;
  const uint32_t pad = (kTile - n % kTile) % kTile; /* virtual zero bytes in front */
  const uint32_t tiles = (n + pad) / kTile;
  const uint32_t v0 = lane * kSeg;                  /* virtual position of my bytes in the first tile */
  const uint8_t* q = p + v0 - pad;                  /* (in front of p for the lanes of the padding: not looked at) */
  uint32_t s = 0;
  Piece<kSeg> cur;
  /* the first tile: the message starts somewhere inside it */
  if (v0 >= pad) {
    cur.load(q);
    s = cur.into(slice, v0 == pad ? 0xffffffffu : 0u); /* the message starts here: every CRC-32 starts from all ones */
  } else if (v0 + kSeg > pad) {
    /* the message starts inside my bytes: byte by byte from its first byte on */
    s = 0xffffffffu;
    for (uint32_t i = pad - v0; i < kSeg; ++i) {
      s = slice[(s ^ wave::gload_u8(q + i)) & 0xffu] ^ (s >> 8);
    }
  }
  /* whole tiles */
  for (uint32_t t = 1; t < tiles; ++t) {
    cur.load(q + (size_t)t * kTile);
    s = cur.into(slice, lookup4(skip, s));
  }
  /* (63 - lane) * kSeg bytes of the message follow my last byte */
  const uint32_t behind = 63u - lane;
#pragma unroll
  for (uint32_t j = 0; j < 6; ++j) {
    const uint32_t m = mul_const(s, kTables<kSeg>.fin[j]);
    s = ((behind >> j) & 1u) ? m : s;
  }
  /* XOR across the wave */
  uint32_t x = s;
  x ^= wave::shuffle(x, lane ^ 32u);
  x ^= wave::shuffle(x, lane ^ 16u);
  x ^= wave::shuffle(x, lane ^ 8u);
  x ^= wave::shuffle(x, lane ^ 4u);
  x ^= wave::shuffle(x, lane ^ 2u);
  x ^= wave::shuffle(x, lane ^ 1u);
  return wave::uniform(x) ^ 0xffffffffu;
}

} // namespace crc32w
