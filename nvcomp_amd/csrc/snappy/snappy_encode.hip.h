/*
 * snappy/snappy_encode.hip.h -- batched Snappy raw-format compressor for gfx950.
 *
 * Replaces the device side of nvcompBatchedSnappyCompressAsync (reference call
 * sites: benchmarks/benchmark_snappy_synth.cpp:163-173,
 * benchmarks/benchmark_template_chunked.cuh:441-451). One wavefront per chunk;
 * the match finder is common/lz_match.hip.h, this file is the element emitter.
 * Output decodes with snappy::RawUncompress (tests) and never exceeds
 * 32 + n + n/6 bytes.
 */
#pragma once

#include "common/lz_match.hip.h"
#include "common/lz_match_wide.hip.h"
#include "common/lz_match_runs.hip.h"

namespace snappy {

/* literal element: tag (+1..4 length bytes) + bytes. Returns bytes written. */
__device__ __forceinline__ uint32_t emit_literal(uint8_t* dst, const uint8_t* lit, uint32_t len)
{
  if (len == 0) {
    return 0;
  }
  const uint32_t n = len - 1;
  uint32_t hdr = 1;
  if (n < 60) {
    if (wave::lane_id() == 0) {
      dst[0] = (uint8_t)(n << 2);
    }
  } else {
    const uint32_t nb = n < (1u << 8) ? 1u : n < (1u << 16) ? 2u : n < (1u << 24) ? 3u : 4u;
    if (wave::lane_id() == 0) {
      dst[0] = (uint8_t)((59 + nb) << 2);
      for (uint32_t i = 0; i < nb; ++i) {
        dst[1 + i] = (uint8_t)(n >> (8 * i));
      }
    }
    hdr += nb;
  }
  lz::wave_copy(dst + hdr, lit, len);
  return hdr + len;
}

/* One copy element of length 1..64 (lane 0 writes). Returns bytes written. */
__device__ __forceinline__ uint32_t emit_copy_piece(uint8_t* dst, uint32_t offset, uint32_t len)
{
  const bool short_form = len >= 4 && len < 12 && offset < 2048;
  if (wave::lane_id() == 0) {
    if (short_form) {
      dst[0] = (uint8_t)(1u | ((len - 4) << 2) | ((offset >> 8) << 5));
      dst[1] = (uint8_t)(offset & 255u);
    } else {
      dst[0] = (uint8_t)(2u | ((len - 1) << 2));
      dst[1] = (uint8_t)(offset & 255u);
      dst[2] = (uint8_t)(offset >> 8);
    }
  }
  return short_form ? 2u : 3u;
}

/* A match of any length as copy elements: 64-byte pieces while >= 68 remain
 * (written in parallel, one piece per lane), one 60-byte piece if > 64 remain
 * (so the last piece is never shorter than 4), then the rest. */
__device__ __forceinline__ uint32_t emit_copy(uint8_t* dst, uint32_t offset, uint32_t len)
{
  uint32_t pos = 0;
  if (len >= 68) {
    const uint32_t full = (len - 68) / 64 + 1;
    for (uint32_t i = (uint32_t)wave::lane_id(); i < full; i += 64) {
      dst[3 * i] = (uint8_t)(2u | (63u << 2));
      dst[3 * i + 1] = (uint8_t)(offset & 255u);
      dst[3 * i + 2] = (uint8_t)(offset >> 8);
    }
    pos = 3 * full;
    len -= 64 * full;
  }
  if (len > 64) {
    pos += emit_copy_piece(dst + pos, offset, 60);
    len -= 60;
  }
  pos += emit_copy_piece(dst + pos, offset, len);
  return pos;
}

struct Emitter
{
  static constexpr bool kStream = false;   /* sequences start at byte boundaries: lzm writes them where they go */
  static constexpr uint32_t kReach = 65535; /* 2-byte offsets */
  static __device__ __forceinline__ uint32_t literal_size(uint32_t len)
  {
    if (len == 0) {
      return 0;
    }
    const uint32_t n = len - 1;
    return len + 1 + (n < 60 ? 0u : n < (1u << 8) ? 1u : n < (1u << 16) ? 2u : n < (1u << 24) ? 3u : 4u);
  }
  static __device__ __forceinline__ uint32_t piece_size(uint32_t offset, uint32_t len)
  {
    return (len >= 4 && len < 12 && offset < 2048) ? 2u : 3u;
  }
  static __device__ __forceinline__ uint32_t copy_size(uint32_t offset, uint32_t len)
  {
    uint32_t sz = 0;
    if (len >= 68) {
      const uint32_t full = (len - 68) / 64 + 1;
      sz = 3 * full;
      len -= 64 * full;
    }
    if (len > 64) {
      sz += piece_size(offset, 60);
      len -= 60;
    }
    return sz + piece_size(offset, len);
  }
  static __device__ __forceinline__ uint32_t seq_size(uint32_t lit_len, uint32_t match_len, uint32_t offset)
  {
    return literal_size(lit_len) + copy_size(offset, match_len);
  }
  static __device__ __forceinline__ bool is_small(uint32_t lit_len, uint32_t match_len)
  {
    return lit_len <= 64 && match_len <= 64;
  }
  static __device__ __forceinline__ uint32_t lit_offset(uint32_t lit_len)
  {
    return lit_len == 0 ? 0u : (lit_len - 1 < 60 ? 1u : 2u);
  }
  static __device__ __forceinline__ void emit_small_header(uint8_t* dst, uint32_t lit_len, uint32_t offset, uint32_t match_len)
  {
    uint32_t pos = 0;
    if (lit_len != 0) {
      const uint32_t n = lit_len - 1;
      if (n < 60) {
        dst[0] = (uint8_t)(n << 2);
        pos = 1;
      } else {
        dst[0] = (uint8_t)(60u << 2);
        dst[1] = (uint8_t)n;
        pos = 2;
      }
      pos += lit_len;
    }
    if (match_len >= 4 && match_len < 12 && offset < 2048) {
      dst[pos] = (uint8_t)(1u | ((match_len - 4) << 2) | ((offset >> 8) << 5));
      dst[pos + 1] = (uint8_t)(offset & 255u);
    } else {
      dst[pos] = (uint8_t)(2u | ((match_len - 1) << 2));
      dst[pos + 1] = (uint8_t)(offset & 255u);
      dst[pos + 2] = (uint8_t)(offset >> 8);
    }
  }
  static __device__ __forceinline__ uint32_t match(
      uint8_t* dst, const uint8_t* lit, uint32_t lit_len, uint32_t offset, uint32_t match_len)
  {
    const uint32_t a = emit_literal(dst, lit, lit_len);
    return a + emit_copy(dst + a, offset, match_len);
  }
  static __device__ __forceinline__ uint32_t tail(uint8_t* dst, const uint8_t* lit, uint32_t lit_len)
  {
    return emit_literal(dst, lit, lit_len);
  }
  /* match() written by ONE lane (common/lz_match_runs.hip.h: 64 sequences at a time, literal runs of at most 64 bytes);
   * seq_size() bytes: the same pieces as emit_copy */
  static __device__ __forceinline__ void emit_lane(uint8_t* dst, const uint8_t* lit, uint32_t lit_len, uint32_t offset, uint32_t match_len)
  {
    uint32_t pos = 0;
    if (lit_len != 0) {
      const uint32_t n = lit_len - 1;
      if (n < 60) {
        dst[pos++] = (uint8_t)(n << 2);
      } else {
        dst[pos++] = (uint8_t)(60u << 2);
        dst[pos++] = (uint8_t)n;
      }
      for (uint32_t i = 0; i < lit_len; ++i) {
        dst[pos + i] = lit[i];
      }
      pos += lit_len;
    }
    uint32_t len = match_len;
    while (len >= 68) {
      dst[pos] = (uint8_t)(2u | (63u << 2));
      dst[pos + 1] = (uint8_t)(offset & 255u);
      dst[pos + 2] = (uint8_t)(offset >> 8);
      pos += 3;
      len -= 64;
    }
    for (;;) {
      const uint32_t piece = len > 64 ? 60u : len;
      if (piece >= 4 && piece < 12 && offset < 2048) {
        dst[pos] = (uint8_t)(1u | ((piece - 4) << 2) | ((offset >> 8) << 5));
        dst[pos + 1] = (uint8_t)(offset & 255u);
        pos += 2;
      } else {
        dst[pos] = (uint8_t)(2u | ((piece - 1) << 2));
        dst[pos + 1] = (uint8_t)(offset & 255u);
        dst[pos + 2] = (uint8_t)(offset >> 8);
        pos += 3;
      }
      len -= piece;
      if (len == 0) {
        break;
      }
    }
  }
};

/* preamble: varint32 of n. Returns its length. */
__device__ __forceinline__ uint32_t put_preamble(uint8_t* dst, uint32_t n)
{
  uint32_t hdr = 0;
  uint32_t v = n;
  uint8_t bytes[5];
  do {
    bytes[hdr] = (uint8_t)((v & 127u) | (v >= 128 ? 128u : 0u));
    v >>= 7;
    ++hdr;
  } while (v);
  if (wave::lane_id() == 0) {
    for (uint32_t i = 0; i < hdr; ++i) {
      dst[i] = bytes[i];
    }
  }
  return hdr;
}

/* Compress src[0,n) into dst (capacity >= 32 + n + n/6). Returns compressed size. */
__device__ __forceinline__ uint32_t encode_chunk(
    const uint8_t* __restrict__ src, uint32_t n, uint8_t* dst, uint16_t* table, uint8_t* image)
{
  const uint32_t hdr = put_preamble(dst, n);
  const bool any = n >= 8;
  return hdr + lzm::encode_chunk<Emitter>(src, n, dst + hdr, table, image, any ? n - 4 : 0, n, any);
}

/* The same with the 256-position steps of common/lz_match_wide.hip.h. */
__device__ __forceinline__ uint32_t encode_chunk_wide(
    const uint8_t* __restrict__ src, uint32_t n, uint8_t* dst, uint16_t* table, uint8_t* image, uint8_t* scratch)
{
  const uint32_t hdr = put_preamble(dst, n);
  /* the wide probe's word check reads 12 bytes at a time (lz_match_wide.hip.h probe_step: a lane without a candidate reads
   * the chunk's first twelve): a chunk of 8 .. 11 bytes is written as literals, never probed */
  const bool any = n >= 12;
  /* runs (sorted keys, typed columns, zeros) first: common/lz_match_runs.hip.h */
  static_assert(lzm::wide::kEntries * 2 >= lzm::runs::kListBytes, "the run compressor's list lives in the hash table's LDS");
  const uint32_t as_runs = lzm::runs::encode_chunk<Emitter>(src, n, dst + hdr, (uint32_t*)table, any ? n - 4 : 0, n, any);
  if (as_runs != lzm::runs::kNotRuns) {
    return hdr + as_runs;
  }
  return hdr + lzm::wide::encode_chunk<Emitter>(src, n, dst + hdr, table, image, scratch, any ? n - 4 : 0, n, any);
}

} // namespace snappy
