/*
 * lz4/lz4_encode.hip.h -- batched LZ4 block-format compressor for gfx950.
 *
 * Replaces the device side of nvcompBatchedLZ4CompressAsync (reference call
 * sites: benchmarks/benchmark_template_chunked.cuh:441-451,
 * examples/lz4_cpu_decompression.cu:94-104). The output must be accepted by
 * liblz4's LZ4_decompress_safe (examples/lz4_cpu_decompression.cu:142-157), so
 * the end-of-block rules of the format are honoured: the last 5 bytes are
 * literals, the last match starts at least 12 bytes before the end, chunks
 * shorter than 13 bytes are stored as literals.
 *
 * One wavefront per chunk; the match finder is common/lz_match.hip.h, this file
 * is the LZ4 sequence emitter.
 */
#pragma once

#include "common/lz_match.hip.h"
#include "common/lz_match_wide.hip.h"
#include "common/lz_match_runs.hip.h"

namespace lz4 {

constexpr uint32_t kMinMatch = 4;
constexpr uint32_t kMfLimit = 12;     /* last match must start this far before the end */
constexpr uint32_t kLastLiterals = 5; /* and the last 5 bytes are always literals */

/* Length-extension bytes for value v (15 already subtracted by the caller):
 * v / 255 + 1 bytes, all 255 except the last which is v % 255. */
__device__ __forceinline__ void put_extension(uint8_t* dst, uint32_t v)
{
  const uint32_t count = v / 255 + 1;
  for (uint32_t i = (uint32_t)wave::lane_id(); i < count; i += 64) {
    dst[i] = (i + 1 < count) ? (uint8_t)255 : (uint8_t)(v % 255);
  }
}

/* Emit one sequence; returns the number of bytes written. match_len == 0 emits
 * the final literal-only sequence. */
__device__ __forceinline__ uint32_t emit_sequence(
    uint8_t* dst, const uint8_t* lit, uint32_t lit_len, uint32_t offset, uint32_t match_len)
{
  const uint32_t ml = match_len ? match_len - kMinMatch : 0;
  const uint32_t tok = ((lit_len < 15 ? lit_len : 15u) << 4) | (ml < 15 ? ml : 15u);
  uint32_t pos = 1;
  if (wave::lane_id() == 0) {
    dst[0] = (uint8_t)tok;
  }
  if (lit_len >= 15) {
    put_extension(dst + pos, lit_len - 15);
    pos += (lit_len - 15) / 255 + 1;
  }
  lz::wave_copy(dst + pos, lit, lit_len);
  pos += lit_len;
  if (match_len) {
    if (wave::lane_id() == 0) {
      dst[pos] = (uint8_t)(offset & 255u);
      dst[pos + 1] = (uint8_t)(offset >> 8);
    }
    pos += 2;
    if (ml >= 15) {
      put_extension(dst + pos, ml - 15);
      pos += (ml - 15) / 255 + 1;
    }
  }
  return pos;
}

/* The same as a function of its own, for the match finder's rare whole-wave emissions (a sequence no single lane can write):
 * inlined into its loop these cost the kernel 10-12 spilled registers; the chunk's tail (one call, and for incompressible
 * data the whole chunk: through a call its copies lose their address space, -6 %) stays inlined. */
#ifndef NVCOMP_LZ4_EMIT_NOINLINE
#define NVCOMP_LZ4_EMIT_NOINLINE 1
#endif
#if NVCOMP_LZ4_EMIT_NOINLINE
__device__ __attribute__((noinline)) uint32_t emit_sequence_call(
#else
__device__ __forceinline__ uint32_t emit_sequence_call(
#endif
    uint8_t* dst, const uint8_t* lit, uint32_t lit_len, uint32_t offset, uint32_t match_len)
{
  return emit_sequence(dst, lit, lit_len, offset, match_len);
}

struct Emitter
{
  static constexpr bool kStream = false;   /* sequences start at byte boundaries: lzm writes them where they go */
#ifndef NVCOMP_LZ4_REACH
#define NVCOMP_LZ4_REACH 65535
#endif
  static constexpr uint32_t kReach = NVCOMP_LZ4_REACH; /* 2-byte offsets */
  static __device__ __forceinline__ uint32_t ext_bytes(uint32_t v) /* extension bytes for a length code v */
  {
    return v >= 15 ? (v - 15) / 255 + 1 : 0;
  }
  static __device__ __forceinline__ uint32_t seq_size(uint32_t lit_len, uint32_t match_len, uint32_t /*offset*/)
  {
    return 1 + ext_bytes(lit_len) + lit_len + 2 + ext_bytes(match_len - kMinMatch);
  }
  /* one lane can write it: at most one extension byte per length, literal run <= 64 bytes */
  static __device__ __forceinline__ bool is_small(uint32_t lit_len, uint32_t match_len)
  {
    return lit_len <= 64 && match_len - kMinMatch < 15 + 255;
  }
  static __device__ __forceinline__ uint32_t lit_offset(uint32_t lit_len)
  {
    return 1 + (lit_len >= 15 ? 1u : 0u);
  }
  static __device__ __forceinline__ void emit_small_header(uint8_t* dst, uint32_t lit_len, uint32_t offset, uint32_t match_len)
  {
    const uint32_t ml = match_len - kMinMatch;
    dst[0] = (uint8_t)(((lit_len < 15 ? lit_len : 15u) << 4) | (ml < 15 ? ml : 15u));
    uint32_t pos = 1;
    if (lit_len >= 15) {
      dst[pos++] = (uint8_t)(lit_len - 15);
    }
    pos += lit_len;
    dst[pos] = (uint8_t)(offset & 255u);
    dst[pos + 1] = (uint8_t)(offset >> 8);
    if (ml >= 15) {
      dst[pos + 2] = (uint8_t)(ml - 15);
    }
  }
  static __device__ __forceinline__ uint32_t match(
      uint8_t* dst, const uint8_t* lit, uint32_t lit_len, uint32_t offset, uint32_t match_len)
  {
    return emit_sequence_call(dst, lit, lit_len, offset, match_len);
  }
  /* the same written by ONE lane (common/lz_match_runs.hip.h: 64 sequences at a time, literal runs of at most 64 bytes);
   * seq_size() bytes */
  static __device__ __forceinline__ void emit_lane(uint8_t* dst, const uint8_t* lit, uint32_t lit_len, uint32_t offset, uint32_t match_len)
  {
    const uint32_t ml = match_len - kMinMatch;
    dst[0] = (uint8_t)(((lit_len < 15 ? lit_len : 15u) << 4) | (ml < 15 ? ml : 15u));
    uint32_t pos = 1;
    if (lit_len >= 15) {
      uint32_t v = lit_len - 15;
      for (; v >= 255; v -= 255) {
        dst[pos++] = 255;
      }
      dst[pos++] = (uint8_t)v;
    }
    for (uint32_t i = 0; i < lit_len; ++i) {
      dst[pos + i] = lit[i];
    }
    pos += lit_len;
    dst[pos] = (uint8_t)(offset & 255u);
    dst[pos + 1] = (uint8_t)(offset >> 8);
    pos += 2;
    if (ml >= 15) {
      uint32_t v = ml - 15;
      for (; v >= 255; v -= 255) {
        dst[pos++] = 255;
      }
      dst[pos++] = (uint8_t)v;
    }
  }
  static __device__ __forceinline__ uint32_t tail(uint8_t* dst, const uint8_t* lit, uint32_t lit_len)
  {
    return emit_sequence(dst, lit, lit_len, 0, 0);
  }
};

/* Compress src[0,n) into dst (capacity >= n + n/255 + 16) with the calling
 * wave; `table` is this wave's LDS hash table, lzm::kTableU16 x uint16, `image` its
 * lzm::kStageBytes of LDS for the input image (8-byte aligned). Returns
 * the compressed size. */
template <uint32_t STRIDE = 1>
__device__ __forceinline__ uint32_t encode_chunk(
    const uint8_t* __restrict__ src, uint32_t n, uint8_t* dst, uint16_t* table, uint8_t* image)
{
  const bool any = n > kMfLimit;
  /* runs first (common/lz_match_runs.hip.h): typed columns are what the data_type option is for */
  static_assert(lzm::kTableU16 * 2 >= lzm::runs::kListBytes, "the run compressor's list lives in the hash table's LDS");
  const uint32_t as_runs = lzm::runs::encode_chunk<Emitter>(src, n, dst, (uint32_t*)table, any ? n - kMfLimit : 0, any ? n - kLastLiterals : 0, any);
  if (as_runs != lzm::runs::kNotRuns) {
    return as_runs;
  }
  return lzm::encode_chunk<Emitter, STRIDE>(
      src, n, dst, table, image, any ? n - kMfLimit : 0, any ? n - kLastLiterals : 0, any);
}

/* The same for untyped data with the 256-position steps of common/lz_match_wide.hip.h (`table`: lzm::wide::kEntries x
 * uint16, `image`: lzm::wide::kImage bytes, `scratch`: lzm::wide::kScratch bytes, both 16-byte aligned). */
__device__ __forceinline__ uint32_t encode_chunk_wide(
    const uint8_t* __restrict__ src, uint32_t n, uint8_t* dst, uint16_t* table, uint8_t* image, uint8_t* scratch)
{
  const bool any = n > kMfLimit;
  /* runs (sorted keys, typed columns, zeros) first: common/lz_match_runs.hip.h */
  static_assert(lzm::wide::kEntries * 2 >= lzm::runs::kListBytes, "the run compressor's list lives in the hash table's LDS");
  const uint32_t as_runs = lzm::runs::encode_chunk<Emitter>(src, n, dst, (uint32_t*)table, any ? n - kMfLimit : 0, any ? n - kLastLiterals : 0, any);
  if (as_runs != lzm::runs::kNotRuns) {
    return as_runs;
  }
  return lzm::wide::encode_chunk<Emitter>(
      src, n, dst, table, image, scratch, any ? n - kMfLimit : 0, any ? n - kLastLiterals : 0, any);
}

} // namespace lz4
