/*
 * deflate/deflate_encode.hip.h -- batched DEFLATE compressor for gfx950, first stage.
 *
 * Replaces the device side of nvcompBatchedDeflateCompressAsync (reference call site:
 * examples/deflate_cpu_decompression.cu:93-103; the output must be accepted by libdeflate / zlib inflate,
 * :128-170). The round-2 scope row is the DECODER ("Deflate/Gzip decode"); this compressor exists so
 * that the reference's harness and round-trip callers run: it writes STORED blocks (RFC 1951 3.2.4) -- standard
 * streams every inflater reads, compression ratio 1.0 less five bytes per 65 535. An LZ77 + Huffman stage on top of
 * common/lz_match.hip.h is the next step.
 */
#pragma once

#include "common/lz_common.hip.h"

namespace deflate {

constexpr uint32_t kStoredMax = 65535;

__host__ __device__ inline size_t max_compressed_size(size_t n)
{
  return n + 5 * (n / kStoredMax + 1);
}

/* Compress src[0, n) into dst (capacity >= max_compressed_size(n)) with the calling wave. Returns the size. */
__device__ __forceinline__ uint32_t encode_chunk(const uint8_t* __restrict__ src, uint32_t n, uint8_t* dst)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  uint32_t ip = 0, op = 0;
  do {
    const uint32_t len = n - ip < kStoredMax ? n - ip : kStoredMax;
    const bool final_block = ip + len == n;
    if (lane == 0) {
      dst[op] = final_block ? 1 : 0; /* BFINAL, BTYPE = 00, padding to the byte boundary */
      dst[op + 1] = (uint8_t)len;
      dst[op + 2] = (uint8_t)(len >> 8);
      dst[op + 3] = (uint8_t)~len;
      dst[op + 4] = (uint8_t)(~len >> 8);
    }
    lz::wave_copy(dst + op + 5, src + ip, len);
    ip += len;
    op += 5 + len;
  } while (ip < n);
  return op;
}

} // namespace deflate
