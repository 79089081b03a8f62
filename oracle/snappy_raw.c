/*
 * oracle/snappy_raw.c -- CPU restatement of the Snappy *raw* format codec.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/lz4_block.c header for the rule).
 *
 * What it restates. The reference's Snappy path is closed source; call sites
 * benchmarks/benchmark_snappy_chunked.cu:50-64 and
 * benchmarks/benchmark_snappy_synth.cpp:128-143,163-190,220-268 only require a
 * byte-exact round trip, and CHANGELOG.md:182-184 that every standard-legal
 * stream decodes. BASELINE.json's north_star adds bit-exactness against the
 * third-party snappy CPU decoder (container: snappy 1.1.8, /opt/conda). The
 * published raw format, restated:
 *   preamble: uncompressed length, little-endian base-128 varint (<= 5 bytes)
 *   element by (tag & 3):
 *     00 literal : len-1 = tag>>2 if < 60, else the next (tag>>2)-59 bytes (LE) hold len-1
 *     01 copy-1  : len = 4 + ((tag>>2)&7), offset = ((tag>>5)<<8) | next byte
 *     10 copy-2  : len = 1 + (tag>>2),     offset = next 2 bytes LE
 *     11 copy-4  : len = 1 + (tag>>2),     offset = next 4 bytes LE
 *   offset 0 or > bytes produced is an error; copies are byte-serial (overlap legal);
 *   the elements must produce exactly the preamble length.
 * Bound: 32 + n + n/6.
 *
 * Parity pin: tests/test_oracle_cpu.py vs libsnappy (oracle/_ref) on fixtures,
 * synthetic generators and hand-built streams using every element kind, plus
 * the golden vectors in tests/golden/.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "oracle.h"

static int read_varint(const uint8_t* src, size_t n, size_t* ip, uint64_t* val)
{
  uint64_t v = 0;
  for (unsigned shift = 0; shift <= 28; shift += 7) {
    if (*ip >= n) {
      return 0;
    }
    const unsigned b = src[(*ip)++];
    v |= (uint64_t)(b & 127) << shift;
    if (!(b & 128)) {
      if (shift == 28 && b > 15) {
        return 0; /* does not fit 32 bits */
      }
      *val = v;
      return 1;
    }
  }
  return 0;
}

int oracle_snappy_decompressed_size(const uint8_t* src, size_t src_len, size_t* out_len)
{
  size_t ip = 0;
  uint64_t v;
  *out_len = 0;
  if (!read_varint(src, src_len, &ip, &v)) {
    return ORACLE_ERR_INPUT;
  }
  *out_len = (size_t)v;
  return ORACLE_OK;
}

int oracle_snappy_decompress(
    const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_cap, size_t* out_len)
{
  size_t ip = 0, op = 0;
  uint64_t total;
  *out_len = 0;
  if (!read_varint(src, src_len, &ip, &total)) {
    return ORACLE_ERR_INPUT;
  }
  if (total > dst_cap) {
    return ORACLE_ERR_OUTPUT;
  }
  while (ip < src_len) {
    const unsigned tag = src[ip++];
    size_t len, offset;
    switch (tag & 3) {
    case 0: {
      len = tag >> 2;
      if (len >= 60) {
        const unsigned nb = (unsigned)len - 59;
        if (src_len - ip < nb) {
          return ORACLE_ERR_INPUT;
        }
        len = 0;
        for (unsigned i = 0; i < nb; ++i) {
          len |= (size_t)src[ip + i] << (8 * i);
        }
        ip += nb;
      }
      len += 1;
      if (len > src_len - ip) {
        return ORACLE_ERR_INPUT;
      }
      if (len > total - op) {
        return ORACLE_ERR_OUTPUT;
      }
      memcpy(dst + op, src + ip, len);
      ip += len;
      op += len;
      continue;
    }
    case 1:
      if (src_len - ip < 1) {
        return ORACLE_ERR_INPUT;
      }
      len = 4 + ((tag >> 2) & 7);
      offset = ((size_t)(tag >> 5) << 8) | src[ip];
      ip += 1;
      break;
    case 2:
      if (src_len - ip < 2) {
        return ORACLE_ERR_INPUT;
      }
      len = 1 + (tag >> 2);
      offset = (size_t)src[ip] | ((size_t)src[ip + 1] << 8);
      ip += 2;
      break;
    default:
      if (src_len - ip < 4) {
        return ORACLE_ERR_INPUT;
      }
      len = 1 + (tag >> 2);
      offset = (size_t)src[ip] | ((size_t)src[ip + 1] << 8) | ((size_t)src[ip + 2] << 16)
               | ((size_t)src[ip + 3] << 24);
      ip += 4;
      break;
    }
    if (offset == 0 || offset > op) {
      return ORACLE_ERR_OFFSET;
    }
    if (len > total - op) {
      return ORACLE_ERR_OUTPUT;
    }
    const uint8_t* m = dst + op - offset;
    for (size_t i = 0; i < len; ++i) {
      dst[op + i] = m[i];
    }
    op += len;
  }
  if (op != total) {
    return ORACLE_ERR_INPUT;
  }
  *out_len = op;
  return ORACLE_OK;
}

/* ---- compressor: greedy single-probe hash, 16-bit offsets ---------------- */

size_t oracle_snappy_compress_bound(size_t n)
{
  return 32 + n + n / 6;
}

static inline uint32_t rd32(const uint8_t* p)
{
  uint32_t v;
  memcpy(&v, p, 4);
  return v;
}

static uint8_t* emit_literal(uint8_t* op, const uint8_t* lit, size_t len)
{
  const size_t n = len - 1;
  if (n < 60) {
    *op++ = (uint8_t)(n << 2);
  } else {
    unsigned nb = 0;
    for (size_t t = n; t; t >>= 8) {
      ++nb;
    }
    *op++ = (uint8_t)((59 + nb) << 2);
    for (unsigned i = 0; i < nb; ++i) {
      *op++ = (uint8_t)(n >> (8 * i));
    }
  }
  memcpy(op, lit, len);
  return op + len;
}

static uint8_t* emit_copy_upto64(uint8_t* op, size_t offset, size_t len)
{
  if (len < 12 && offset < 2048 && len >= 4) {
    *op++ = (uint8_t)(1 | ((len - 4) << 2) | ((offset >> 8) << 5));
    *op++ = (uint8_t)(offset & 255);
  } else {
    *op++ = (uint8_t)(2 | ((len - 1) << 2));
    *op++ = (uint8_t)(offset & 255);
    *op++ = (uint8_t)(offset >> 8);
  }
  return op;
}

static uint8_t* emit_copy(uint8_t* op, size_t offset, size_t len)
{
  while (len >= 68) {
    op = emit_copy_upto64(op, offset, 64);
    len -= 64;
  }
  if (len > 64) {
    op = emit_copy_upto64(op, offset, 60);
    len -= 60;
  }
  return emit_copy_upto64(op, offset, len);
}

size_t oracle_snappy_compress(const uint8_t* src, size_t n, uint8_t* dst, size_t dst_cap)
{
  enum { HASH_BITS = 14 };
  if (dst_cap < oracle_snappy_compress_bound(n) || n > 0xffffffffu) {
    return 0;
  }
  uint8_t* op = dst;
  for (size_t v = n;;) {
    if (v < 128) {
      *op++ = (uint8_t)v;
      break;
    }
    *op++ = (uint8_t)(v | 128);
    v >>= 7;
  }
  static __thread uint32_t table[1 << HASH_BITS];
  memset(table, 0xff, sizeof(table));
  size_t anchor = 0, ip = 0;
  while (n >= 4 && ip + 4 <= n) {
    const uint32_t seq = rd32(src + ip);
    const uint32_t h = (seq * 0x1e35a7bdu) >> (32 - HASH_BITS);
    const uint32_t cand = table[h];
    table[h] = (uint32_t)ip;
    if (cand != 0xffffffffu && ip - cand <= 65535 && rd32(src + cand) == seq) {
      size_t mlen = 4;
      while (ip + mlen < n && src[cand + mlen] == src[ip + mlen]) {
        ++mlen;
      }
      if (ip > anchor) {
        op = emit_literal(op, src + anchor, ip - anchor);
      }
      op = emit_copy(op, ip - cand, mlen);
      ip += mlen;
      anchor = ip;
    } else {
      ++ip;
    }
  }
  if (n > anchor) {
    op = emit_literal(op, src + anchor, n - anchor);
  }
  return (size_t)(op - dst);
}
