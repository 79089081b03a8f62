/*
 * oracle/lz4_block.c -- CPU restatement of the LZ4 *block* format codec.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under nvcomp_amd/ or include/ may call,
 * link or import this file; it exists so that tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg can check (never produce) results.
 *
 * What it restates. The reference (NVIDIA/nvcomp @ 2024_10_08) ships no codec
 * source (README.md:10); its LZ4 path is pinned only by interoperability with
 * the third-party dependency liblz4 (container: lz4 1.9.3, /opt/conda):
 *   - the GPU decoder must decode LZ4_compress_HC(...,12) output
 *       examples/lz4_cpu_compression.cu:59-74,137
 *   - LZ4_decompress_safe must decode the GPU compressor's output
 *       examples/lz4_cpu_decompression.cu:142-157
 * so the wire format is the published LZ4 block format, restated here:
 *   sequence := token(1B: hi nibble = literal length L, lo nibble = match code M)
 *               [L==15: extra length bytes, summed until one != 255]
 *               L literal bytes
 *               -- the block ends here if the input is exhausted --
 *               offset (2B little endian, 1..65535; 0 is invalid)
 *               [M==15: extra length bytes, summed until one != 255]
 *               match of length M+4 copied byte-by-byte from out-offset
 * Compressor end rules (needed for LZ4_decompress_safe to accept a block):
 *   last 5 bytes are literals; the last match starts >= 12 bytes before the end
 *   of the block; inputs shorter than 13 bytes are emitted as literals only.
 *
 * Parity pin: tests/test_oracle_cpu.py checks this decoder against liblz4
 * (oracle/_ref/libcpucodecs.so) on the reference's fixture files, the
 * reference's synthetic generators and fuzzed streams, and against the
 * committed golden vectors in tests/golden/ (made by scripts/make_golden.py).
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "oracle.h"

/* ---- decoder ------------------------------------------------------------ */

int oracle_lz4_decompress(
    const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_cap, size_t* out_len)
{
  size_t ip = 0, op = 0;
  *out_len = 0;
  if (src_len == 0) {
    return ORACLE_OK; /* zero-length chunk decodes to zero bytes (CHANGELOG.md:66) */
  }
  for (;;) {
    if (ip >= src_len) {
      return ORACLE_ERR_INPUT; /* a token must follow every match */
    }
    const unsigned token = src[ip++];
    size_t lit = token >> 4;
    if (lit == 15) {
      unsigned b;
      do {
        if (ip >= src_len) {
          return ORACLE_ERR_INPUT;
        }
        b = src[ip++];
        lit += b;
      } while (b == 255);
    }
    if (lit > src_len - ip) {
      return ORACLE_ERR_INPUT;
    }
    if (lit > dst_cap - op) {
      return ORACLE_ERR_OUTPUT;
    }
    memcpy(dst + op, src + ip, lit);
    ip += lit;
    op += lit;
    if (ip == src_len) {
      break; /* last sequence: literals only */
    }
    if (src_len - ip < 2) {
      return ORACLE_ERR_INPUT;
    }
    const size_t offset = (size_t)src[ip] | ((size_t)src[ip + 1] << 8);
    ip += 2;
    if (offset == 0 || offset > op) {
      return ORACLE_ERR_OFFSET;
    }
    size_t mlen = token & 15;
    if (mlen == 15) {
      unsigned b;
      do {
        if (ip >= src_len) {
          return ORACLE_ERR_INPUT;
        }
        b = src[ip++];
        mlen += b;
      } while (b == 255);
    }
    mlen += 4;
    if (mlen > dst_cap - op) {
      return ORACLE_ERR_OUTPUT;
    }
    /* byte-serial copy: overlapping (offset < mlen) matches replicate a pattern */
    const uint8_t* m = dst + op - offset;
    for (size_t i = 0; i < mlen; ++i) {
      dst[op + i] = m[i];
    }
    op += mlen;
  }
  *out_len = op;
  return ORACLE_OK;
}

/* Token walk without copying: what nvcompBatchedLZ4GetDecompressSizeAsync
 * computes (examples/low_level_quickstart_example.cpp:112-117). Returns 0 for
 * streams whose structure is broken. */
size_t oracle_lz4_decompressed_size(const uint8_t* src, size_t src_len)
{
  size_t ip = 0, op = 0;
  if (src_len == 0) {
    return 0;
  }
  for (;;) {
    if (ip >= src_len) {
      return 0;
    }
    const unsigned token = src[ip++];
    size_t lit = token >> 4;
    if (lit == 15) {
      unsigned b;
      do {
        if (ip >= src_len) {
          return 0;
        }
        b = src[ip++];
        lit += b;
      } while (b == 255);
    }
    if (lit > src_len - ip) {
      return 0;
    }
    ip += lit;
    op += lit;
    if (ip == src_len) {
      return op;
    }
    if (src_len - ip < 2) {
      return 0;
    }
    ip += 2;
    size_t mlen = token & 15;
    if (mlen == 15) {
      unsigned b;
      do {
        if (ip >= src_len) {
          return 0;
        }
        b = src[ip++];
        mlen += b;
      } while (b == 255);
    }
    op += mlen + 4;
  }
}

/* ---- compressor (greedy single-probe hash, like the "fast" CPU class) ---- */

size_t oracle_lz4_compress_bound(size_t n)
{
  return n + n / 255 + 16;
}

static inline uint32_t rd32(const uint8_t* p)
{
  uint32_t v;
  memcpy(&v, p, 4);
  return v;
}

static uint8_t* put_len(uint8_t* op, size_t len)
{
  while (len >= 255) {
    *op++ = 255;
    len -= 255;
  }
  *op++ = (uint8_t)len;
  return op;
}

size_t oracle_lz4_compress(const uint8_t* src, size_t n, uint8_t* dst, size_t dst_cap)
{
  enum { HASH_BITS = 14, MINMATCH = 4, MFLIMIT = 12, LASTLITERALS = 5 };
  if (dst_cap < oracle_lz4_compress_bound(n)) {
    return 0;
  }
  if (n == 0) {
    return 0; /* empty chunk -> empty stream */
  }
  static __thread uint32_t table[1 << HASH_BITS];
  memset(table, 0xff, sizeof(table));
  uint8_t* op = dst;
  size_t anchor = 0, ip = 0;
  if (n >= (size_t)MFLIMIT + 1) {
    const size_t mflimit = n - MFLIMIT;     /* last position a match may start at */
    const size_t matchlimit = n - LASTLITERALS; /* matches may not extend past this */
    while (ip <= mflimit) {
      const uint32_t seq = rd32(src + ip);
      const uint32_t h = (seq * 2654435761u) >> (32 - HASH_BITS);
      const uint32_t cand = table[h];
      table[h] = (uint32_t)ip;
      if (cand != 0xffffffffu && ip - cand <= 65535 && rd32(src + cand) == seq) {
        size_t mlen = MINMATCH;
        while (ip + mlen < matchlimit && src[cand + mlen] == src[ip + mlen]) {
          ++mlen;
        }
        const size_t lit = ip - anchor;
        uint8_t* token = op++;
        if (lit >= 15) {
          *token = 15 << 4;
          op = put_len(op, lit - 15);
        } else {
          *token = (uint8_t)(lit << 4);
        }
        memcpy(op, src + anchor, lit);
        op += lit;
        const size_t off = ip - cand;
        *op++ = (uint8_t)(off & 255);
        *op++ = (uint8_t)(off >> 8);
        const size_t mc = mlen - MINMATCH;
        if (mc >= 15) {
          *token |= 15;
          op = put_len(op, mc - 15);
        } else {
          *token |= (uint8_t)mc;
        }
        ip += mlen;
        anchor = ip;
      } else {
        ++ip;
      }
    }
  }
  /* last literals */
  {
    const size_t lit = n - anchor;
    uint8_t* token = op++;
    if (lit >= 15) {
      *token = 15 << 4;
      op = put_len(op, lit - 15);
    } else {
      *token = (uint8_t)(lit << 4);
    }
    memcpy(op, src + anchor, lit);
    op += lit;
  }
  return (size_t)(op - dst);
}
