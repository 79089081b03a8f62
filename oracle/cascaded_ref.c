/*
 * oracle/cascaded_ref.c -- CPU restatement of this library's Cascaded codec.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/lz4_block.c header for the rule).
 *
 * PARITY UNPINNED: the reference documents the Cascaded *scheme* but not its
 * bitstream (doc/cascaded_overview.md:6-44; the closed library only decodes its
 * own output, README.md:13), so there is no golden vector to pin against. This
 * file restates the scheme of doc/cascaded_overview.md -- RLE and delta layers
 * interleaved, the RLE values feeding the next layer, then bit-packing of every
 * resulting stream as (value - min) in bits(max - min) bits -- over the container
 * defined in DESIGN.md ("Cascaded stream layout"). It is the bit-exact model of
 * nvcomp_amd/csrc/cascaded/*.hip.h: tests require identical compressed bytes and
 * identical decompressed bytes, and property-test the round trip over all 8
 * element types x RLE/delta/bit-packing combinations.
 *
 * Container (little endian, 4-byte aligned):
 *   chunk header : u32 magic 'CASC' | u8 type, u8 num_RLEs, u8 num_deltas, u8 use_bp
 *                | u32 uncompressed_bytes | u32 sub_chunk_bytes | u32 num_sub
 *                | u32 sub_end[num_sub]   (cumulative payload bytes after each sub-chunk)
 *   sub-chunk    : u32 n_elems, or 0xffffffff followed by the raw bytes (padded to 4)
 *                  when the cascade would not be smaller than the raw sub-chunk;
 *                  u32 count[l] for each RLE layer l (elements left after that layer);
 *                  then the streams runs[0] .. runs[R-1], values, each as
 *                  u32 bits | u64 min | ceil(count * bits / 32) x u32 packed words.
 *   Delta keeps the first value: d[0] = v[0], d[i] = v[i] - v[i-1] (wrapping at the
 *   element width). Run lengths are >= 1 and stored as-is. With use_bp = 0 streams
 *   are stored at their native width (bits = 16 for runs, 8 * width for values, min 0).
 *   Values after a delta layer are compared as signed for min/max, otherwise with
 *   the signedness of the element type (doc/cascaded_overview.md:35).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

#define CASC_MAGIC 0x43534143u /* 'CASC' */
#define CASC_MAX_ELEMS 16384

static unsigned type_width(int type)
{
  switch (type) {
  case 0: case 1: return 1;
  case 2: case 3: return 2;
  case 4: case 5: return 4;
  case 6: case 7: return 8;
  default: return 0;
  }
}

static int type_signed(int type)
{
  return type == 0 || type == 2 || type == 4 || type == 6;
}

static uint64_t width_mask(unsigned w)
{
  return w == 8 ? ~0ull : ((1ull << (8 * w)) - 1);
}

/* sign-extend a w-byte value held in the low bits */
static int64_t sext(uint64_t v, unsigned w)
{
  const unsigned sh = 64 - 8 * w;
  return (int64_t)(v << sh) >> sh;
}

static unsigned bits_for(uint64_t range)
{
  unsigned b = 0;
  while (range) {
    ++b;
    range >>= 1;
  }
  return b;
}

static void put32(uint8_t* p, uint32_t v)
{
  memcpy(p, &v, 4);
}

static uint32_t get32(const uint8_t* p)
{
  uint32_t v;
  memcpy(&v, p, 4);
  return v;
}

/* Pack `count` values; returns bytes written. as_signed selects the min/max order. */
static size_t pack_stream(uint8_t* dst, const uint64_t* v, uint32_t count, unsigned w, int as_signed, int use_bp,
                          unsigned native_bits)
{
  uint64_t mn = 0;
  unsigned bits = native_bits;
  if (use_bp) {
    if (count == 0) {
      bits = 0;
    } else if (as_signed) {
      int64_t lo = sext(v[0], w), hi = lo;
      for (uint32_t i = 1; i < count; ++i) {
        const int64_t x = sext(v[i], w);
        if (x < lo) lo = x;
        if (x > hi) hi = x;
      }
      mn = (uint64_t)lo & width_mask(w);
      bits = bits_for((uint64_t)hi - (uint64_t)lo);
    } else {
      uint64_t lo = v[0], hi = v[0];
      for (uint32_t i = 1; i < count; ++i) {
        if (v[i] < lo) lo = v[i];
        if (v[i] > hi) hi = v[i];
      }
      mn = lo;
      bits = bits_for(hi - lo);
    }
  }
  put32(dst, bits);
  memcpy(dst + 4, &mn, 8);
  const size_t words = ((size_t)count * bits + 31) / 32;
  uint32_t* out = (uint32_t*)calloc(words ? words : 1, 4);
  const uint64_t vmask = bits == 64 ? ~0ull : ((1ull << bits) - 1);
  for (uint32_t i = 0; i < count && bits; ++i) {
    const uint64_t x = ((v[i] - mn) & width_mask(w)) & vmask;
    const size_t bit = (size_t)i * bits;
    const size_t k = bit / 32;
    const unsigned sh = (unsigned)(bit % 32);
    out[k] |= (uint32_t)(x << sh);
    if (sh + bits > 32) {
      out[k + 1] |= (uint32_t)(x >> (32 - sh));
    }
    if (sh + bits > 64) {
      out[k + 2] |= (uint32_t)(x >> (64 - sh));
    }
  }
  memcpy(dst + 12, out, words * 4);
  free(out);
  return 12 + words * 4;
}

static size_t unpack_stream(const uint8_t* src, size_t avail, uint64_t* v, uint32_t count, unsigned w, int* ok)
{
  *ok = 0;
  if (avail < 12) {
    return 0;
  }
  const unsigned bits = get32(src);
  uint64_t mn;
  memcpy(&mn, src + 4, 8);
  if (bits > 64) {
    return 0;
  }
  const size_t words = ((size_t)count * bits + 31) / 32;
  if (avail < 12 + words * 4) {
    return 0;
  }
  const uint64_t vmask = bits == 64 ? ~0ull : ((1ull << bits) - 1);
  for (uint32_t i = 0; i < count; ++i) {
    uint64_t x = 0;
    if (bits) {
      const size_t bit = (size_t)i * bits;
      const size_t k = bit / 32;
      const unsigned sh = (unsigned)(bit % 32);
      x = (uint64_t)get32(src + 12 + 4 * k) >> sh;
      if (sh + bits > 32) {
        x |= (uint64_t)get32(src + 12 + 4 * (k + 1)) << (32 - sh);
      }
      if (sh + bits > 64) {
        x |= (uint64_t)get32(src + 12 + 4 * (k + 2)) << (64 - sh);
      }
      x &= vmask;
    }
    v[i] = (x + mn) & width_mask(w);
  }
  *ok = 1;
  return 12 + words * 4;
}

size_t oracle_cascaded_max_compressed(size_t n_bytes, size_t sub_chunk_bytes, int type)
{
  (void)type;
  if (sub_chunk_bytes == 0) {
    return 0;
  }
  const size_t num_sub = (n_bytes + sub_chunk_bytes - 1) / sub_chunk_bytes;
  /* every sub-chunk falls back to raw (+4 marker, padded to 4) in the worst case */
  return 20 + 4 * num_sub + num_sub * 8 + ((n_bytes + 3) & ~(size_t)3);
}

size_t oracle_cascaded_compress(
    const uint8_t* src, size_t n_bytes, uint8_t* dst, size_t dst_cap, size_t sub_chunk_bytes, int type, int num_rles,
    int num_deltas, int use_bp)
{
  const unsigned w = type_width(type);
  if (w == 0 || sub_chunk_bytes == 0 || sub_chunk_bytes % w || n_bytes % w || sub_chunk_bytes / w > CASC_MAX_ELEMS
      || num_rles < 0 || num_rles > 7 || num_deltas < 0 || num_deltas > 7
      || dst_cap < oracle_cascaded_max_compressed(n_bytes, sub_chunk_bytes, type)) {
    return 0;
  }
  const uint32_t num_sub = (uint32_t)((n_bytes + sub_chunk_bytes - 1) / sub_chunk_bytes);
  put32(dst, CASC_MAGIC);
  dst[4] = (uint8_t)type;
  dst[5] = (uint8_t)num_rles;
  dst[6] = (uint8_t)num_deltas;
  dst[7] = (uint8_t)(use_bp ? 1 : 0);
  put32(dst + 8, (uint32_t)n_bytes);
  put32(dst + 12, (uint32_t)sub_chunk_bytes);
  put32(dst + 16, num_sub);
  uint8_t* table = dst + 20;
  uint8_t* payload = table + 4 * (size_t)num_sub;
  size_t pay = 0;
  const int layers = num_rles > num_deltas ? num_rles : num_deltas;
  uint64_t* vals = (uint64_t*)malloc(sizeof(uint64_t) * CASC_MAX_ELEMS);
  uint64_t* tmp = (uint64_t*)malloc(sizeof(uint64_t) * CASC_MAX_ELEMS);
  uint64_t* runs[8];
  uint32_t counts[8];
  for (int l = 0; l < 8; ++l) {
    runs[l] = (uint64_t*)malloc(sizeof(uint64_t) * CASC_MAX_ELEMS);
  }
  uint8_t* scratch = (uint8_t*)malloc(16 + 9 * (12 + 8 * CASC_MAX_ELEMS + 8));
  for (uint32_t s = 0; s < num_sub; ++s) {
    const size_t off = (size_t)s * sub_chunk_bytes;
    const size_t bytes = n_bytes - off < sub_chunk_bytes ? n_bytes - off : sub_chunk_bytes;
    const uint32_t n = (uint32_t)(bytes / w);
    for (uint32_t i = 0; i < n; ++i) {
      uint64_t x = 0;
      memcpy(&x, src + off + (size_t)i * w, w);
      vals[i] = x;
    }
    uint32_t c = n;
    for (int l = 0; l < layers; ++l) {
      if (l < num_rles) {
        uint32_t m = 0;
        for (uint32_t i = 0; i < c; ++i) {
          if (i == 0 || vals[i] != vals[i - 1]) {
            tmp[m] = vals[i];
            runs[l][m] = 1;
            ++m;
          } else {
            runs[l][m - 1]++;
          }
        }
        /* a layer that would take out fewer than one element in eight is left out (bit-packed streams only, where a
         * stream of ones is just a header): the values pass through, every run length is 1 */
        if (use_bp && m + (c >> 3) > c) {
          for (uint32_t i = 0; i < c; ++i) {
            runs[l][i] = 1;
          }
        } else {
          memcpy(vals, tmp, sizeof(uint64_t) * m);
          c = m;
        }
        counts[l] = c;
      }
      if (l < num_deltas) {
        for (uint32_t i = c; i-- > 1;) {
          vals[i] = (vals[i] - vals[i - 1]) & width_mask(w);
        }
      }
    }
    size_t sz = 0;
    put32(scratch, n);
    sz += 4;
    for (int l = 0; l < num_rles; ++l) {
      put32(scratch + sz, counts[l]);
      sz += 4;
    }
    for (int l = 0; l < num_rles; ++l) {
      sz += pack_stream(scratch + sz, runs[l], counts[l], 2, 0, use_bp, 16);
    }
    sz += pack_stream(scratch + sz, vals, c, w, num_deltas > 0 ? 1 : type_signed(type), use_bp, 8 * w);
    const size_t raw_sz = 4 + ((bytes + 3) & ~(size_t)3);
    if (sz >= raw_sz) {
      put32(payload + pay, 0xffffffffu);
      memset(payload + pay + 4, 0, raw_sz - 4);
      memcpy(payload + pay + 4, src + off, bytes);
      pay += raw_sz;
    } else {
      memcpy(payload + pay, scratch, sz);
      pay += sz;
    }
    put32(table + 4 * (size_t)s, (uint32_t)pay);
  }
  free(vals);
  free(tmp);
  free(scratch);
  for (int l = 0; l < 8; ++l) {
    free(runs[l]);
  }
  return 20 + 4 * (size_t)num_sub + pay;
}

int oracle_cascaded_decompress(const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_cap, size_t* out_len)
{
  *out_len = 0;
  if (src_len < 20 || get32(src) != CASC_MAGIC) {
    return ORACLE_ERR_INPUT;
  }
  const int type = src[4], num_rles = src[5], num_deltas = src[6];
  const unsigned w = type_width(type);
  const size_t n_bytes = get32(src + 8), sub = get32(src + 12);
  const uint32_t num_sub = get32(src + 16);
  if (w == 0 || num_rles > 7 || num_deltas > 7 || sub == 0 || sub % w || n_bytes % w || sub / w > CASC_MAX_ELEMS
      || num_sub != (n_bytes + sub - 1) / sub || src_len < 20 + 4 * (size_t)num_sub) {
    return ORACLE_ERR_INPUT;
  }
  if (n_bytes > dst_cap) {
    return ORACLE_ERR_OUTPUT;
  }
  const uint8_t* table = src + 20;
  const uint8_t* payload = table + 4 * (size_t)num_sub;
  const size_t pay_len = src_len - (size_t)(payload - src);
  const int layers = num_rles > num_deltas ? num_rles : num_deltas;
  uint64_t* vals = (uint64_t*)malloc(sizeof(uint64_t) * CASC_MAX_ELEMS);
  uint64_t* tmp = (uint64_t*)malloc(sizeof(uint64_t) * CASC_MAX_ELEMS);
  uint64_t* runs[8];
  for (int l = 0; l < 8; ++l) {
    runs[l] = (uint64_t*)malloc(sizeof(uint64_t) * CASC_MAX_ELEMS);
  }
  int rc = ORACLE_OK;
  size_t begin = 0;
  for (uint32_t s = 0; s < num_sub && rc == ORACLE_OK; ++s) {
    const size_t end = get32(table + 4 * (size_t)s);
    const size_t off = (size_t)s * sub;
    const size_t bytes = n_bytes - off < sub ? n_bytes - off : sub;
    const uint32_t n = (uint32_t)(bytes / w);
    if (end < begin || end > pay_len || end - begin < 4) {
      rc = ORACLE_ERR_INPUT;
      break;
    }
    const uint8_t* p = payload + begin;
    size_t avail = end - begin;
    const uint32_t first = get32(p);
    if (first == 0xffffffffu) {
      if (avail < 4 + bytes) {
        rc = ORACLE_ERR_INPUT;
        break;
      }
      memcpy(dst + off, p + 4, bytes);
      begin = end;
      continue;
    }
    if (first != n || avail < 4 + 4 * (size_t)num_rles) {
      rc = ORACLE_ERR_INPUT;
      break;
    }
    uint32_t counts[8];
    size_t pos = 4;
    uint32_t prev = n;
    for (int l = 0; l < num_rles; ++l) {
      counts[l] = get32(p + pos);
      pos += 4;
      if (counts[l] > prev || (counts[l] == 0 && prev != 0)) {
        rc = ORACLE_ERR_INPUT;
      }
      prev = counts[l];
    }
    if (rc != ORACLE_OK) {
      break;
    }
    int ok = 1;
    for (int l = 0; l < num_rles && ok; ++l) {
      pos += unpack_stream(p + pos, avail - pos, runs[l], counts[l], 2, &ok);
    }
    uint32_t c = num_rles ? counts[num_rles - 1] : n;
    if (ok) {
      pos += unpack_stream(p + pos, avail - pos, vals, c, w, &ok);
    }
    if (!ok) {
      rc = ORACLE_ERR_INPUT;
      break;
    }
    for (int l = layers - 1; l >= 0 && rc == ORACLE_OK; --l) {
      if (l < num_deltas) {
        for (uint32_t i = 1; i < c; ++i) {
          vals[i] = (vals[i] + vals[i - 1]) & width_mask(w);
        }
      }
      if (l < num_rles) {
        const uint32_t target = l == 0 ? n : counts[l - 1];
        uint32_t m = 0;
        for (uint32_t i = 0; i < c; ++i) {
          const uint64_t r = runs[l][i];
          if (r == 0 || r > target - m) {
            rc = ORACLE_ERR_INPUT;
            break;
          }
          for (uint64_t k = 0; k < r; ++k) {
            tmp[m++] = vals[i];
          }
        }
        if (rc == ORACLE_OK && m != target) {
          rc = ORACLE_ERR_INPUT;
        }
        memcpy(vals, tmp, sizeof(uint64_t) * m);
        c = m;
      }
    }
    if (rc == ORACLE_OK) {
      for (uint32_t i = 0; i < n; ++i) {
        memcpy(dst + off + (size_t)i * w, &vals[i], w);
      }
    }
    begin = end;
  }
  free(vals);
  free(tmp);
  for (int l = 0; l < 8; ++l) {
    free(runs[l]);
  }
  if (rc == ORACLE_OK) {
    *out_len = n_bytes;
  }
  return rc;
}
