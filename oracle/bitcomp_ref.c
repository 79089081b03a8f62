/*
 * oracle/bitcomp_ref.c -- CPU model of this library's Bitcomp stream. TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED: the reference's Bitcomp bitstream is closed and undocumented
 * (/root/reference/README.md:13 -- its decompressor only accepts its own compressor's
 * output), its tree holds no Bitcomp fixture, and nothing here can be checked against
 * the reference beyond the API behaviour visible at
 * /root/reference/benchmarks/benchmark_bitcomp_chunked.cu:32-127 (options
 * {algorithm_type 0|1, data_type 0..7}; chunk sizes multiples of the element size;
 * lossless round trip). This file is a scalar, independent restatement of the layout in
 * nvcomp_amd/csrc/bitcomp/bitcomp.hip.h (written element by element with a bit cursor
 * per lane instead of the kernels' register streaming) so that tests can demand
 * byte-identical streams from the HIP compressor and decode them without the GPU code.
 */
#include "oracle.h"

#include <string.h>

#define BC_HEADER 12
#define BC_ROWS 32

/* elements of one lane in one row: a dword's worth for 1- and 2-byte elements */
static unsigned bc_lane_elems(unsigned s)
{
  return s < 4 ? 4 / s : 1;
}

static uint64_t bc_load(const uint8_t* p, unsigned s)
{
  uint64_t v = 0;
  memcpy(&v, p, s); /* little endian host */
  return v;
}

static void bc_store(uint8_t* p, uint64_t v, unsigned s)
{
  memcpy(p, &v, s);
}

static uint64_t bc_mask(unsigned bits)
{
  return bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
}

static unsigned bc_width(uint64_t v)
{
  unsigned w = 0;
  while (v) {
    ++w;
    v >>= 1;
  }
  return w;
}

static unsigned bc_pad4(unsigned n)
{
  return (n + 3u) & ~3u;
}

static size_t bc_block_bound(size_t rows, size_t s)
{
  return ((rows + 3) & ~(size_t)3) + (rows * 8 * s * bc_lane_elems((unsigned)s) + 31) / 32 * 256;
}

size_t oracle_bitcomp_max_compressed(size_t n, int elem_size)
{
  const size_t s = (size_t)elem_size;
  const size_t nelem = n / s;
  const size_t row_elems = 64 * bc_lane_elems((unsigned)s);
  const size_t block = row_elems * BC_ROWS;
  const size_t full = nelem / block, rest = nelem % block;
  return BC_HEADER + full * bc_block_bound(BC_ROWS, s) + (rest ? bc_block_bound((rest + row_elems - 1) / row_elems, s) : 0)
         + n % s;
}

/* value of element i as it is packed: algo 0 = zigzag(e[i] - e[i-1]) in W bits, algo 1 = e[i] */
static uint64_t bc_value(const uint8_t* src, size_t i, unsigned s, int algo)
{
  const unsigned w = 8 * s;
  const uint64_t e = bc_load(src + i * s, s);
  if (algo != 0) {
    return e;
  }
  const uint64_t prev = i ? bc_load(src + (i - 1) * s, s) : 0;
  const uint64_t d = (e - prev) & bc_mask(w);
  const uint64_t sign = (d >> (w - 1)) & 1;
  return ((d << 1) ^ (sign ? ~0ull : 0ull)) & bc_mask(w);
}

/* set `bits` bits of `v` at bit position `pos` of lane `lane`'s bit string inside `payload` */
/* The lane's bit string lives in its dwords: dword d of lane l at payload + (64 d + l) * 4, little endian, bit b of
 * the string = bit b % 32 of dword b / 32. A value of up to 64 bits touches at most three of them. */
static uint32_t bc_dword(const uint8_t* payload, unsigned lane, uint64_t d)
{
  const uint8_t* p = payload + (d * 64 + lane) * 4;
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

static void bc_or_dword(uint8_t* payload, unsigned lane, uint64_t d, uint32_t x)
{
  uint8_t* p = payload + (d * 64 + lane) * 4;
  p[0] |= (uint8_t)x;
  p[1] |= (uint8_t)(x >> 8);
  p[2] |= (uint8_t)(x >> 16);
  p[3] |= (uint8_t)(x >> 24);
}

static void bc_put(uint8_t* payload, unsigned lane, uint64_t pos, uint64_t v, unsigned bits)
{
  if (bits == 0) {
    return;
  }
  v &= bc_mask(bits);
  const uint64_t d = pos / 32;
  const unsigned sh = (unsigned)(pos % 32);
  bc_or_dword(payload, lane, d, (uint32_t)(v << sh));
  if (sh + bits > 32) {
    bc_or_dword(payload, lane, d + 1, (uint32_t)(v >> (32 - sh)));
  }
  if (sh + bits > 64) {
    bc_or_dword(payload, lane, d + 2, (uint32_t)(v >> (64 - sh)));
  }
}

static uint64_t bc_get(const uint8_t* payload, unsigned lane, uint64_t pos, unsigned bits)
{
  if (bits == 0) {
    return 0;
  }
  const uint64_t d = pos / 32;
  const unsigned sh = (unsigned)(pos % 32);
  uint64_t v = (uint64_t)bc_dword(payload, lane, d) >> sh;
  if (sh + bits > 32) {
    v |= (uint64_t)bc_dword(payload, lane, d + 1) << (32 - sh);
  }
  if (sh + bits > 64) { /* sh > 0 here */
    v |= (uint64_t)bc_dword(payload, lane, d + 2) << (64 - sh);
  }
  return v & bc_mask(bits);
}

size_t oracle_bitcomp_compress(const uint8_t* src, size_t n, uint8_t* dst, size_t dst_cap, int algo, int elem_size)
{
  const unsigned s = (unsigned)elem_size;
  if ((s != 1 && s != 2 && s != 4 && s != 8) || (algo != 0 && algo != 1) || n > 0xffffffffu
      || dst_cap < oracle_bitcomp_max_compressed(n, elem_size)) {
    return 0;
  }
  const size_t nelem = n / s;
  dst[0] = 'B';
  dst[1] = 'T';
  dst[2] = 'C';
  dst[3] = 1;
  dst[4] = (uint8_t)algo;
  dst[5] = (uint8_t)(s == 1 ? 0 : s == 2 ? 1 : s == 4 ? 2 : 3);
  dst[6] = 0;
  dst[7] = 0;
  bc_store(dst + 8, n, 4);
  const unsigned E = bc_lane_elems(s);
  const size_t row_elems = 64 * (size_t)E;
  const size_t block = row_elems * BC_ROWS;
  size_t op = BC_HEADER;
  for (size_t base = 0; base < nelem; base += block) {
    const size_t count = nelem - base < block ? nelem - base : block;
    const unsigned rows = (unsigned)((count + row_elems - 1) / row_elems);
    unsigned widths[BC_ROWS];
    uint64_t total = 0;
    for (unsigned r = 0; r < rows; ++r) {
      unsigned w = 0;
      for (size_t j = 0; j < row_elems; ++j) {
        const size_t i = base + row_elems * r + j;
        if (i < nelem) {
          const unsigned x = bc_width(bc_value(src, i, s, algo));
          w = x > w ? x : w;
        }
      }
      widths[r] = w;
      total += (uint64_t)w * E; /* bits of one lane's string */
    }
    if (total == 0) { /* zero block marker */
      dst[op] = 0xFF;
      dst[op + 1] = dst[op + 2] = dst[op + 3] = 0;
      op += 4;
      continue;
    }
    const unsigned wbytes = bc_pad4(rows);
    for (unsigned r = 0; r < wbytes; ++r) {
      dst[op + r] = (uint8_t)(r < rows ? widths[r] : 0);
    }
    uint8_t* payload = dst + op + wbytes;
    const size_t dwords = (size_t)((total + 31) / 32);
    memset(payload, 0, dwords * 256);
    for (unsigned l = 0; l < 64; ++l) {
      uint64_t pos = 0;
      for (unsigned r = 0; r < rows; ++r) {
        for (unsigned k = 0; k < E; ++k) { /* the lane's E consecutive elements of this row */
          const size_t i = base + row_elems * r + (size_t)E * l + k;
          if (i < nelem) {
            bc_put(payload, l, pos, bc_value(src, i, s, algo), widths[r]);
          }
          pos += widths[r];
        }
      }
    }
    op += wbytes + dwords * 256;
  }
  const size_t tail = n - nelem * s;
  memcpy(dst + op, src + nelem * s, tail);
  return op + tail;
}

int oracle_bitcomp_decompress(const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_cap, size_t* out_len)
{
  *out_len = 0;
  if (src_len < BC_HEADER || src[0] != 'B' || src[1] != 'T' || src[2] != 'C' || src[3] != 1 || src[4] > 1 || src[5] > 3
      || src[6] != 0 || src[7] != 0) {
    return ORACLE_ERR_INPUT;
  }
  const int algo = src[4];
  const unsigned s = 1u << src[5];
  const unsigned w = 8 * s;
  const size_t n = (size_t)bc_load(src + 8, 4);
  if (n > dst_cap) {
    return ORACLE_ERR_OUTPUT;
  }
  const size_t nelem = n / s;
  const unsigned E = bc_lane_elems(s);
  const size_t row_elems = 64 * (size_t)E;
  const size_t block = row_elems * BC_ROWS;
  size_t ip = BC_HEADER;
  uint64_t prev = 0;
  for (size_t base = 0; base < nelem; base += block) {
    const size_t count = nelem - base < block ? nelem - base : block;
    const unsigned rows = (unsigned)((count + row_elems - 1) / row_elems);
    if (src_len < ip || src_len - ip < 4) {
      return ORACLE_ERR_INPUT;
    }
    const int zero_block = src[ip] == 0xFF;
    const unsigned wbytes = zero_block ? 4 : bc_pad4(rows);
    if (src_len - ip < wbytes) {
      return ORACLE_ERR_INPUT;
    }
    uint64_t total = 0;
    uint64_t row_pos[BC_ROWS];
    uint8_t wd[BC_ROWS];
    for (unsigned r = 0; r < rows; ++r) {
      wd[r] = zero_block ? 0 : src[ip + r];
      if (wd[r] > w) {
        return ORACLE_ERR_INPUT;
      }
      row_pos[r] = total;
      total += (uint64_t)wd[r] * E;
    }
    const size_t dwords = (size_t)((total + 31) / 32);
    if ((src_len - ip - wbytes) / 256 < dwords) {
      return ORACLE_ERR_INPUT;
    }
    const uint8_t* payload = src + ip + wbytes;
    for (size_t j = 0; j < count; ++j) { /* element order */
      const unsigned r = (unsigned)(j / row_elems), l = (unsigned)((j % row_elems) / E), k = (unsigned)(j % E);
      const uint64_t v = bc_get(payload, l, row_pos[r] + (uint64_t)k * wd[r], wd[r]);
      uint64_t e = v;
      if (algo == 0) {
        const uint64_t d = (v >> 1) ^ ((v & 1) ? ~0ull : 0ull);
        e = (prev + d) & bc_mask(w);
        prev = e;
      }
      bc_store(dst + (base + j) * s, e, s);
    }
    ip += wbytes + dwords * 256;
  }
  const size_t tail = n - nelem * s;
  if (src_len < ip || src_len - ip < tail) {
    return ORACLE_ERR_INPUT;
  }
  memcpy(dst + nelem * s, src + ip, tail);
  *out_len = n;
  return ORACLE_OK;
}
