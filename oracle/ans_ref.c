/*
 * oracle/ans_ref.c -- CPU model of this library's ANS stream. TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED: the reference's ANS bitstream is closed and absent from its tree
 * (/root/reference/README.md:10,17; only the call sites at
 * /root/reference/benchmarks/benchmark_ans_chunked.cu:29-81 are visible: byte data, one
 * format type, chunks below 2^32 bytes, lossless round trip). This file restates the layout
 * of nvcomp_amd/csrc/ans/ans.hip.h with plain scalar loops over the 64 interleaved rANS
 * states so that tests can demand byte-identical streams from the HIP compressor and
 * decode them without the GPU code. The coder itself is the published range-ANS
 * construction (J. Duda, "Asymmetric numeral systems", arXiv:1311.2540; state update
 * x' = (x / f) * M + x mod f + start, byte-wise renormalisation generalised to 16-bit words).
 */
#include "oracle.h"

#include <string.h>

#define ANS_PROB_BITS 10
#define ANS_SCALE (1u << ANS_PROB_BITS)
#define ANS_LOW (1u << 16)
#define ANS_HEADER 12
#define ANS_FREQ_OFF 16
#define ANS_STATE_OFF (ANS_FREQ_OFF + 512)
#define ANS_WORDS_OFF (ANS_STATE_OFF + 512) /* 64 A states (even groups), then 64 B states (odd groups) */
#define ANS_MIN_CODED 2048

size_t oracle_ans_max_compressed(size_t n)
{
  return (n + ANS_HEADER + 7) & ~(size_t)7;
}

static void ans_header(uint8_t* dst, size_t n, int mode)
{
  const uint32_t n32 = (uint32_t)n;
  dst[0] = 'A';
  dst[1] = 'N';
  dst[2] = 'S';
  dst[3] = 1;
  memcpy(dst + 4, &n32, 4);
  dst[8] = (uint8_t)mode;
  dst[9] = dst[10] = dst[11] = 0;
}

static size_t ans_store(const uint8_t* src, size_t n, uint8_t* dst)
{
  ans_header(dst, n, 0);
  memcpy(dst + ANS_HEADER, src, n);
  return ANS_HEADER + n;
}

static void ans_normalise(const uint32_t* count, size_t n, uint32_t* freq)
{
  uint32_t sum = 0;
  for (int s = 0; s < 256; ++s) {
    const uint32_t q = (uint32_t)(((uint64_t)count[s] << ANS_PROB_BITS) / n);
    freq[s] = count[s] == 0 ? 0 : (q == 0 ? 1 : q);
    sum += freq[s];
  }
  while (sum != ANS_SCALE) {
    int top = 0;
    for (int s = 1; s < 256; ++s) {
      if (freq[s] > freq[top]) { /* strict: the lowest index wins ties */
        top = s;
      }
    }
    if (sum < ANS_SCALE) {
      freq[top] += ANS_SCALE - sum;
      sum = ANS_SCALE;
    } else {
      const uint32_t excess = sum - ANS_SCALE;
      const uint32_t take = excess < freq[top] - 1 ? excess : freq[top] - 1;
      freq[top] -= take;
      sum -= take;
    }
  }
}

/* Symbol index of (pair of groups q, state h: 0 = A = even group / 1 = B = odd group, lane l, byte r of the lane's
 * dword). A row of the coder is (q, r): first the 64 A symbols, then the 64 B symbols. */
static size_t ans_index(size_t q, unsigned h, unsigned l, unsigned r)
{
  return 256 * (2 * q + h) + 4 * l + r;
}

size_t oracle_ans_compress(const uint8_t* src, size_t n, uint8_t* dst, size_t dst_cap)
{
  if (n > 0xffffffffu || dst_cap < oracle_ans_max_compressed(n)) {
    return 0;
  }
  if (n < ANS_MIN_CODED) {
    return ans_store(src, n, dst);
  }
  uint32_t count[256] = {0}, freq[256], start[257];
  for (size_t i = 0; i < n; ++i) {
    ++count[src[i]];
  }
  ans_normalise(count, n, freq);
  start[0] = 0;
  for (int s = 0; s < 256; ++s) {
    start[s + 1] = start[s] + freq[s];
  }
  const size_t limit_words = (n + ANS_HEADER - ANS_WORDS_OFF) / 2;
  uint16_t* words = (uint16_t*)(dst + ANS_WORDS_OFF); /* dst + 784: 2-byte aligned whenever dst is */
  uint32_t x[2][64];
  for (unsigned l = 0; l < 64; ++l) {
    x[0][l] = x[1][l] = ANS_LOW;
  }
  size_t p = 0;
  const size_t groups = (n + 255) / 256;
  const size_t pairs = (groups + 1) / 2;
  for (size_t q = pairs; q-- > 0;) {
    for (unsigned rr = 0; rr < 4; ++rr) {
      const unsigned r = 3 - rr;
      /* the states that renormalise in this row append their words: A's in lane order, then B's */
      unsigned cnt = 0;
      for (unsigned h = 0; h < 2; ++h) {
        for (unsigned l = 0; l < 64; ++l) {
          const size_t i = ans_index(q, h, l, r);
          if (i < n && (x[h][l] >> (32 - ANS_PROB_BITS)) >= freq[src[i]]) {
            ++cnt;
          }
        }
      }
      if (p + cnt >= limit_words) {
        return ans_store(src, n, dst);
      }
      for (unsigned h = 0; h < 2; ++h) {
        for (unsigned l = 0; l < 64; ++l) {
          const size_t i = ans_index(q, h, l, r);
          if (i >= n) {
            continue;
          }
          const uint32_t f = freq[src[i]];
          if ((x[h][l] >> (32 - ANS_PROB_BITS)) >= f) {
            const uint16_t wv = (uint16_t)x[h][l];
            memcpy((uint8_t*)words + 2 * p, &wv, 2);
            ++p;
            x[h][l] >>= 16;
          }
          x[h][l] = ((x[h][l] / f) << ANS_PROB_BITS) + (x[h][l] % f) + start[src[i]];
        }
      }
    }
  }
  ans_header(dst, n, 1);
  const uint32_t p32 = (uint32_t)p;
  memcpy(dst + 12, &p32, 4);
  for (int s = 0; s < 256; ++s) {
    const uint16_t f16 = (uint16_t)freq[s];
    memcpy(dst + ANS_FREQ_OFF + 2 * s, &f16, 2);
  }
  memcpy(dst + ANS_STATE_OFF, x, 512);
  return ANS_WORDS_OFF + 2 * p;
}

int oracle_ans_decompress(const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_cap, size_t* out_len)
{
  *out_len = 0;
  if (src_len < ANS_HEADER || src[0] != 'A' || src[1] != 'N' || src[2] != 'S' || src[3] != 1 || src[8] > 1 || src[9] != 0
      || src[10] != 0 || src[11] != 0) {
    return ORACLE_ERR_INPUT;
  }
  uint32_t n32;
  memcpy(&n32, src + 4, 4);
  const size_t n = n32;
  if (n > dst_cap) {
    return ORACLE_ERR_OUTPUT;
  }
  if (src[8] == 0) {
    if (src_len - ANS_HEADER < n) {
      return ORACLE_ERR_INPUT;
    }
    memcpy(dst, src + ANS_HEADER, n);
    *out_len = n;
    return ORACLE_OK;
  }
  if (src_len < ANS_WORDS_OFF) {
    return ORACLE_ERR_INPUT;
  }
  uint32_t n_words;
  memcpy(&n_words, src + 12, 4);
  if ((src_len - ANS_WORDS_OFF) / 2 < n_words) {
    return ORACLE_ERR_INPUT;
  }
  uint32_t freq[256], start[257], sum = 0;
  for (int s = 0; s < 256; ++s) {
    uint16_t f16;
    memcpy(&f16, src + ANS_FREQ_OFF + 2 * s, 2);
    freq[s] = f16;
    sum += f16;
  }
  if (sum != ANS_SCALE) {
    return ORACLE_ERR_INPUT;
  }
  start[0] = 0;
  for (int s = 0; s < 256; ++s) {
    start[s + 1] = start[s] + freq[s];
  }
  uint8_t sym_of[ANS_SCALE];
  for (int s = 0; s < 256; ++s) {
    for (uint32_t k = start[s]; k < start[s + 1]; ++k) {
      sym_of[k] = (uint8_t)s;
    }
  }
  uint32_t x[2][64];
  memcpy(x, src + ANS_STATE_OFF, 512);
  const uint8_t* words = src + ANS_WORDS_OFF;
  size_t p = n_words;
  const size_t groups = (n + 255) / 256;
  const size_t pairs = (groups + 1) / 2;
  for (size_t q = 0; q < pairs; ++q) {
    for (unsigned r = 0; r < 4; ++r) {
      uint32_t nx[2][64];
      unsigned cnt = 0;
      for (unsigned h = 0; h < 2; ++h) {
        for (unsigned l = 0; l < 64; ++l) {
          const size_t i = ans_index(q, h, l, r);
          if (i >= n) {
            continue;
          }
          const uint32_t slot = x[h][l] & (ANS_SCALE - 1);
          const uint8_t s = sym_of[slot];
          dst[i] = s;
          nx[h][l] = freq[s] * (x[h][l] >> ANS_PROB_BITS) + slot - start[s];
          if (nx[h][l] < ANS_LOW) {
            ++cnt;
          }
        }
      }
      if (cnt > p) {
        return ORACLE_ERR_INPUT;
      }
      p -= cnt;
      unsigned k = 0;
      for (unsigned h = 0; h < 2; ++h) {
        for (unsigned l = 0; l < 64; ++l) {
          if (ans_index(q, h, l, r) >= n) {
            continue;
          }
          if (nx[h][l] < ANS_LOW) {
            uint16_t wv;
            memcpy(&wv, words + 2 * (p + k), 2);
            nx[h][l] = (nx[h][l] << 16) | wv;
            ++k;
          }
          x[h][l] = nx[h][l];
        }
      }
    }
  }
  if (p != 0) {
    return ORACLE_ERR_INPUT;
  }
  for (unsigned l = 0; l < 64; ++l) {
    if (x[0][l] != ANS_LOW || x[1][l] != ANS_LOW) {
      return ORACLE_ERR_INPUT;
    }
  }
  *out_len = n;
  return ORACLE_OK;
}
