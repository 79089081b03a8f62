/*
 * oracle/ref_shim.c -> oracle/_ref/libcpucodecs.so
 *
 * Thin C shim over the third-party CPU codecs the reference itself uses as its
 * interoperability pin and CPU peer: liblz4 (lz4.h, lz4hc.h: examples/
 * lz4_cpu_compression.cu:31-33,61-66; examples/lz4_cpu_decompression.cu:143-147)
 * and snappy (named by BASELINE.json north_star). The libraries are the
 * container's own (/opt/conda: lz4 1.9.3, snappy 1.1.8); nothing from
 * /root/reference is compiled or copied here because the reference ships no
 * codec source (README.md:10) -- its codec path is "unbuildable" and these
 * libraries are the published implementations of the same wire formats. zlib (examples/deflate_cpu_compression.cu:69-104,
 * examples/deflate_cpu_decompression.cu:128-170: the CPU peer of the DEFLATE path) is bound the same way, and so is
 * libdeflate (/opt/conda: 1.8), the reference's algo 0 of the same examples (deflate_cpu_compression.cu:60-67:
 * libdeflate_alloc_compressor(6), libdeflate_deflate_compress; deflate_cpu_decompression.cu: libdeflate_deflate_decompress)
 * and the CPU peer BASELINE.json's north_star names beside liblz4.
 *
 * TEST INFRASTRUCTURE ONLY: used to pin the oracle sources, to make golden vectors
 * (scripts/make_golden.py), to prepare CPU-compressed inputs for tests and
 * bench.py, and as bench.py's cpu_baseline ("reference" kind).
 */
#ifdef HAVE_LIBDEFLATE
#include <libdeflate.h>
#endif
#include <lz4.h>
#include <lz4hc.h>
#include <snappy-c.h>
#include <string.h>
#include <zlib.h>

#include "batch.h"

int ref_lz4_version(void) { return LZ4_versionNumber(); }
size_t ref_lz4_bound(size_t n) { return (size_t)LZ4_compressBound((int)n); }
size_t ref_snappy_bound(size_t n) { return snappy_max_compressed_length(n); }

static int hc_level = 12;

static int r_lz4_dec(const uint8_t* s, size_t n, uint8_t* d, size_t cap, size_t* out)
{
  if (n == 0) {
    *out = 0;
    return 0;
  }
  const int r = LZ4_decompress_safe((const char*)s, (char*)d, (int)n, (int)cap);
  *out = r < 0 ? 0 : (size_t)r;
  return r < 0;
}
static int r_lz4_enc(const uint8_t* s, size_t n, uint8_t* d, size_t cap, size_t* out)
{
  const int r = LZ4_compress_default((const char*)s, (char*)d, (int)n, (int)cap);
  *out = (size_t)r;
  return r <= 0 && n != 0;
}
static int r_lz4_enc_hc(const uint8_t* s, size_t n, uint8_t* d, size_t cap, size_t* out)
{
  const int r = LZ4_compress_HC((const char*)s, (char*)d, (int)n, (int)cap, hc_level);
  *out = (size_t)r;
  return r <= 0 && n != 0;
}
static int r_snappy_dec(const uint8_t* s, size_t n, uint8_t* d, size_t cap, size_t* out)
{
  size_t len = cap;
  const snappy_status st = snappy_uncompress((const char*)s, n, (char*)d, &len);
  *out = st == SNAPPY_OK ? len : 0;
  return st != SNAPPY_OK;
}
static int r_snappy_enc(const uint8_t* s, size_t n, uint8_t* d, size_t cap, size_t* out)
{
  size_t len = cap;
  const snappy_status st = snappy_compress((const char*)s, n, (char*)d, &len);
  *out = st == SNAPPY_OK ? len : 0;
  return st != SNAPPY_OK;
}

/* single-chunk entry points (return 0 on success) */
int ref_lz4_decompress(const uint8_t* s, size_t n, uint8_t* d, size_t cap, size_t* out) { return r_lz4_dec(s, n, d, cap, out); }
int ref_lz4_compress(const uint8_t* s, size_t n, uint8_t* d, size_t cap, size_t* out) { return r_lz4_enc(s, n, d, cap, out); }
int ref_lz4_compress_hc(const uint8_t* s, size_t n, uint8_t* d, size_t cap, int level, size_t* out)
{
  const int r = LZ4_compress_HC((const char*)s, (char*)d, (int)n, (int)cap, level);
  *out = (size_t)r;
  return r <= 0 && n != 0;
}
int ref_snappy_decompress(const uint8_t* s, size_t n, uint8_t* d, size_t cap, size_t* out) { return r_snappy_dec(s, n, d, cap, out); }
int ref_snappy_compress(const uint8_t* s, size_t n, uint8_t* d, size_t cap, size_t* out) { return r_snappy_enc(s, n, d, cap, out); }
int ref_snappy_uncompressed_length(const uint8_t* s, size_t n, size_t* out)
{
  return snappy_uncompressed_length((const char*)s, n, out) != SNAPPY_OK;
}

/* raw DEFLATE streams (windowBits -15), as the reference's examples make and read them. The stream states are kept
 * per thread and reset per chunk (a fresh deflateInit2 per 64 KiB chunk spends most of its time in malloc when 256
 * threads do it at once): the figure that favours the CPU. */
static int r_zlib_inflate(const uint8_t* s, size_t n, uint8_t* d, size_t cap, size_t* out)
{
  static __thread z_stream zs;
  static __thread int ready = 0;
  *out = 0;
  if (!ready) {
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, -15) != Z_OK) {
      return 1;
    }
    ready = 1;
  } else {
    inflateReset(&zs);
  }
  zs.next_in = (Bytef*)s;
  zs.avail_in = (uInt)n;
  zs.next_out = d;
  zs.avail_out = (uInt)cap;
  const int r = inflate(&zs, Z_FINISH);
  *out = r == Z_STREAM_END ? zs.total_out : 0;
  return r != Z_STREAM_END;
}
static int zlib_deflate_level(const uint8_t* s, size_t n, uint8_t* d, size_t cap, size_t* out, int level)
{
  static __thread z_stream zs[10];
  static __thread int ready[10];
  *out = 0;
  if (!ready[level]) {
    memset(&zs[level], 0, sizeof(z_stream));
    if (deflateInit2(&zs[level], level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) {
      return 1;
    }
    ready[level] = 1;
  } else {
    deflateReset(&zs[level]);
  }
  z_stream* z = &zs[level];
  z->next_in = (Bytef*)s;
  z->avail_in = (uInt)n;
  z->next_out = d;
  z->avail_out = (uInt)cap;
  const int r = deflate(z, Z_FINISH);
  *out = r == Z_STREAM_END ? z->total_out : 0;
  return r != Z_STREAM_END;
}
static int r_zlib_deflate1(const uint8_t* s, size_t n, uint8_t* d, size_t cap, size_t* out) { return zlib_deflate_level(s, n, d, cap, out, 1); }
static int r_zlib_deflate9(const uint8_t* s, size_t n, uint8_t* d, size_t cap, size_t* out) { return zlib_deflate_level(s, n, d, cap, out, 9); }

#ifdef HAVE_LIBDEFLATE
/* libdeflate, raw DEFLATE streams; one compressor / decompressor per thread, kept (the reference's example allocates one
 * per chunk: the figure that favours the CPU again) */
static int r_libdeflate_dec(const uint8_t* s, size_t n, uint8_t* d, size_t cap, size_t* out)
{
  static __thread struct libdeflate_decompressor* dec = NULL;
  *out = 0;
  if (dec == NULL && (dec = libdeflate_alloc_decompressor()) == NULL) {
    return 1;
  }
  size_t got = 0;
  const enum libdeflate_result r = libdeflate_deflate_decompress(dec, s, n, d, cap, &got);
  *out = r == LIBDEFLATE_SUCCESS ? got : 0;
  return r != LIBDEFLATE_SUCCESS;
}
static int r_libdeflate_enc6(const uint8_t* s, size_t n, uint8_t* d, size_t cap, size_t* out)
{
  static __thread struct libdeflate_compressor* enc = NULL;
  *out = 0;
  if (enc == NULL && (enc = libdeflate_alloc_compressor(6)) == NULL) { /* level 6: deflate_cpu_compression.cu:62 */
    return 1;
  }
  const size_t got = libdeflate_deflate_compress(enc, s, n, d, cap);
  *out = got;
  return got == 0 && n != 0;
}
int ref_libdeflate_decompress(const uint8_t* s, size_t n, uint8_t* d, size_t cap, size_t* out) { return r_libdeflate_dec(s, n, d, cap, out); }
int ref_libdeflate_compress(const uint8_t* s, size_t n, uint8_t* d, size_t cap, size_t* out) { return r_libdeflate_enc6(s, n, d, cap, out); }
size_t ref_libdeflate_bound(size_t n) { return libdeflate_deflate_compress_bound(NULL, n); }
#else /* built without libdeflate: codecs 8 and 9 are absent and the ref_libdeflate_* symbols with them (oracle_py.have_libdeflate) */
#define r_libdeflate_dec NULL
#define r_libdeflate_enc6 NULL
#endif

/* codec: 0 lz4 dec, 1 snappy dec, 2 lz4 enc (default), 3 snappy enc, 4 lz4 enc HC level 12, 5 zlib inflate (raw),
 * 6 zlib deflate level 1 (raw), 7 zlib deflate level 9 (raw), 8 libdeflate decompress (raw), 9 libdeflate compress level 6 */
double ref_batch_run(
    int codec, int threads, int repeats, size_t n_chunks,
    const uint8_t* const* in_ptrs, const size_t* in_sizes,
    uint8_t* const* out_ptrs, const size_t* out_caps, size_t* out_sizes, int* errors)
{
  static const batch_codec_fn table[10] = {r_lz4_dec, r_snappy_dec, r_lz4_enc, r_snappy_enc, r_lz4_enc_hc,
                                           r_zlib_inflate, r_zlib_deflate1, r_zlib_deflate9, r_libdeflate_dec, r_libdeflate_enc6};
  if (codec < 0 || codec > 9 || table[codec] == NULL) {
    return -1.0;
  }
  return batch_run_generic(
      table[codec], threads, repeats, n_chunks, in_ptrs, in_sizes, out_ptrs, out_caps, out_sizes, errors);
}
