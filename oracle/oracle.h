/*
 * oracle/oracle.h -- CPU oracle for the batched codecs (TEST INFRASTRUCTURE ONLY).
 * See the per-file headers for what each function restates and how it is pinned.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  ORACLE_OK = 0,
  ORACLE_ERR_INPUT = 1,  /* compressed stream truncated / malformed */
  ORACLE_ERR_OUTPUT = 2, /* output capacity too small */
  ORACLE_ERR_OFFSET = 3  /* match offset 0 or beyond produced output */
};

/* LZ4 block format (oracle/lz4_block.c) */
int oracle_lz4_decompress(const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_cap, size_t* out_len);
size_t oracle_lz4_decompressed_size(const uint8_t* src, size_t src_len);
size_t oracle_lz4_compress_bound(size_t n);
size_t oracle_lz4_compress(const uint8_t* src, size_t n, uint8_t* dst, size_t dst_cap);

/* Snappy raw format (oracle/snappy_raw.c) */
int oracle_snappy_decompress(const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_cap, size_t* out_len);
int oracle_snappy_decompressed_size(const uint8_t* src, size_t src_len, size_t* out_len);
size_t oracle_snappy_compress_bound(size_t n);
size_t oracle_snappy_compress(const uint8_t* src, size_t n, uint8_t* dst, size_t dst_cap);

/* Cascaded (oracle/cascaded_ref.c; own container, parity unpinned) */
size_t oracle_cascaded_max_compressed(size_t n_bytes, size_t sub_chunk_bytes, int type);
size_t oracle_cascaded_compress(const uint8_t* src, size_t n_bytes, uint8_t* dst, size_t dst_cap, size_t sub_chunk_bytes,
                                int type, int num_rles, int num_deltas, int use_bp);
int oracle_cascaded_decompress(const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_cap, size_t* out_len);

/* Bitcomp (oracle/bitcomp_ref.c; own stream, parity unpinned). elem_size in {1,2,4,8}; algo 0 = delta, 1 = plain. */
size_t oracle_bitcomp_max_compressed(size_t n_bytes, int elem_size);
size_t oracle_bitcomp_compress(const uint8_t* src, size_t n_bytes, uint8_t* dst, size_t dst_cap, int algo, int elem_size);
int oracle_bitcomp_decompress(const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_cap, size_t* out_len);

/* ANS (oracle/ans_ref.c; own stream, parity unpinned) */
size_t oracle_ans_max_compressed(size_t n_bytes);
size_t oracle_ans_compress(const uint8_t* src, size_t n_bytes, uint8_t* dst, size_t dst_cap);
int oracle_ans_decompress(const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_cap, size_t* out_len);

/* Batched, threaded drivers used for the cpu_baseline timing (oracle/batch.c).
 * codec: 0 = lz4 decompress, 1 = snappy decompress, 2 = lz4 compress, 3 = snappy compress,
 * 4 / 5 / 6 = cascaded / bitcomp / ans decompress (the own-stream CPU models).
 * Returns wall seconds of the best of `repeats` runs; per-chunk result sizes in out_sizes. */
double oracle_batch_run(
    int codec, int threads, int repeats, size_t n_chunks,
    const uint8_t* const* in_ptrs, const size_t* in_sizes,
    uint8_t* const* out_ptrs, const size_t* out_caps, size_t* out_sizes, int* errors);

#ifdef __cplusplus
}
#endif
#endif
