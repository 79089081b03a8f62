/* oracle/batch.h -- threaded batch driver shared by the port and the liblz4/snappy shim.
 * TEST INFRASTRUCTURE ONLY. */
#ifndef ORACLE_BATCH_H
#define ORACLE_BATCH_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef int (*batch_codec_fn)(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* out);
double batch_run_generic(
    batch_codec_fn fn, int threads, int repeats, size_t n_chunks,
    const uint8_t* const* in_ptrs, const size_t* in_sizes,
    uint8_t* const* out_ptrs, const size_t* out_caps, size_t* out_sizes, int* errors);
#ifdef __cplusplus
}
#endif
#endif
