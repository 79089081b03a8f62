/*
 * nvcomp/gzip.h -- batched gzip (RFC 1952) decompression, low-level C API, MI355X build.
 *
 * Replaces nvcompBatchedGzipDecompress* of the reference's closed libnvcomp.so (call sites:
 * examples/gzip_gpu_decompression.cu:110-164; the chunks there are written by zlib's deflateInit2(windowBits =
 * 15 | 16)). One gzip member per chunk: the 10-byte header with its optional FEXTRA / FNAME / FCOMMENT / FHCRC
 * fields is skipped, the DEFLATE stream is decoded by the decoder of nvcomp/deflate.h, and the trailer's ISIZE is
 * compared with the bytes produced (the CRC-32 is not recomputed). Decompression only, as in the reference.
 */
#ifndef NVCOMP_GZIP_H
#define NVCOMP_GZIP_H

#include "shared_types.h"
#include <hip/hip_runtime_api.h>

#ifdef __cplusplus
extern "C" {
#endif

static const size_t nvcompGzipRequiredAlignment = 1;

/* reference call site: examples/gzip_gpu_decompression.cu:110-111 */
nvcompStatus_t nvcompBatchedGzipDecompressGetTempSize(
    size_t num_chunks,
    size_t max_uncompressed_chunk_bytes,
    size_t* temp_bytes);

nvcompStatus_t nvcompBatchedGzipDecompressGetTempSizeEx(
    size_t num_chunks,
    size_t max_uncompressed_chunk_bytes,
    size_t* temp_bytes,
    size_t max_total_uncompressed_bytes);

/* reference call site: examples/gzip_gpu_decompression.cu:128-138 (and :151-161) */
nvcompStatus_t nvcompBatchedGzipDecompressAsync(
    const void* const* device_compressed_ptrs,
    const size_t* device_compressed_bytes,
    const size_t* device_uncompressed_bytes,
    size_t* device_actual_uncompressed_bytes,
    size_t batch_size,
    void* const device_temp_ptr,
    size_t temp_bytes,
    void* const* device_uncompressed_ptrs,
    nvcompStatus_t* device_statuses,
    hipStream_t stream);

/* The uncompressed size a member declares: its trailer's ISIZE (mod 2^32); 0 when the chunk is no gzip member. */
nvcompStatus_t nvcompBatchedGzipGetDecompressSizeAsync(
    const void* const* device_compressed_ptrs,
    const size_t* device_compressed_bytes,
    size_t* device_uncompressed_bytes,
    size_t batch_size,
    hipStream_t stream);

#ifdef __cplusplus
}
#endif

#endif /* NVCOMP_GZIP_H */
