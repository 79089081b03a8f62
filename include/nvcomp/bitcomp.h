/*
 * nvcomp/bitcomp.h -- batched Bitcomp (delta + per-row bit-packing of numerical
 * data) low-level C API, MI355X build.
 *
 * Entry points replace the like-named symbols of the reference's closed
 * libnvcomp.so (call sites: benchmarks/benchmark_bitcomp_chunked.cu:32-127;
 * HLIF use: benchmarks/benchmark_hlif.cpp:193). The reference never documents
 * the Bitcomp bitstream and states that its decompressor only accepts its own
 * compressor's output (README.md:13), so the stream written here is this
 * library's own (DESIGN.md "Bitcomp stream layout") and parity is round-trip only.
 */
#ifndef NVCOMP_BITCOMP_H
#define NVCOMP_BITCOMP_H

#include "shared_types.h"
#include <hip/hip_runtime_api.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Aggregate order pinned by `{0, NVCOMP_TYPE_UCHAR}` and the field names used at
 * benchmarks/benchmark_bitcomp_chunked.cu:35-36,47,58 and benchmarks/benchmark_hlif.cpp:193. */
typedef struct
{
  int algorithm_type;     /* 0: delta between neighbours + bit-packing (default); 1: bit-packing only ("sparse") */
  nvcompType_t data_type; /* element type, 0..7; chunk sizes must be multiples of its size */
} nvcompBatchedBitcompFormatOpts;

static const nvcompBatchedBitcompFormatOpts nvcompBatchedBitcompDefaultOpts = {0, NVCOMP_TYPE_UCHAR};

static const size_t nvcompBitcompCompressionMaxAllowedChunkSize = 1 << 24;
/* Any alignment is accepted; 4-byte aligned compressed chunks and element-aligned
 * uncompressed chunks are the fast case. */
static const size_t nvcompBitcompRequiredAlignment = 1;

nvcompStatus_t nvcompBatchedBitcompCompressGetTempSize(
    size_t batch_size,
    size_t max_uncompressed_chunk_bytes,
    nvcompBatchedBitcompFormatOpts format_opts,
    size_t* temp_bytes);

/* reference: CHANGELOG.md:36-41 (nvcompBatched*CompressGetTempSizeEx; never called in tree) */
nvcompStatus_t nvcompBatchedBitcompCompressGetTempSizeEx(
    size_t batch_size,
    size_t max_uncompressed_chunk_bytes,
    nvcompBatchedBitcompFormatOpts format_opts,
    size_t* temp_bytes,
    const size_t max_total_uncompressed_bytes);

nvcompStatus_t nvcompBatchedBitcompCompressGetMaxOutputChunkSize(
    size_t max_uncompressed_chunk_bytes,
    nvcompBatchedBitcompFormatOpts format_opts,
    size_t* max_compressed_bytes);

/* A chunk LARGER than max_uncompressed_chunk_bytes would not fit the output slot sized from
 * ...CompressGetMaxOutputChunkSize(max_uncompressed_chunk_bytes): it is not compressed and its entry of
 * device_compressed_bytes reads 0. The call still returns nvcompSuccess (it is asynchronous and has no per-chunk status
 * array to write to): a caller that cannot vouch for its chunk sizes checks for 0; the nvcomp::*Manager layer does and
 * reports nvcompErrorInvalidValue through the compression status. */
nvcompStatus_t nvcompBatchedBitcompCompressAsync(
    const void* const* device_uncompressed_ptrs,
    const size_t* device_uncompressed_bytes,
    size_t max_uncompressed_chunk_bytes,
    size_t batch_size,
    void* device_temp_ptr,
    size_t temp_bytes,
    void* const* device_compressed_ptrs,
    size_t* device_compressed_bytes,
    nvcompBatchedBitcompFormatOpts format_opts,
    hipStream_t stream);

nvcompStatus_t nvcompBatchedBitcompDecompressGetTempSize(
    size_t num_chunks,
    size_t max_uncompressed_chunk_bytes,
    size_t* temp_bytes);

/* reference: CHANGELOG.md:114-117 (nvcompBatched<Format>DecompressGetTempSizeEx) */
nvcompStatus_t nvcompBatchedBitcompDecompressGetTempSizeEx(
    size_t num_chunks,
    size_t max_uncompressed_chunk_bytes,
    size_t* temp_bytes,
    size_t max_total_uncompressed_bytes);

/* The reference requires non-NULL actual-size and status arrays for Bitcomp and is "not
 * fully asynchronous" (README.md:14-15); this build accepts NULL for either and never
 * synchronises (superset). */
nvcompStatus_t nvcompBatchedBitcompDecompressAsync(
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

nvcompStatus_t nvcompBatchedBitcompGetDecompressSizeAsync(
    const void* const* device_compressed_ptrs,
    const size_t* device_compressed_bytes,
    size_t* device_uncompressed_bytes,
    size_t batch_size,
    hipStream_t stream);

#ifdef __cplusplus
}
#endif

#endif /* NVCOMP_BITCOMP_H */
