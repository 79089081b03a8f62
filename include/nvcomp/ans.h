/*
 * nvcomp/ans.h -- batched ANS (range-variant asymmetric numeral systems entropy
 * coder over bytes) low-level C API, MI355X build.
 *
 * Entry points replace the like-named symbols of the reference's closed
 * libnvcomp.so (call sites: benchmarks/benchmark_ans_chunked.cu:29-81; HLIF use:
 * benchmarks/benchmark_hlif.cpp:195). The reference's ANS bitstream is not
 * documented and not part of its tree, so the stream written here is this
 * library's own (DESIGN.md "ANS stream layout") and parity is round-trip only.
 */
#ifndef NVCOMP_ANS_H
#define NVCOMP_ANS_H

#include "shared_types.h"
#include <hip/hip_runtime_api.h>

#ifdef __cplusplus
extern "C" {
#endif

/* "the enum has only one value at the moment" (benchmarks/benchmark_ans_chunked.cu:38-40);
 * the harness value-initialises the struct (`nvcompBatchedANSOpts_t{}`, :31; benchmark_hlif.cpp:195). */
typedef enum
{
  nvcomp_rANS = 0
} nvcompANSType_t;

typedef struct
{
  nvcompANSType_t type;
} nvcompBatchedANSOpts_t;

static const nvcompBatchedANSOpts_t nvcompBatchedANSDefaultOpts = {nvcomp_rANS};

/* benchmarks/benchmark_ans_chunked.cu:46-52 rejects chunks of 2^32 bytes and more; this build's limit: */
static const size_t nvcompANSCompressionMaxAllowedChunkSize = 1 << 24;
/* Any alignment is accepted; 4-byte aligned chunks are the fast case. */
static const size_t nvcompANSRequiredAlignment = 1;

nvcompStatus_t nvcompBatchedANSCompressGetTempSize(
    size_t batch_size,
    size_t max_uncompressed_chunk_bytes,
    nvcompBatchedANSOpts_t format_opts,
    size_t* temp_bytes);

/* reference: CHANGELOG.md:36-41 (nvcompBatched*CompressGetTempSizeEx; never called in tree) */
nvcompStatus_t nvcompBatchedANSCompressGetTempSizeEx(
    size_t batch_size,
    size_t max_uncompressed_chunk_bytes,
    nvcompBatchedANSOpts_t format_opts,
    size_t* temp_bytes,
    const size_t max_total_uncompressed_bytes);

nvcompStatus_t nvcompBatchedANSCompressGetMaxOutputChunkSize(
    size_t max_uncompressed_chunk_bytes,
    nvcompBatchedANSOpts_t format_opts,
    size_t* max_compressed_bytes);

/* A chunk LARGER than max_uncompressed_chunk_bytes would not fit the output slot sized from
 * ...CompressGetMaxOutputChunkSize(max_uncompressed_chunk_bytes): it is not compressed and its entry of
 * device_compressed_bytes reads 0. The call still returns nvcompSuccess (it is asynchronous and has no per-chunk status
 * array to write to): a caller that cannot vouch for its chunk sizes checks for 0; the nvcomp::*Manager layer does and
 * reports nvcompErrorInvalidValue through the compression status. */
nvcompStatus_t nvcompBatchedANSCompressAsync(
    const void* const* device_uncompressed_ptrs,
    const size_t* device_uncompressed_bytes,
    size_t max_uncompressed_chunk_bytes,
    size_t batch_size,
    void* device_temp_ptr,
    size_t temp_bytes,
    void* const* device_compressed_ptrs,
    size_t* device_compressed_bytes,
    nvcompBatchedANSOpts_t format_opts,
    hipStream_t stream);

nvcompStatus_t nvcompBatchedANSDecompressGetTempSize(
    size_t num_chunks,
    size_t max_uncompressed_chunk_bytes,
    size_t* temp_bytes);

/* reference: CHANGELOG.md:114-117 (nvcompBatched<Format>DecompressGetTempSizeEx) */
nvcompStatus_t nvcompBatchedANSDecompressGetTempSizeEx(
    size_t num_chunks,
    size_t max_uncompressed_chunk_bytes,
    size_t* temp_bytes,
    size_t max_total_uncompressed_bytes);

/* device_actual_uncompressed_bytes and device_statuses may each be NULL. */
nvcompStatus_t nvcompBatchedANSDecompressAsync(
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

nvcompStatus_t nvcompBatchedANSGetDecompressSizeAsync(
    const void* const* device_compressed_ptrs,
    const size_t* device_compressed_bytes,
    size_t* device_uncompressed_bytes,
    size_t batch_size,
    hipStream_t stream);

#ifdef __cplusplus
}
#endif

#endif /* NVCOMP_ANS_H */
