/*
 * nvcomp/cascaded.h -- batched Cascaded (RLE + delta + bit-packing) low-level
 * C API, MI355X build.
 *
 * Entry points replace the like-named symbols of the reference's closed
 * libnvcomp.so (call sites: benchmarks/benchmark_cascaded_chunked.cu:137-151).
 * The scheme follows doc/cascaded_overview.md:6-44; the reference never
 * documents its bitstream, so the container written here is this library's own
 * (DESIGN.md "Cascaded stream layout") and parity is round-trip only.
 */
#ifndef NVCOMP_CASCADED_H
#define NVCOMP_CASCADED_H

#include "shared_types.h"
#include <hip/hip_runtime_api.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Aggregate order pinned by `{4096, NVCOMP_TYPE_UINT, 2, 1, 1}` and the field
 * names used at benchmarks/benchmark_cascaded_chunked.cu:35-36,47,58,69,80 and
 * benchmarks/benchmark_hlif.cpp:201-203. */
typedef struct
{
  size_t chunk_size;  /* internal sub-chunk size in bytes (512..16384, multiple of the type size) */
  nvcompType_t type;  /* element type, 0..7 */
  int num_RLEs;       /* number of run-length layers, 0..7 */
  int num_deltas;     /* number of delta layers, 0..7 */
  int use_bp;         /* 0/1: bit-pack every resulting stream */
} nvcompBatchedCascadedOpts_t;

static const nvcompBatchedCascadedOpts_t nvcompBatchedCascadedDefaultOpts = {4096, NVCOMP_TYPE_INT, 2, 1, 1};

static const size_t nvcompCascadedCompressionMaxAllowedChunkSize = 1 << 24;
/* Uncompressed and compressed chunk pointers must be aligned to the element
 * type size / 4 bytes respectively (nvcompErrorAlignment is written per chunk otherwise). */
static const size_t nvcompCascadedRequiredAlignment = 4;

nvcompStatus_t nvcompBatchedCascadedCompressGetTempSize(
    size_t batch_size,
    size_t max_uncompressed_chunk_bytes,
    nvcompBatchedCascadedOpts_t format_opts,
    size_t* temp_bytes);

/* reference: CHANGELOG.md:36-41 (nvcompBatched*CompressGetTempSizeEx; never called in tree) */
nvcompStatus_t nvcompBatchedCascadedCompressGetTempSizeEx(
    size_t batch_size,
    size_t max_uncompressed_chunk_bytes,
    nvcompBatchedCascadedOpts_t format_opts,
    size_t* temp_bytes,
    const size_t max_total_uncompressed_bytes);

nvcompStatus_t nvcompBatchedCascadedCompressGetMaxOutputChunkSize(
    size_t max_uncompressed_chunk_bytes,
    nvcompBatchedCascadedOpts_t format_opts,
    size_t* max_compressed_bytes);

nvcompStatus_t nvcompBatchedCascadedCompressAsync(
    const void* const* device_uncompressed_ptrs,
    const size_t* device_uncompressed_bytes,
    size_t max_uncompressed_chunk_bytes,
    size_t batch_size,
    void* device_temp_ptr,
    size_t temp_bytes,
    void* const* device_compressed_ptrs,
    size_t* device_compressed_bytes,
    nvcompBatchedCascadedOpts_t format_opts,
    hipStream_t stream);

nvcompStatus_t nvcompBatchedCascadedDecompressGetTempSize(
    size_t num_chunks,
    size_t max_uncompressed_chunk_bytes,
    size_t* temp_bytes);

/* reference: CHANGELOG.md:114-117 (nvcompBatched<Format>DecompressGetTempSizeEx) */
nvcompStatus_t nvcompBatchedCascadedDecompressGetTempSizeEx(
    size_t num_chunks,
    size_t max_uncompressed_chunk_bytes,
    size_t* temp_bytes,
    size_t max_total_uncompressed_bytes);

/* The reference requires non-NULL actual-size and status arrays for Cascaded
 * (README.md:14); this build accepts NULL for either (superset). */
nvcompStatus_t nvcompBatchedCascadedDecompressAsync(
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

nvcompStatus_t nvcompBatchedCascadedGetDecompressSizeAsync(
    const void* const* device_compressed_ptrs,
    const size_t* device_compressed_bytes,
    size_t* device_uncompressed_bytes,
    size_t batch_size,
    hipStream_t stream);

#ifdef __cplusplus
}
#endif

#endif /* NVCOMP_CASCADED_H */
