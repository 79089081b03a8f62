/*
 * nvcomp/lz4.h -- batched LZ4 (block format) low-level C API, MI355X build.
 *
 * Every entry point below replaces the like-named symbol of the reference's
 * closed libnvcomp.so; the signature is reconstructed from the reference's
 * call sites (file:line cited per function). Wire format: the public LZ4
 * *block* format, so chunks interoperate with liblz4
 * (examples/lz4_cpu_compression.cu:61-66,137; examples/lz4_cpu_decompression.cu:142-157).
 *
 * All pointers named device_* must be dereferenceable by the GPU that owns
 * `stream`. The library allocates nothing; *Async calls only enqueue kernels
 * on `stream` and never synchronise the host.
 */
#ifndef NVCOMP_LZ4_H
#define NVCOMP_LZ4_H

#include "shared_types.h"
#include <hip/hip_runtime_api.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference: benchmarks/benchmark_lz4_chunked.cu:32,43 ; examples/low_level_quickstart_example.cpp:62 */
typedef struct
{
  /* Compress-side hint: matches are only emitted at multiples of the element
   * size. Legal: NVCOMP_TYPE_CHAR..NVCOMP_TYPE_UINT (0-5) and NVCOMP_TYPE_BITS. */
  nvcompType_t data_type;
} nvcompBatchedLZ4Opts_t;

static const nvcompBatchedLZ4Opts_t nvcompBatchedLZ4DefaultOpts = {NVCOMP_TYPE_CHAR};

/* Largest uncompressed chunk the compressor accepts (CHANGELOG.md:57). */
static const size_t nvcompLZ4CompressionMaxAllowedChunkSize = 1 << 24;

/* Alignment the API requires of chunk pointers (CHANGELOG.md:15-16): none.
 * Compressed and uncompressed chunks may start at any byte
 * (examples/BatchData.h:97-103 packs compressed chunks tight). */
static const size_t nvcompLZ4RequiredAlignment = 1;

/* reference call site: benchmarks/benchmark_template_chunked.cuh:420-421 */
nvcompStatus_t nvcompBatchedLZ4CompressGetTempSize(
    size_t batch_size,
    size_t max_uncompressed_chunk_bytes,
    nvcompBatchedLZ4Opts_t format_opts,
    size_t* temp_bytes);

/* reference: CHANGELOG.md:36-41 (never called in tree) */
nvcompStatus_t nvcompBatchedLZ4CompressGetTempSizeEx(
    size_t batch_size,
    size_t max_uncompressed_chunk_bytes,
    nvcompBatchedLZ4Opts_t format_opts,
    size_t* temp_bytes,
    const size_t max_total_uncompressed_bytes);

/* reference call site: benchmarks/benchmark_template_chunked.cuh:429-430 ;
 * examples/low_level_quickstart_example.cpp:68 */
nvcompStatus_t nvcompBatchedLZ4CompressGetMaxOutputChunkSize(
    size_t max_uncompressed_chunk_bytes,
    nvcompBatchedLZ4Opts_t format_opts,
    size_t* max_compressed_bytes);

/* reference call site: benchmarks/benchmark_template_chunked.cuh:441-451 ;
 * doc/lowlevel_c_quickstart.md:53-63 ; examples/lz4_cpu_decompression.cu:94-104
 *
 * A chunk LARGER than max_uncompressed_chunk_bytes would not fit the output slot sized from
 * ...CompressGetMaxOutputChunkSize(max_uncompressed_chunk_bytes): it is not compressed and its entry of
 * device_compressed_bytes reads 0 (no other chunk has a compressed size of 0 unless it was empty itself). The call
 * still returns nvcompSuccess -- it is asynchronous and has no per-chunk status array to write to -- so a caller
 * that cannot vouch for its chunk sizes checks for 0; the nvcomp::*Manager layer does and reports
 * nvcompErrorInvalidValue through the compression status.
 *
 * device_temp_ptr holds working state of the launch (the persistent waves' chunk counter): ONE TEMP BUFFER PER
 * IN-FLIGHT CALL. Two *Async calls that may overlap -- on different streams, or from different host threads -- must
 * be given different temp buffers (calls queued on one stream may share one). */
nvcompStatus_t nvcompBatchedLZ4CompressAsync(
    const void* const* device_uncompressed_ptrs,
    const size_t* device_uncompressed_bytes,
    size_t max_uncompressed_chunk_bytes,
    size_t batch_size,
    void* device_temp_ptr,
    size_t temp_bytes,
    void* const* device_compressed_ptrs,
    size_t* device_compressed_bytes,
    nvcompBatchedLZ4Opts_t format_opts,
    hipStream_t stream);

/* reference call site: benchmarks/benchmark_template_chunked.cuh:494-495 ;
 * examples/lz4_cpu_compression.cu:103-104 */
nvcompStatus_t nvcompBatchedLZ4DecompressGetTempSize(
    size_t num_chunks,
    size_t max_uncompressed_chunk_bytes,
    size_t* temp_bytes);

nvcompStatus_t nvcompBatchedLZ4DecompressGetTempSizeEx(
    size_t num_chunks,
    size_t max_uncompressed_chunk_bytes,
    size_t* temp_bytes,
    size_t max_total_uncompressed_bytes);

/* reference call site: benchmarks/benchmark_template_chunked.cuh:520-530 ;
 * examples/lz4_cpu_compression.cu:121-131 ; doc/lowlevel_c_quickstart.md:127-140.
 * device_actual_uncompressed_bytes and device_statuses may each be NULL;
 * with device_statuses == NULL nobody is told which chunks failed (a failed chunk still reads 0 in
 * device_actual_uncompressed_bytes); bounds are checked either way -- no stream, however corrupt, writes past its
 * output slot (the reference skips the checks with NULL statuses; here skipping them bought nothing measurable).
 *
 * device_temp_ptr holds working state of the launch (the persistent waves' chunk counter): ONE TEMP BUFFER PER
 * IN-FLIGHT CALL. Two *Async calls that may overlap -- on different streams, or from different host threads -- must
 * be given different temp buffers (calls queued on one stream may share one). */
nvcompStatus_t nvcompBatchedLZ4DecompressAsync(
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

/* reference call site: examples/low_level_quickstart_example.cpp:112-117 ;
 * doc/lowlevel_c_quickstart.md:104-109 */
nvcompStatus_t nvcompBatchedLZ4GetDecompressSizeAsync(
    const void* const* device_compressed_ptrs,
    const size_t* device_compressed_bytes,
    size_t* device_uncompressed_bytes,
    size_t batch_size,
    hipStream_t stream);

#ifdef __cplusplus
}
#endif

#endif /* NVCOMP_LZ4_H */
