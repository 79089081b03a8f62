/*
 * nvcomp/snappy.h -- batched Snappy (raw format) low-level C API, MI355X build.
 *
 * Every entry point below replaces the like-named symbol of the reference's
 * closed libnvcomp.so; the signature is reconstructed from the reference's
 * call sites (file:line cited per function). Wire format: the public Snappy
 * *raw* format (varint32 uncompressed length, then tagged elements), so chunks
 * interoperate with snappy::RawCompress / RawUncompress. The decoder accepts
 * every legal stream, including element kinds its own compressor never emits
 * (CHANGELOG.md:182-184).
 *
 * All pointers named device_* must be dereferenceable by the GPU that owns
 * `stream`. The library allocates nothing; *Async calls only enqueue kernels
 * on `stream` and never synchronise the host.
 */
#ifndef NVCOMP_SNAPPY_H
#define NVCOMP_SNAPPY_H

#include "shared_types.h"
#include <hip/hip_runtime_api.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference: benchmarks/benchmark_hlif.cpp:191 (value-initialised with {}) ;
 * benchmarks/benchmark_snappy_chunked.cu:37,57 */
typedef struct
{
  int reserved; /* no options; must be 0 */
} nvcompBatchedSnappyOpts_t;

static const nvcompBatchedSnappyOpts_t nvcompBatchedSnappyDefaultOpts = {0};

/* Largest uncompressed chunk the compressor accepts (CHANGELOG.md:57). */
static const size_t nvcompSnappyCompressionMaxAllowedChunkSize = 1 << 24;

/* Alignment the API requires of chunk pointers (CHANGELOG.md:15-16): none.
 * Compressed and uncompressed chunks may start at any byte
 * (benchmark_template_chunked.cuh:181-183 aligns inputs to 8 B, outputs are exact-size). */
static const size_t nvcompSnappyRequiredAlignment = 1;

/* reference call site: benchmarks/benchmark_template_chunked.cuh:420-421 ; benchmarks/benchmark_snappy_synth.cpp:128-133 */
nvcompStatus_t nvcompBatchedSnappyCompressGetTempSize(
    size_t batch_size,
    size_t max_uncompressed_chunk_bytes,
    nvcompBatchedSnappyOpts_t format_opts,
    size_t* temp_bytes);

/* reference: CHANGELOG.md:36-41 (never called in tree) */
nvcompStatus_t nvcompBatchedSnappyCompressGetTempSizeEx(
    size_t batch_size,
    size_t max_uncompressed_chunk_bytes,
    nvcompBatchedSnappyOpts_t format_opts,
    size_t* temp_bytes,
    const size_t max_total_uncompressed_bytes);

/* reference call site: benchmarks/benchmark_template_chunked.cuh:429-430 ;
 * benchmarks/benchmark_snappy_synth.cpp:138-143 */
nvcompStatus_t nvcompBatchedSnappyCompressGetMaxOutputChunkSize(
    size_t max_uncompressed_chunk_bytes,
    nvcompBatchedSnappyOpts_t format_opts,
    size_t* max_compressed_bytes);

/* reference call site: benchmarks/benchmark_template_chunked.cuh:441-451 ;
 * doc/lowlevel_c_quickstart.md:53-63 ; benchmarks/benchmark_snappy_synth.cpp:163-173
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
nvcompStatus_t nvcompBatchedSnappyCompressAsync(
    const void* const* device_uncompressed_ptrs,
    const size_t* device_uncompressed_bytes,
    size_t max_uncompressed_chunk_bytes,
    size_t batch_size,
    void* device_temp_ptr,
    size_t temp_bytes,
    void* const* device_compressed_ptrs,
    size_t* device_compressed_bytes,
    nvcompBatchedSnappyOpts_t format_opts,
    hipStream_t stream);

/* reference call site: benchmarks/benchmark_template_chunked.cuh:494-495 ;
 * benchmarks/benchmark_snappy_synth.cpp:220-223 */
nvcompStatus_t nvcompBatchedSnappyDecompressGetTempSize(
    size_t num_chunks,
    size_t max_uncompressed_chunk_bytes,
    size_t* temp_bytes);

nvcompStatus_t nvcompBatchedSnappyDecompressGetTempSizeEx(
    size_t num_chunks,
    size_t max_uncompressed_chunk_bytes,
    size_t* temp_bytes,
    size_t max_total_uncompressed_bytes);

/* reference call site: benchmarks/benchmark_template_chunked.cuh:520-530 ;
 * benchmarks/benchmark_snappy_synth.cpp:241-251 (passes the SAME device array as
 * device_uncompressed_bytes and device_actual_uncompressed_bytes: aliasing is legal) ;
 * doc/lowlevel_c_quickstart.md:127-140.
 * device_actual_uncompressed_bytes and device_statuses may each be NULL;
 * with device_statuses == NULL nobody is told which chunks failed (a failed chunk still reads 0 in
 * device_actual_uncompressed_bytes); bounds are checked either way -- no stream, however corrupt, writes past its
 * output slot (the reference skips the checks with NULL statuses; here skipping them bought nothing measurable).
 *
 * device_temp_ptr holds working state of the launch (the persistent waves' chunk counter): ONE TEMP BUFFER PER
 * IN-FLIGHT CALL. Two *Async calls that may overlap -- on different streams, or from different host threads -- must
 * be given different temp buffers (calls queued on one stream may share one). */
nvcompStatus_t nvcompBatchedSnappyDecompressAsync(
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

/* reference call site: doc/lowlevel_c_quickstart.md:104-109 (reads the varint preamble only) */
nvcompStatus_t nvcompBatchedSnappyGetDecompressSizeAsync(
    const void* const* device_compressed_ptrs,
    const size_t* device_compressed_bytes,
    size_t* device_uncompressed_bytes,
    size_t batch_size,
    hipStream_t stream);

#ifdef __cplusplus
}
#endif

#endif /* NVCOMP_SNAPPY_H */
