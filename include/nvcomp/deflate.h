/*
 * nvcomp/deflate.h -- batched DEFLATE (RFC 1951, raw streams) low-level C API, MI355X build.
 *
 * Every entry point below replaces the like-named symbol of the reference's closed libnvcomp.so; the signatures are
 * reconstructed from the reference's call sites (file:line cited per function). Wire format: one raw DEFLATE stream
 * per chunk, no zlib or gzip wrapper -- what libdeflate_deflate_compress, zlib's deflateInit2(windowBits = -15) and
 * compress2 with its 2-byte header and 4-byte trailer cut off produce (examples/deflate_cpu_compression.cu:58-104).
 * The decoder accepts every legal stream: stored, fixed and dynamic blocks, any number of blocks per chunk,
 * distances up to 32 768.
 *
 * All pointers named device_* must be dereferenceable by the GPU that owns `stream`. The library allocates
 * nothing; *Async calls only enqueue kernels on `stream` and never synchronise the host.
 */
#ifndef NVCOMP_DEFLATE_H
#define NVCOMP_DEFLATE_H

#include "shared_types.h"
#include <hip/hip_runtime_api.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference: benchmarks/benchmark_deflate_chunked.cu:32,43-47 ("Deflate algorithm must be 0, 1, or 2"),
 * examples/deflate_cpu_decompression.cu:61 */
typedef struct
{
  int algo; /* compressor setting 0 .. 2 as in the reference's harness; every value produces standard streams. This
             * build: greedy LZ77 throughout; 0 = the fixed Huffman code (fastest), 1 and 2 = per-chunk Huffman codes
             * (two runs of the match finder and a code construction per chunk: better ratio, less than half the speed) */
} nvcompBatchedDeflateOpts_t;

static const nvcompBatchedDeflateOpts_t nvcompBatchedDeflateDefaultOpts = {0};

/* Largest uncompressed chunk the compressor accepts (benchmarks/benchmark_deflate_chunked.cu:53-63: "Deflate doesn't
 * support chunk sizes larger than 65536 bytes"). The decoder has no such limit. */
static const size_t nvcompDeflateCompressionMaxAllowedChunkSize = 1 << 16;

/* Alignment the API requires of chunk pointers (CHANGELOG.md:15-16): none. */
static const size_t nvcompDeflateRequiredAlignment = 1;

/* reference call site: examples/deflate_cpu_decompression.cu:64-68 */
nvcompStatus_t nvcompBatchedDeflateCompressGetTempSize(
    size_t batch_size,
    size_t max_uncompressed_chunk_bytes,
    nvcompBatchedDeflateOpts_t format_opts,
    size_t* temp_bytes);

nvcompStatus_t nvcompBatchedDeflateCompressGetTempSizeEx(
    size_t batch_size,
    size_t max_uncompressed_chunk_bytes,
    nvcompBatchedDeflateOpts_t format_opts,
    size_t* temp_bytes,
    const size_t max_total_uncompressed_bytes);

/* reference call site: examples/deflate_cpu_decompression.cu:77-78 */
nvcompStatus_t nvcompBatchedDeflateCompressGetMaxOutputChunkSize(
    size_t max_uncompressed_chunk_bytes,
    nvcompBatchedDeflateOpts_t format_opts,
    size_t* max_compressed_bytes);

/* reference call site: examples/deflate_cpu_decompression.cu:93-103 ; the output must be accepted by
 * libdeflate_deflate_decompress / zlib inflate (examples/deflate_cpu_decompression.cu:128-170)
 *
 * A chunk LARGER than max_uncompressed_chunk_bytes would not fit the output slot sized from
 * ...CompressGetMaxOutputChunkSize(max_uncompressed_chunk_bytes): it is not compressed and its entry of
 * device_compressed_bytes reads 0. The call still returns nvcompSuccess (it is asynchronous and has no per-chunk status
 * array to write to): a caller that cannot vouch for its chunk sizes checks for 0; the nvcomp::*Manager layer does and
 * reports nvcompErrorInvalidValue through the compression status. */
nvcompStatus_t nvcompBatchedDeflateCompressAsync(
    const void* const* device_uncompressed_ptrs,
    const size_t* device_uncompressed_bytes,
    size_t max_uncompressed_chunk_bytes,
    size_t batch_size,
    void* device_temp_ptr,
    size_t temp_bytes,
    void* const* device_compressed_ptrs,
    size_t* device_compressed_bytes,
    nvcompBatchedDeflateOpts_t format_opts,
    hipStream_t stream);

/* reference call site: examples/deflate_cpu_compression.cu:133-134 */
nvcompStatus_t nvcompBatchedDeflateDecompressGetTempSize(
    size_t num_chunks,
    size_t max_uncompressed_chunk_bytes,
    size_t* temp_bytes);

nvcompStatus_t nvcompBatchedDeflateDecompressGetTempSizeEx(
    size_t num_chunks,
    size_t max_uncompressed_chunk_bytes,
    size_t* temp_bytes,
    size_t max_total_uncompressed_bytes);

/* reference call site: examples/deflate_cpu_compression.cu:151-161 (and :174-184, the timed call).
 * device_actual_uncompressed_bytes and device_statuses may each be NULL; with device_statuses == NULL no per-chunk
 * bounds checking is performed. A chunk that fails reads size 0 and status nvcompErrorCannotDecompress. */
nvcompStatus_t nvcompBatchedDeflateDecompressAsync(
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

/* reference: doc/lowlevel_c_quickstart.md:104-109 (the generic signature). DEFLATE streams carry no length: the
 * symbols are decoded and counted, nothing is written; 0 for a malformed chunk. */
nvcompStatus_t nvcompBatchedDeflateGetDecompressSizeAsync(
    const void* const* device_compressed_ptrs,
    const size_t* device_compressed_bytes,
    size_t* device_uncompressed_bytes,
    size_t batch_size,
    hipStream_t stream);

#ifdef __cplusplus
}
#endif

#endif /* NVCOMP_DEFLATE_H */
