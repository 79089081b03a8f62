/*
 * nvcomp/amd_ext.h -- MI355X-build extensions. Nothing here exists in the reference's
 * interface and no reference-side caller needs it; the functions only move
 * performance trade-offs that the library otherwise decides by itself. They never
 * change a single output byte.
 */
#ifndef NVCOMP_AMD_EXT_H
#define NVCOMP_AMD_EXT_H

#include <stddef.h>

#include <hip/hip_runtime_api.h>

#include "nvcomp/shared_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* nvcompBatchedLZ4DecompressAsync can run as two kernels -- a token indexer with one LANE per chunk, then the
 * decoder proper fed from that index (held in the caller's temp buffer) -- when the batch has at least this many
 * chunks. Measured on MI355X (65 536 x 64 KiB, profiles/r02_decode_alternatives.json) the indexer's serial walk costs more
 * than the in-kernel token chase it replaces (322 vs 452 GB/s), so the path is OFF by default (threshold = SIZE_MAX)
 * and nvcompBatchedLZ4DecompressGetTempSize asks for the index space only for batches at or above the threshold:
 * set the threshold BEFORE the size query. Returns the previous value. Process-wide; not synchronised with
 * concurrent *Async calls. */
#define NVCOMP_AMD_LZ_INDEX_MIN_BATCH_DEFAULT ((size_t)-1)
size_t nvcompAmdSetLZIndexMinBatch(size_t min_batch);

/* nvcompBatched{LZ4,Snappy}DecompressAsync decode a batch of at most this many chunks with TWO waves per chunk -- one chases and
 * parses the tokens, the other executes the sequences, a queue in LDS between them -- because one wave per chunk cannot
 * fill the card below ~7 000 chunks and a chunk's latency is its wave's own dependent chain. Larger batches use one wave
 * per chunk (more chunks in flight per CU). 0 = never. Returns the previous value; process-wide. */
#define NVCOMP_AMD_LZ_PAIR_MAX_BATCH_DEFAULT 3072
size_t nvcompAmdSetLZPairMaxBatch(size_t max_batch);

/* Pack the chunks of a batch (e.g. what nvcompBatched<Fmt>CompressAsync left in its worst-case-sized slots) into one
 * contiguous buffer, in batch order and without gaps: device_offsets[i] = sum of device_chunk_bytes[0..i),
 * device_offsets[batch_size] = the packed size; chunk i is copied to device_packed + device_offsets[i]. Everything
 * happens on `stream`: a device-side prefix sum, then one wavefront per chunk. A chunk that would end behind
 * `packed_capacity` is left out (size the buffer as batch_size x the compressor's declared bound). This is the step
 * between "compress" and "send" of the reference's all-gather benchmark (benchmarks/benchmark_allgather.cpp:322-361
 * moves the whole slots instead); bench.py --allgather is built on it. */
nvcompStatus_t nvcompAmdBatchedPackAsync(
    const void* const* device_chunk_ptrs,
    const size_t* device_chunk_bytes,
    size_t batch_size,
    void* device_packed,
    size_t packed_capacity,
    size_t* device_offsets,
    hipStream_t stream);

#ifdef __cplusplus
}
#endif

#endif /* NVCOMP_AMD_EXT_H */
