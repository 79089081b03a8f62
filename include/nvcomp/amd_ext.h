/*
 * nvcomp/amd_ext.h -- MI355X-build extensions. Nothing here exists in the reference's
 * interface and no reference-side caller needs it. The library has no run-time
 * tuning state: which kernel a batch takes depends on the arguments of the call alone.
 */
#ifndef NVCOMP_AMD_EXT_H
#define NVCOMP_AMD_EXT_H

#include <stddef.h>

#include <hip/hip_runtime_api.h>

#include "nvcomp/shared_types.h"

#ifdef __cplusplus
extern "C" {
#endif

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

/* The token index that nvcompBatched{LZ4,Snappy}DecompressAsync builds per chunk when a batch runs on its persistent
 * one-wave-per-chunk kernels and the temp buffer has the size nvcompBatched<Fmt>DecompressGetTempSize reports
 * (csrc/common/lz_index.hip.h): the stream offsets of a PREFIX of the chunk's sequences, found by 64 joined serial walks,
 * and the offset where the decoder's classic token chase takes over. Exposed for inspection and tests: after the call (on
 * `stream`) chunk i has device_info[2 i] positions in device_lists[22016 i ...] (16 bits each, increasing; 22 016 =
 * 64 x 344 entries of room per chunk) and
 * device_info[2 i + 1] = the offset of the first sequence that is not in the list (0 entries, offset 0: a stream the
 * index is not made for -- shorter than 2 KiB, longer than 65 535 bytes). */
nvcompStatus_t nvcompAmdBatchedLZ4TokenIndexAsync(
    const void* const* device_compressed_ptrs,
    const size_t* device_compressed_bytes,
    size_t batch_size,
    unsigned short* device_lists,
    unsigned* device_info,
    hipStream_t stream);

#ifdef __cplusplus
}
#endif

#endif /* NVCOMP_AMD_EXT_H */
