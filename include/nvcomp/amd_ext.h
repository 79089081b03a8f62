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

/* The order in which nvcompBatched{LZ4,Snappy}DecompressAsync hands the chunks of a batch to its persistent waves when
 * the batch is large enough for them (more chunks than waves stay resident) and the temp buffer has the size
 * nvcompBatched<Fmt>DecompressGetTempSize reports: the chunks that will take longest first. The cost of a chunk is
 * estimated from the first tokens of its stream. Exposed for inspection and tests: after the call (on `stream`)
 * `device_order[0 .. batch_size)` is a permutation of the chunk indices and `device_cost_class[i]` the class (0 ... 15,
 * a power-of-two scale of the estimated sequence count) of chunk i. `device_temp_ptr` as for the decompress call. */
nvcompStatus_t nvcompAmdBatchedLZ4DecompressOrderAsync(
    const void* const* device_compressed_ptrs,
    const size_t* device_compressed_bytes,
    size_t batch_size,
    void* device_temp_ptr,
    size_t temp_bytes,
    unsigned* device_order,
    unsigned char* device_cost_class,
    hipStream_t stream);
nvcompStatus_t nvcompAmdBatchedSnappyDecompressOrderAsync(
    const void* const* device_compressed_ptrs,
    const size_t* device_compressed_bytes,
    size_t batch_size,
    void* device_temp_ptr,
    size_t temp_bytes,
    unsigned* device_order,
    unsigned char* device_cost_class,
    hipStream_t stream);

#ifdef __cplusplus
}
#endif

#endif /* NVCOMP_AMD_EXT_H */
