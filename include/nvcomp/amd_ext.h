/*
 * nvcomp/amd_ext.h -- MI355X-build extensions. Nothing here exists in the reference's
 * interface and no reference-side caller needs it; the functions only move
 * performance trade-offs that the library otherwise decides by itself. They never
 * change a single output byte.
 */
#ifndef NVCOMP_AMD_EXT_H
#define NVCOMP_AMD_EXT_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* nvcompBatched{LZ4,Snappy}DecompressAsync run as two kernels -- a token indexer with one
 * LANE per chunk, then the decoder proper fed from that index (held in the caller's temp
 * buffer) -- when the batch has at least this many chunks; smaller batches, which cannot
 * fill the indexer's lanes, use the single-kernel decoder that chases tokens itself.
 * Default: NVCOMP_AMD_LZ_INDEX_MIN_BATCH_DEFAULT. Returns the previous value. Process-wide;
 * not synchronised with concurrent *Async calls. */
#define NVCOMP_AMD_LZ_INDEX_MIN_BATCH_DEFAULT 8192
size_t nvcompAmdSetLZIndexMinBatch(size_t min_batch);

#ifdef __cplusplus
}
#endif

#endif /* NVCOMP_AMD_EXT_H */
