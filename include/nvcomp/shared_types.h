/*
 * nvcomp/shared_types.h -- status and element-type enums of the batched
 * low-level C API (LLIF), MI355X-native build.
 *
 * Drop-in boundary: these are the types every reference call site uses
 *   nvcompStatus_t   reference: benchmarks/benchmark_template_chunked.cuh:422,506-508,554
 *                               examples/high_level_quickstart_example.cpp:314 (nvcompErrorBadChecksum)
 *                               CHANGELOG.md:16 (nvcompErrorAlignment)
 *   nvcompType_t     reference: benchmarks/benchmark_template_chunked.cuh:88-123,
 *                               benchmarks/benchmark_cascaded_chunked.cu:109-112 ("must be 0-7"),
 *                               benchmarks/benchmark_lz4_chunked.cu:69-72 ("0-5 or 255")
 *
 * The stream type of every *Async entry point is hipStream_t: callers are
 * HIP programs on ROCm (the reference's callers are CUDA-runtime programs and
 * are recompiled against these headers).
 */
#ifndef NVCOMP_SHARED_TYPES_H
#define NVCOMP_SHARED_TYPES_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Return / per-chunk status. sizeof == 4 (device status arrays are
 * batch_size * sizeof(nvcompStatus_t), benchmark_template_chunked.cuh:506-508). */
typedef enum nvcompStatus_t
{
  nvcompSuccess = 0,
  nvcompErrorInvalidValue = 10,
  nvcompErrorNotSupported = 11,
  nvcompErrorCannotDecompress = 12,
  nvcompErrorBadChecksum = 13,
  nvcompErrorCannotVerifyChecksums = 14,
  nvcompErrorOutputBufferTooSmall = 15,
  nvcompErrorWrongHeaderLength = 16,
  nvcompErrorAlignment = 17,
  nvcompErrorChunkSizeTooLarge = 18,
  nvcompErrorCudaError = 1000, /* name kept for source compatibility: a HIP runtime error */
  nvcompErrorInternal = 10000
} nvcompStatus_t;

/* Element-type hint for the typed codecs. */
typedef enum nvcompType_t
{
  NVCOMP_TYPE_CHAR = 0,
  NVCOMP_TYPE_UCHAR = 1,
  NVCOMP_TYPE_SHORT = 2,
  NVCOMP_TYPE_USHORT = 3,
  NVCOMP_TYPE_INT = 4,
  NVCOMP_TYPE_UINT = 5,
  NVCOMP_TYPE_LONGLONG = 6,
  NVCOMP_TYPE_ULONGLONG = 7,
  NVCOMP_TYPE_BITS = 0xff
} nvcompType_t;

#ifdef __cplusplus
}
#endif

#endif /* NVCOMP_SHARED_TYPES_H */
