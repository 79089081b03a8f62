/*
 * tests/emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
 *
 * A host-side stand-in for <hip/hip_runtime.h> that lets the library's kernel
 * sources be compiled with g++ and executed on the CPU, one workgroup at a time,
 * every lane as a ucontext coroutine. It exists so the kernels' logic can be
 * exercised (with ASan/UBSan and fuzzers) in the CPU-only container; it is never
 * linked into the product library and plays no part on a GPU box. The product
 * kernels are written for gfx950 only: this directory shadows exactly two
 * headers (<hip/hip_runtime.h> and "common/wave.h") and nothing in
 * nvcomp_amd/csrc is conditional on it.
 *
 * Semantics: lanes of a wave run one after another between cross-lane
 * operations (in a seeded pseudo-random order, to flush out order dependence);
 * every cross-lane operation (wave::ballot, wave::shuffle, __syncthreads, ...)
 * is a rendezvous of all live lanes of the wave / workgroup and the harness
 * aborts if lanes meet at different operations. Memory is sequentially
 * consistent, fences are no-ops.
 */
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __constant__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)

struct dim3
{
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

typedef int hipError_t;
typedef struct emuStream* hipStream_t;
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return hipSuccess; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
enum { hipHostMallocMapped = 2 };
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n ? n : 1); return hipSuccess; }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
#define hipStreamNonBlocking 1u
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
/* events: wall-clock stamps (the harness programs time their launches with them) */
struct emuEvent { double ms; };
typedef emuEvent* hipEvent_t;
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new emuEvent{0.0}; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  e->ms = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
  return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
#define hipEventDisableTiming 2u
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; } /* launches run at once */
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->ms - a->ms); return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
/* a "card" of one CU that keeps two workgroups resident: launches sized by residency (persistent waves,
 * common/lz_launch.hip.h) draw almost all their chunks from the ticket counter, which is what the tests want to see */
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 1 };
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 1; return hipSuccess; }
template <class K> inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, K, int, size_t) { *n = 2; return hipSuccess; }

namespace emu {

struct Idx3 { unsigned x, y, z; };

struct Lane
{
  Idx3 tid;
  Idx3 bid;
  int lane;     /* 0..63 */
  int wave;     /* wave index within the workgroup */
};

Lane* cur();
const dim3& block_dim();
const dim3& grid_dim();

/* rendezvous of all live lanes of the calling lane's wave. Every lane passes
 * two 64-bit operands; after the call `peer(i)` exposes lane i's operands and
 * live_mask() the set of lanes that took part. */
struct Contribution { uint64_t a, b; };
void wave_rendezvous(int op_id, uint64_t a, uint64_t b);
const Contribution& peer(int lane);
uint64_t live_mask();
void block_rendezvous();

void launch(const std::function<void()>& body, dim3 grid, dim3 block);
uint8_t* dynamic_lds(); /* 160 KiB, 16-byte aligned, shared by the lanes of the running workgroup */
void set_seed(uint64_t seed);

} // namespace emu

#define threadIdx (emu::cur()->tid)
#define blockIdx (emu::cur()->bid)
#define blockDim (emu::block_dim())
#define gridDim (emu::grid_dim())

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  emu::launch([&]() { kernel(__VA_ARGS__); }, dim3(grid), dim3(block))

inline void __syncthreads() { emu::block_rendezvous(); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
inline int __ffs(unsigned v) { return __builtin_ffs((int)v); }
inline int __clz(unsigned v) { return v ? __builtin_clz(v) : 32; }
inline int __clzll(unsigned long long v) { return v ? __builtin_clzll(v) : 64; }
inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xffffffu) * (b & 0xffffffu); }
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }

template <typename T> inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <typename T> inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T> inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <typename T> inline T atomicCAS(T* p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }
template <typename T> inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <typename T> inline T min(T a, T b) { return a < b ? a : b; }
template <typename T> inline T max(T a, T b) { return a > b ? a : b; }
