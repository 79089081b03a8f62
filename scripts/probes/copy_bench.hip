// How fast can one wave per chunk copy 64 KiB literal runs (an incompressible LZ4 / Snappy chunk is one such run)?
// 16 384 chunks, source 259 bytes into a 65 809-byte slot (any alignment), destination 64 KiB slots; persistent waves,
// 7 workgroups of 4 waves per CU, a ticket counter -- the launch shape of the decoders (common/lz_launch.hip.h).
//   plain:      four 1 KiB loads in flight, then four aligned stores (lzw::stream_copy as of round 3)
//   nt stores / nt both:  the same with non-temporal stores (and loads)
//   pipe:       the next four loads are issued before the stores of the current four
//   8 in flight: eight loads, then eight stores
// build: hipcc -O3 --offload-arch=gfx950 -Invcomp_amd/csrc scripts/probes/copy_bench.hip -o scripts/probes/copy_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#include "common/wave.h"

using wave::u32x4;

template <int NT>
__device__ __forceinline__ u32x4 ld(const uint8_t* p)
{
  if (NT >= 2) {
    const uint32_t* q = (const uint32_t*)p; // dword-aligned in this probe's nt-load variant only if p is; use 4 dword loads
    u32x4 r;
    r.x = __builtin_nontemporal_load(q), r.y = __builtin_nontemporal_load(q + 1);
    r.z = __builtin_nontemporal_load(q + 2), r.w = __builtin_nontemporal_load(q + 3);
    return r;
  }
  return wave::gload_u32x4(p);
}
template <int NT>
__device__ __forceinline__ void st(uint8_t* p, u32x4 v)
{
  if (NT >= 1) {
    __builtin_nontemporal_store(v, (u32x4*)p);
  } else {
    wave::gstore_u32x4_aligned(p, v);
  }
}

template <int NT, int MODE>
__device__ __forceinline__ void copy_chunk(uint8_t* dst, const uint8_t* src, uint32_t n)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  if (MODE == 0) {
    for (uint32_t base = 0; base + 4096 <= n; base += 4096) {
      const uint32_t at = base + 16 * lane;
      const u32x4 a = ld<NT>(src + at), b = ld<NT>(src + at + 1024), c = ld<NT>(src + at + 2048), d = ld<NT>(src + at + 3072);
      st<NT>(dst + at, a), st<NT>(dst + at + 1024, b), st<NT>(dst + at + 2048, c), st<NT>(dst + at + 3072, d);
    }
  } else if (MODE == 1) {
    uint32_t at = 16 * lane;
    u32x4 a = ld<NT>(src + at), b = ld<NT>(src + at + 1024), c = ld<NT>(src + at + 2048), d = ld<NT>(src + at + 3072);
    for (uint32_t base = 4096; base + 4096 <= n; base += 4096) {
      const uint32_t nx = base + 16 * lane;
      const u32x4 a2 = ld<NT>(src + nx), b2 = ld<NT>(src + nx + 1024), c2 = ld<NT>(src + nx + 2048), d2 = ld<NT>(src + nx + 3072);
      st<NT>(dst + at, a), st<NT>(dst + at + 1024, b), st<NT>(dst + at + 2048, c), st<NT>(dst + at + 3072, d);
      a = a2, b = b2, c = c2, d = d2;
      at = nx;
    }
    st<NT>(dst + at, a), st<NT>(dst + at + 1024, b), st<NT>(dst + at + 2048, c), st<NT>(dst + at + 3072, d);
  } else {
    for (uint32_t base = 0; base + 8192 <= n; base += 8192) {
      const uint32_t at = base + 16 * lane;
      u32x4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = ld<NT>(src + at + 1024 * i);
#pragma unroll
      for (int i = 0; i < 8; ++i) st<NT>(dst + at + 1024 * i, v[i]);
    }
  }
}

template <int NT, int MODE>
__global__ void __launch_bounds__(256) copy_kernel(uint8_t* dst, const uint8_t* src, uint32_t n_chunks, uint32_t src_stride, uint32_t src_off,
                                                   uint32_t* ticket, uint32_t first_dynamic)
{
  uint32_t i = blockIdx.x * 4 + wave::uniform(threadIdx.x >> 6);
  while (i < n_chunks) {
    copy_chunk<NT, MODE>(dst + (size_t)i * 65536, src + (size_t)i * src_stride + src_off, 65536);
    uint32_t nx = 0;
    if (wave::lane_id() == 0) {
      nx = atomicAdd(ticket, 1u);
    }
    i = first_dynamic + wave::uniform(__shfl(nx, 0));
  }
}

template <int NT, int MODE>
static void run(const char* name, uint8_t* dst, const uint8_t* src, uint32_t n, uint32_t stride, uint32_t off, uint32_t* ticket, unsigned wgs)
{
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  const int reps = 40;
  for (int r = 0; r < 10 + reps; ++r) {
    if (r == 10) (void)hipEventRecord(e0);
    (void)hipMemsetAsync(ticket, 0, 4);
    hipLaunchKernelGGL((copy_kernel<NT, MODE>), dim3(wgs), dim3(256), 0, 0, dst, src, n, stride, off, ticket, wgs * 4);
  }
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  std::vector<uint8_t> a(65536), b(65536);
  (void)hipMemcpy(a.data(), dst + (size_t)(n - 1) * 65536, 65536, hipMemcpyDeviceToHost);
  (void)hipMemcpy(b.data(), src + (size_t)(n - 1) * stride + off, 65536, hipMemcpyDeviceToHost);
  printf("{\"variant\": \"%s\", \"src_offset\": %u, \"workgroups\": %u, \"ms\": %.4f, \"copy_GBps\": %.1f, \"check\": \"%s\"}\n", name, off, wgs, ms,
         (double)n * 65536 / ms / 1e6, a == b ? "ok" : "MISMATCH");
  fflush(stdout);
  (void)hipMemset(dst, 0, (size_t)n * 65536);
}

int main()
{
  const uint32_t n = 16384, stride = 65809 + 15 & ~15u;
  uint8_t *src, *dst;
  uint32_t* ticket;
  (void)hipMalloc(&src, (size_t)n * stride + 4096);
  (void)hipMalloc(&dst, (size_t)n * 65536);
  (void)hipMalloc(&ticket, 64);
  std::vector<uint8_t> h((size_t)n * stride + 4096);
  uint64_t x = 88172645463325252ull;
  for (size_t i = 0; i + 8 <= h.size(); i += 8) {
    x ^= x << 13, x ^= x >> 7, x ^= x << 17;
    *(uint64_t*)&h[i] = x;
  }
  (void)hipMemcpy(src, h.data(), h.size(), hipMemcpyHostToDevice);
  int cus = 0;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  for (unsigned per_cu : {7u, 4u, 8u}) {
    const unsigned wgs = per_cu * cus;
    for (uint32_t off : {259u, 256u}) {
      run<0, 0>("plain", dst, src, n, stride, off, ticket, wgs);
      run<1, 0>("nt stores", dst, src, n, stride, off, ticket, wgs);
      if (off % 4 == 0) run<2, 0>("nt loads (4 dwords) and stores", dst, src, n, stride, off, ticket, wgs);
      run<0, 1>("pipe", dst, src, n, stride, off, ticket, wgs);
      run<1, 1>("pipe, nt stores", dst, src, n, stride, off, ticket, wgs);
      run<0, 2>("8 in flight", dst, src, n, stride, off, ticket, wgs);
      run<1, 2>("8 in flight, nt stores", dst, src, n, stride, off, ticket, wgs);
    }
  }
  (void)hipMemcpy(dst, src, (size_t)n * 65536, hipMemcpyDeviceToDevice);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 20; ++r) (void)hipMemcpyAsync(dst, src, (size_t)n * 65536, hipMemcpyDeviceToDevice, 0);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("{\"variant\": \"hipMemcpyAsync device to device, 1 GiB\", \"copy_GBps\": %.1f}\n", (double)n * 65536 / (ms / 20) / 1e6);
  return 0;
}
