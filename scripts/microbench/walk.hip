// Microbenchmark: how fast can a CU run the serial token walk? (design study, not product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ void walk1(uint32_t step, uint32_t limit, uint32_t& r, uint32_t& k, uint32_t& rec)
{
  uint32_t d;
  asm volatile(
      "s_mov_b32 m0, %[k]\n\t"
      "s_nop 0\n"
      "1:\n\t"
      "v_readlane_b32 %[d], %[step], %[r]\n\t"
      "v_writelane_b32 %[rec], %[r], m0\n\t"
      "s_add_u32 m0, m0, 1\n\t"
      "s_add_u32 %[r], %[r], %[d]\n\t"
      "s_cmp_lt_u32 %[r], %[limit]\n\t"
      "s_cbranch_scc1 1b\n\t"
      "s_mov_b32 %[k], m0"
      : [rec] "+v"(rec), [r] "+s"(r), [k] "+s"(k), [d] "=&s"(d)
      : [step] "v"(step), [limit] "s"(limit)
      : "scc");
}

// two tokens per branch: the second step is speculative (its record lands in lane k+1 and is
// simply overwritten/ignored if the first step already left the window)
__device__ __forceinline__ void walk2(uint32_t step, uint32_t limit, uint32_t& r, uint32_t& k, uint32_t& rec)
{
  uint32_t d, d2, r2;
  asm volatile(
      "s_mov_b32 m0, %[k]\n\t"
      "s_nop 0\n"
      "1:\n\t"
      "v_readlane_b32 %[d], %[step], %[r]\n\t"
      "v_writelane_b32 %[rec], %[r], m0\n\t"
      "s_add_u32 %[r2], %[r], %[d]\n\t"
      "s_and_b32 %[d2], %[r2], 63\n\t"
      "v_readlane_b32 %[d2], %[step], %[d2]\n\t"
      "s_add_u32 m0, m0, 1\n\t"
      "v_writelane_b32 %[rec], %[r2], m0\n\t"
      "s_cmp_lt_u32 %[r2], %[limit]\n\t"
      "s_cselect_b32 %[d2], %[d2], 0\n\t"
      "s_addc_u32 m0, m0, 0\n\t"
      "s_add_u32 %[r], %[r2], %[d2]\n\t"
      "s_cmp_lt_u32 %[r], %[limit]\n\t"
      "s_cbranch_scc1 1b\n\t"
      "s_mov_b32 %[k], m0"
      : [rec] "+v"(rec), [r] "+s"(r), [k] "+s"(k), [d] "=&s"(d), [d2] "=&s"(d2), [r2] "=&s"(r2)
      : [step] "v"(step), [limit] "s"(limit)
      : "scc");
}

template <int MODE>
__global__ void __launch_bounds__(256) walk_kernel(uint32_t* out, uint32_t iters, uint32_t delta)
{
  const uint32_t lane = threadIdx.x & 63;
  uint32_t step = delta + (lane & 1);  // mostly constant small steps
  uint32_t acc = 0;
  for (uint32_t it = 0; it < iters; ++it) {
    uint32_t r = __builtin_amdgcn_readfirstlane(it & 3), k = 0, rec = 0;
    if (MODE == 0) {
      walk1(step, 64, r, k, rec);
    } else if (MODE == 1) {
      walk2(step, 64, r, k, rec);
    } else {  // plain C loop (what the compiler makes of it)
      while (r < 64) {
        const uint32_t d = __builtin_amdgcn_readlane(step, r);
        rec = (lane == k) ? r : rec;
        ++k;
        r += d;
      }
    }
    acc += rec + k;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main()
{
  uint32_t* d;
  const int blocks = 256 * 8;  // 8 workgroups of 4 waves per CU = 8 waves/SIMD
  hipMalloc(&d, blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const uint32_t iters = 20000;
  for (int mode = 0; mode < 3; ++mode) {
    for (uint32_t delta : {3u, 8u}) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(walk_kernel<0>, dim3(blocks), dim3(256), 0, 0, d, iters, delta);
        if (mode == 1) hipLaunchKernelGGL(walk_kernel<1>, dim3(blocks), dim3(256), 0, 0, d, iters, delta);
        if (mode == 2) hipLaunchKernelGGL(walk_kernel<2>, dim3(blocks), dim3(256), 0, 0, d, iters, delta);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double tokens_per_walk = 64.0 / (delta + 0.5);
      const double tokens = (double)blocks * 4 * iters * tokens_per_walk;
      printf("mode %d delta %u: %.3f ms, %.1f G tokens/s, %.2f CU-cycles/token @2.1GHz\n", mode, delta, ms, tokens / ms / 1e6,
             ms * 1e-3 * 256 * 2.1e9 / tokens);
    }
  }
  return 0;
}
