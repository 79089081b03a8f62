// Microbenchmark: cost of misaligned / strided LDS dword accesses on gfx950 (design study, not product).
// Each wave issues N ds_write_b32 / ds_read_b32 at byte address lane*stride + off; reports cycles per instruction
// per CU with 16 waves resident.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int WRITE, int WIDTH>
__global__ void __launch_bounds__(256) lds_kernel(uint32_t* out, uint32_t iters, uint32_t stride, uint32_t off)
{
  __shared__ __attribute__((aligned(16))) uint8_t lds[4][8192 + 64];
  const uint32_t lane = threadIdx.x & 63;
  uint8_t* base = lds[threadIdx.x >> 6];
  const uint32_t a = (lane * stride + off) & 8191u;
  uint32_t acc = lane;
  for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      uint8_t* p = base + ((a + 64u * u) & 8191u);
      if (WRITE) {
        if (WIDTH == 4) {
          asm volatile("ds_write_b32 %0, %1" ::"v"((uint32_t)(uintptr_t)p), "v"(acc) : "memory");
        } else {
          asm volatile("ds_write_b8 %0, %1" ::"v"((uint32_t)(uintptr_t)p), "v"(acc) : "memory");
        }
      } else {
        uint32_t v;
        if (WIDTH == 4) {
          asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(uintptr_t)p) : "memory");
        } else {
          asm volatile("ds_read_u8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(uintptr_t)p) : "memory");
        }
        acc += v;
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (acc == 0x12345678u) {
    out[threadIdx.x] = acc;
  }
}

template <int WRITE, int WIDTH>
static double run(uint32_t stride, uint32_t off)
{
  uint32_t* d;
  hipMalloc(&d, 4096);
  const uint32_t iters = 2000;
  const int blocks = 256 * 4; // 4 workgroups (16 waves) per CU
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((lds_kernel<WRITE, WIDTH>), dim3(blocks), dim3(256), 0, 0, d, 10u, stride, off);
  hipEventRecord(e0);
  hipLaunchKernelGGL((lds_kernel<WRITE, WIDTH>), dim3(blocks), dim3(256), 0, 0, d, iters, stride, off);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipFree(d);
  // wave-instructions per CU: 16 waves * iters * 16; cycles at 2.4 GHz
  const double instr_per_cu = 16.0 * iters * 16;
  return ms * 1e-3 * 2.4e9 / instr_per_cu;
}

int main()
{
  printf("cycles per wave-instruction per CU (16 waves/CU resident, 2.4 GHz assumed)\n");
  for (uint32_t stride : {4u, 5u, 12u, 13u, 37u}) {
    for (uint32_t off : {0u, 1u, 2u, 3u}) {
      printf("stride %2u off %u: write_b32 %.2f  read_b32 %.2f  write_b8 %.2f  read_u8 %.2f\n", stride, off,
             run<1, 4>(stride, off), run<0, 4>(stride, off), run<1, 1>(stride, off), run<0, 1>(stride, off));
    }
  }
  return 0;
}
