// What does FETCH_SIZE (rocprofv3 --pmc, gfx950) report for the access shapes of the LZ decoders' far-match gathers?
// MI355X_MICROARCH.md (HBM) calibrates ONE shape -- a wide coalesced streaming read is tallied at 1/2 of its bytes -- and
// calls every other shape uncalibrated. This probe issues a KNOWN number of loads per shape over a 2 GiB buffer (eight
// times the 256 MiB Infinity Cache, so that re-use cannot hide requests) and reports the bytes each kernel asked for;
// scripts/gpu_calib.sh runs it under the counter and divides.
//   stream    : 16 B per lane, consecutive lanes consecutive addresses (the calibrated case: expect counted = bytes / 2)
//   stride64  : 16 B per lane, lanes 64 B apart   (one piece of every 64-byte sector, two of every 128-byte line)
//   stride128 : 16 B per lane, lanes 128 B apart  (one piece of every 128-byte line)
//   gather16  : 16 B per lane at a pseudo-random 16-byte aligned address (never crosses a sector)
//   gather1   : 16 B per lane at a pseudo-random BYTE address (the decoders' far-match loads: 23 % cross a 64-byte sector)
// If a miss fills 128 bytes and is tallied at 64, stride64 counts 32 B per load and stride128 64; if it fills 64-byte
// sectors tallied at 64, both count 64.
// build: hipcc -O3 --offload-arch=gfx950 scripts/probes/gather_calib.hip -o scripts/probes/gather_calib
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed)) P16
{
  uint32_t v[4];
};

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
  x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
  return x;
}

template <int MODE>
__global__ void __launch_bounds__(256) probe(const uint8_t* __restrict__ buf, uint64_t bytes, uint32_t loads_per_lane, uint32_t* sink)
{
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (uint32_t i = 0; i < loads_per_lane; ++i) {
    const uint64_t k = (uint64_t)i * nthreads + tid; // global load index
    uint64_t off;
    if (MODE == 0) {
      off = (k * 16) % bytes;
    } else if (MODE == 1) {
      off = (k * 64) % bytes;
    } else if (MODE == 2) {
      off = (k * 128) % bytes;
    } else {
      const uint64_t r = ((uint64_t)mix((uint32_t)k) << 21) ^ mix((uint32_t)(k >> 11) + 0x9e3779b9u * (uint32_t)k);
      off = r % (bytes - 64);
      if (MODE == 3) {
        off &= ~15ull;
      }
    }
    const P16* p = (const P16*)(buf + off);
    acc ^= p->v[0] ^ p->v[1] ^ p->v[2] ^ p->v[3];
  }
  if (acc == 0x12345679u) { // never: keeps the loads
    sink[tid & 255] = acc;
  }
}

int main(int argc, char** argv)
{
  const uint32_t per_lane = argc > 1 ? (uint32_t)atoi(argv[1]) : 32;
  /* footprint in MiB (default 2 GiB = eight times the Infinity Cache; 128 = inside it: the rate of requests it serves) */
  const uint64_t bytes = (argc > 2 ? (uint64_t)atoi(argv[2]) : 2048ull) << 20;
  uint8_t* buf = nullptr;
  uint32_t* sink = nullptr;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 1024) != hipSuccess) {
    fprintf(stderr, "hipMalloc failed\n");
    return 1;
  }
  hipMemset(buf, 1, bytes);
  const dim3 grid(256 * 8 * 4), block(256); // 2 Mi lanes
  const uint64_t lanes = (uint64_t)grid.x * block.x;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const char* names[5] = {"stream", "stride64", "stride128", "gather16", "gather1"};
  for (int mode = 0; mode < 5; ++mode) {
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      switch (mode) {
      case 0: hipLaunchKernelGGL(probe<0>, grid, block, 0, 0, buf, bytes, per_lane, sink); break;
      case 1: hipLaunchKernelGGL(probe<1>, grid, block, 0, 0, buf, bytes, per_lane, sink); break;
      case 2: hipLaunchKernelGGL(probe<2>, grid, block, 0, 0, buf, bytes, per_lane, sink); break;
      case 3: hipLaunchKernelGGL(probe<3>, grid, block, 0, 0, buf, bytes, per_lane, sink); break;
      default: hipLaunchKernelGGL(probe<4>, grid, block, 0, 0, buf, bytes, per_lane, sink); break;
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    const uint64_t loads = lanes * per_lane;
    printf("{\"probe\": \"%s\", \"footprint_MiB\": %llu, \"mode\": %d, \"loads\": %llu, \"bytes_asked\": %llu, \"ms\": %.3f, \"Gloads_per_s\": %.2f, \"launches\": 3}\n",
           names[mode], (unsigned long long)(bytes >> 20), mode, (unsigned long long)loads, (unsigned long long)(loads * 16), best, loads / best / 1e6);
  }
  hipFree(buf), hipFree(sink);
  return 0;
}
