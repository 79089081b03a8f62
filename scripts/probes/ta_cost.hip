// What one vector-memory instruction costs the CU's address/L1 path on gfx950, by access shape.
// Build: hipcc -O3 --offload-arch=gfx950 ta_cost.hip -o ta_cost ; run on the GPU box. Measurement aid, not product code.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed)) P4 { uint32_t v[4]; };
struct __attribute__((packed)) P1 { uint32_t v; };

constexpr int kIters = 2000;
constexpr uint32_t kRegion = 65536;

__device__ __forceinline__ uint32_t rnd(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int MODE>
__global__ void __launch_bounds__(256) probe(const uint8_t* __restrict__ base, uint8_t* __restrict__ out, uint32_t* sink)
{
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint8_t* src = base + (size_t)wave * kRegion;
  uint8_t* dst = out + (size_t)wave * kRegion;
  uint32_t s = wave * 977 + lane * 131 + 7;
  uint32_t acc = 0;
  for (int it = 0; it < kIters; ++it) {
    const uint32_t r = rnd(s) % (kRegion - 64);
    const uint32_t u = (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
    if (MODE == 0) { const P4* p = (const P4*)(src + r); acc ^= p->v[0] ^ p->v[3]; }                       // x4, 64 lanes random
    if (MODE == 1) { if ((lane & 3) == 0) { const P4* p = (const P4*)(src + r); acc ^= p->v[0] ^ p->v[3]; } } // x4, 16 lanes (one per quad)
    if (MODE == 2) { if (lane < 16) { const P4* p = (const P4*)(src + r); acc ^= p->v[0] ^ p->v[3]; } }       // x4, 16 lanes (lanes 0-15)
    if (MODE == 3) { const P4* p = (const P4*)(src + u + lane); acc ^= p->v[0] ^ p->v[3]; }                // x4, byte stride
    if (MODE == 4) { const P4* p = (const P4*)(src + (u & ~15u) % (kRegion - 1024) + 16 * lane); acc ^= p->v[0] ^ p->v[3]; } // x4 coalesced
    if (MODE == 5) { const P1* p = (const P1*)(src + r); acc ^= p->v; }                                     // dword, 64 lanes random
    if (MODE == 6) { const P1* p = (const P1*)(src + u + lane); acc ^= p->v; }                              // dword, byte stride
    if (MODE == 7) { if (lane < 8) { dst[r] = (uint8_t)it; } }                                              // byte store, 8 lanes random
    if (MODE == 8) { if (lane < 8) { ((P1*)(dst + r))->v = it; } }                                          // dword store, 8 lanes random
    if (MODE == 9) { dst[u + lane] = (uint8_t)it; }                                                         // byte store, 64 lanes consecutive
    if (MODE == 10) { ((P1*)(dst + (u & ~3u) % (kRegion - 256) + 4 * lane))->v = it; }                      // dword store coalesced
    if (MODE == 11) { if (lane < 8) { const P4* p = (const P4*)(src + r); acc ^= p->v[0] ^ p->v[3]; } }       // x4, 8 lanes
    if (MODE == 12) { const uint8_t* p = src + r; acc ^= *p; }                                               // byte load, 64 random
    if (MODE == 13) { const uint8_t* p = src + u + lane; acc ^= *p; }                                        // byte load, consecutive
  }
  if (acc == 0x12345678u) { sink[0] = acc; }
}

template <int MODE>
double run(const uint8_t* in, uint8_t* out, uint32_t* sink, int blocks)
{
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, sink);
  hipEventRecord(a);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main()
{
  const int cus = 256, waves_per_cu = 20;                  // the compressor's residency
  const int blocks = cus * waves_per_cu / 4;
  const size_t bytes = (size_t)blocks * 4 * kRegion;
  uint8_t *in, *out; uint32_t* sink;
  hipMalloc(&in, bytes); hipMalloc(&out, bytes); hipMalloc(&sink, 4);
  hipMemset(in, 1, bytes);
  const char* names[] = {"x4 load, 64 lanes random", "x4 load, 16 lanes (1 per quad) random", "x4 load, lanes 0-15 random", "x4 load, byte stride",
    "x4 load coalesced", "dword load, 64 random", "dword load, byte stride", "byte store, 8 lanes random", "dword store, 8 lanes random",
    "byte store, 64 consecutive", "dword store coalesced", "x4 load, 8 lanes random", "byte load, 64 random", "byte load, 64 consecutive"};
  double ms[14];
  ms[0] = run<0>(in, out, sink, blocks); ms[1] = run<1>(in, out, sink, blocks); ms[2] = run<2>(in, out, sink, blocks);
  ms[3] = run<3>(in, out, sink, blocks); ms[4] = run<4>(in, out, sink, blocks); ms[5] = run<5>(in, out, sink, blocks);
  ms[6] = run<6>(in, out, sink, blocks); ms[7] = run<7>(in, out, sink, blocks); ms[8] = run<8>(in, out, sink, blocks);
  ms[9] = run<9>(in, out, sink, blocks); ms[10] = run<10>(in, out, sink, blocks); ms[11] = run<11>(in, out, sink, blocks);
  ms[12] = run<12>(in, out, sink, blocks); ms[13] = run<13>(in, out, sink, blocks);
  const double clk = 2.4e9;
  for (int m = 0; m < 14; ++m) {
    const double inst_per_cu = (double)waves_per_cu * kIters;
    printf("{\"shape\": \"%s\", \"ms\": %.3f, \"cycles_per_wave_instruction_per_CU\": %.1f}\n", names[m], ms[m], ms[m] * 1e-3 * clk / inst_per_cu);
  }
  return 0;
}
