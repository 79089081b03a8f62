// What bounds the per-chunk CRC-32 kernel of hlif/manager.hip (one wave per chunk): its loads or its table lookups?
// Every variant checksums the same 16 384 chunks x 64 KiB (and the same chunks at ragged sizes / unaligned starts):
//   seg16 / seg32 / seg64: bytes of a tile one lane owns (hlif/crc32.hip.h) -- 16 = fully coalesced loads but a skip (4 lookups)
//                          per 16 bytes, 64 = four loads per lane that each touch all 64 lines of a tile, a skip per 64 bytes
//   loads only:            the same loads, the state is just XORed (no LDS)
//   lookups only:          the same lookups on register data (no loads)
// build: hipcc -O3 --offload-arch=gfx950 -Invcomp_amd/csrc scripts/probes/crc_bench.hip -o scripts/probes/crc_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "hlif/crc32.hip.h"

template <uint32_t SEG, int MODE>
__device__ __forceinline__ uint32_t variant(const uint8_t* p, uint32_t n, const uint32_t* lds)
{
  if (MODE == 0) {
    return crc32w::wave_crc32<SEG>(p, n, lds);
  }
  constexpr uint32_t kTile = 64 * SEG;
  const uint32_t lane = (uint32_t)wave::lane_id();
  const uint32_t tiles = n / kTile;
  uint32_t s = lane;
  for (uint32_t t = 0; t < tiles; ++t) {
    if (MODE == 2 && t != 0) {
      s = crc32w::lookup4(lds + 1024, s);
    }
    wave::u32x4 w[SEG / 16];
#pragma unroll
    for (uint32_t i = 0; i < SEG / 16; ++i) {
      if (MODE == 1) {
        w[i] = wave::gload_u32x4(p + t * kTile + lane * SEG + 16 * i);
      } else {
        w[i].x = t, w[i].y = lane, w[i].z = i, w[i].w = 7;
      }
    }
#pragma unroll
    for (uint32_t i = 0; i < SEG / 16; ++i) {
      if (MODE == 1) {
        s ^= w[i].x ^ w[i].y ^ w[i].z ^ w[i].w;
      } else {
        s = crc32w::step_dword(lds, s, w[i].x), s = crc32w::step_dword(lds, s, w[i].y);
        s = crc32w::step_dword(lds, s, w[i].z), s = crc32w::step_dword(lds, s, w[i].w);
      }
    }
  }
  return s;
}

template <uint32_t SEG, int MODE, unsigned WAVES>
__global__ void __launch_bounds__(64 * WAVES) crc_kernel(const uint8_t* base, const uint32_t* offs, const uint32_t* sizes, uint32_t n, uint32_t* out)
{
  __shared__ uint32_t tables[crc32w::kLdsDwords];
  crc32w::load_tables<SEG>(tables);
  __syncthreads();
  const uint32_t i = blockIdx.x * WAVES + wave::uniform(threadIdx.x >> 6);
  if (i < n) {
    const uint32_t c = variant<SEG, MODE>(base + (size_t)offs[i] * 16, wave::uniform(sizes[i]), tables);
    if (wave::lane_id() == 0) {
      out[i] = c;
    }
  }
}

static uint32_t host_crc(const uint8_t* p, size_t n)
{
  static uint32_t t[256];
  if (!t[1]) {
    for (uint32_t b = 0; b < 256; ++b) {
      uint32_t c = b;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
      t[b] = c;
    }
  }
  uint32_t s = 0xffffffffu;
  for (size_t i = 0; i < n; ++i) s = t[(s ^ p[i]) & 0xff] ^ (s >> 8);
  return ~s;
}

template <uint32_t SEG, int MODE, unsigned WAVES>
static void run(const char* name, const uint8_t* d, const uint32_t* offs, const uint32_t* sizes, uint32_t n, uint32_t* out, size_t bytes,
                const std::vector<uint32_t>* expect)
{
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const unsigned blocks = (n + WAVES - 1) / WAVES;
  for (int i = 0; i < 30; ++i) hipLaunchKernelGGL((crc_kernel<SEG, MODE, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, d, offs, sizes, n, out);
  hipEventRecord(e0);
  const int reps = 100;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((crc_kernel<SEG, MODE, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, d, offs, sizes, n, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const char* ok = "";
  if (expect) {
    std::vector<uint32_t> got(n);
    hipMemcpy(got.data(), out, 4 * n, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (uint32_t i = 0; i < n; ++i) bad += got[i] != (*expect)[i];
    ok = bad ? " MISMATCH" : " ok";
  }
  printf("{\"variant\": \"%s\", \"seg\": %u, \"waves_per_wg\": %u, \"ms\": %.4f, \"GBps\": %.1f, \"check\": \"%s\"}\n", name, SEG, WAVES, ms,
         bytes / ms / 1e6, ok + (*ok ? 1 : 0));
  fflush(stdout);
}

int main()
{
  const uint32_t n = 16384, chunk = 65536;
  const size_t total = (size_t)n * chunk;
  std::vector<uint8_t> h(total + 64);
  uint64_t x = 88172645463325252ull;
  for (size_t i = 0; i < h.size(); i += 8) {
    x ^= x << 13, x ^= x >> 7, x ^= x << 17;
    *(uint64_t*)&h[i] = x;
  }
  uint8_t* d;
  hipMalloc(&d, h.size());
  hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
  for (int ragged = 0; ragged < 2; ++ragged) {
    // ragged: the sizes of compressed chunks (28-40 KiB, any byte count), packed at 16-byte boundaries as the HLIF buffer packs them
    std::vector<uint32_t> offs(n), sizes(n);
    size_t bytes = 0, at = 0;
    for (uint32_t i = 0; i < n; ++i) {
      sizes[i] = ragged ? 28000 + (uint32_t)((i * 2654435761u) >> 20) % 12000 : chunk;
      offs[i] = (uint32_t)(at / 16);
      at += ragged ? (sizes[i] + 15) / 16 * 16 : chunk;
      bytes += sizes[i];
    }
    std::vector<uint32_t> expect(n);
    for (uint32_t i = 0; i < n; i += 97) expect[i] = host_crc(&h[(size_t)offs[i] * 16], sizes[i]);
    uint32_t *doffs, *dsizes, *dout;
    hipMalloc(&doffs, 4 * n), hipMalloc(&dsizes, 4 * n), hipMalloc(&dout, 4 * n);
    hipMemcpy(doffs, offs.data(), 4 * n, hipMemcpyHostToDevice);
    hipMemcpy(dsizes, sizes.data(), 4 * n, hipMemcpyHostToDevice);
    printf("{\"workload\": \"%s\", \"chunks\": %u, \"bytes\": %zu}\n", ragged ? "ragged 28-40 KB" : "64 KiB", n, bytes);
    // reference values of every chunk from the seg64 kernel once the sampled ones agree with the host
    hipLaunchKernelGGL((crc_kernel<64, 0, 4>), dim3(n / 4), dim3(256), 0, 0, d, doffs, dsizes, n, dout);
    std::vector<uint32_t> ref(n);
    hipMemcpy(ref.data(), dout, 4 * n, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (uint32_t i = 0; i < n; i += 97) bad += ref[i] != expect[i];
    printf("{\"seg64 against the host CRC on every 97th chunk\": \"%s\"}\n", bad ? "MISMATCH" : "ok");
    run<64, 0, 4>("full", d, doffs, dsizes, n, dout, bytes, &ref);
    run<32, 0, 4>("full", d, doffs, dsizes, n, dout, bytes, &ref);
    run<16, 0, 4>("full", d, doffs, dsizes, n, dout, bytes, &ref);
    run<64, 0, 8>("full", d, doffs, dsizes, n, dout, bytes, &ref);
    run<32, 0, 8>("full", d, doffs, dsizes, n, dout, bytes, &ref);
    run<16, 0, 8>("full", d, doffs, dsizes, n, dout, bytes, &ref);
    run<64, 0, 2>("full", d, doffs, dsizes, n, dout, bytes, &ref);
    run<32, 0, 2>("full", d, doffs, dsizes, n, dout, bytes, &ref);
    if (!ragged) {
      run<64, 1, 4>("loads only", d, doffs, dsizes, n, dout, bytes, nullptr);
      run<32, 1, 4>("loads only", d, doffs, dsizes, n, dout, bytes, nullptr);
      run<16, 1, 4>("loads only", d, doffs, dsizes, n, dout, bytes, nullptr);
      run<64, 2, 4>("lookups only", d, doffs, dsizes, n, dout, bytes, nullptr);
      run<32, 2, 4>("lookups only", d, doffs, dsizes, n, dout, bytes, nullptr);
      run<16, 2, 4>("lookups only", d, doffs, dsizes, n, dout, bytes, nullptr);
    }
    hipFree(doffs), hipFree(dsizes), hipFree(dout);
  }
  return 0;
}
