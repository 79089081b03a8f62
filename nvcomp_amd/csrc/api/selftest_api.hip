/*
 * api/selftest_api.hip -- a tiny exported kernel that exercises the wave64
 * primitives of common/wave.h on known inputs, so a failure on real hardware can
 * be told apart from a codec bug (tests/test_wave_primitives.py).
 */
#include <hip/hip_runtime.h>

#include "nvcomp/shared_types.h"

#include "common/lz_common.hip.h"

namespace {

__global__ void __launch_bounds__(64) wave_selftest_kernel(const uint32_t* in, uint32_t* out, uint8_t* scratch)
{
  const uint32_t lane = (uint32_t)wave::lane_id();
  const uint32_t v = in[lane];
  out[0 * 64 + lane] = wave::scan_add_inclusive(v);
  out[1 * 64 + lane] = wave::reduce_max(v);
  out[2 * 64 + lane] = wave::reduce_add(v);
  out[3 * 64 + lane] = wave::shuffle(v, (lane * 7 + 3) & 63);
  out[4 * 64 + lane] = wave::read_lane(v, 37);
  const uint64_t b = wave::ballot((v & 1) != 0);
  out[5 * 64 + lane] = (uint32_t)b;
  out[6 * 64 + lane] = (uint32_t)(b >> 32);
  out[7 * 64 + lane] = wave::write_lane(v, 0xabcdu, 11);
  /* same-wave cross-lane read-after-write through global memory, no s_waitcnt */
  scratch[lane] = (uint8_t)(v + 1);
  wave::sync();
  out[8 * 64 + lane] = scratch[63 - lane];
  /* overlapping match copy with a short period and the pattern-doubling path */
  if (lane < 3) {
    scratch[64 + lane] = (uint8_t)(10 + lane);
  }
  wave::sync();
  lz::wave_match_copy(scratch + 67, 3, 3000);
  wave::sync();
  uint32_t bad = 0;
  for (uint32_t i = lane; i < 3003; i += 64) {
    bad += scratch[64 + i] != (uint8_t)(10 + i % 3);
  }
  out[9 * 64 + lane] = wave::reduce_add(bad);
}

} // namespace

extern "C" nvcompStatus_t nvcompAmdSelfTestWave(const uint32_t* device_in64, uint32_t* device_out640,
                                                 uint8_t* device_scratch4096, hipStream_t stream)
{
  (void)hipGetLastError(); /* drop stale sticky errors of earlier, unrelated runtime calls */
  hipLaunchKernelGGL(wave_selftest_kernel, dim3(1), dim3(64), 0, stream, device_in64, device_out640,
                     device_scratch4096);
  return hipGetLastError() == hipSuccess ? nvcompSuccess : nvcompErrorCudaError;
}
