/*
 * common/wave.h -- wave64 cross-lane primitives for gfx950 (CDNA4).
 *
 * Every codec kernel in this library runs "one wavefront per chunk": the 64
 * lanes of a wave cooperate through the operations below and never through
 * __syncthreads(). All of them assume full 64-wide wavefronts and must be
 * called from wave-uniform control flow.
 */
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wave {

constexpr int kSize = 64;

/* 16-byte register quad for naturally aligned LDS / global accesses
 * (ds_read_b128 / global_load_dwordx4). */
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

/* HBM pointers arrive through pointer arrays, so the compiler cannot prove their
 * address space and would emit flat_* accesses (which also occupy the LDS
 * counter). These casts state it: global_load / global_store. */
#define WAVE_GLOBAL __attribute__((address_space(1)))
struct __attribute__((packed)) PackedU32
{
  uint32_t v;
};
struct __attribute__((packed)) PackedU32x4
{
  uint32_t v[4];
};
struct __attribute__((packed)) PackedU32x2
{
  uint32_t v[2];
};
__device__ __forceinline__ uint32_t gload_u8(const uint8_t* p)
{
  return *(const WAVE_GLOBAL uint8_t*)p;
}
__device__ __forceinline__ uint32_t gload_u32(const uint8_t* p) /* any alignment */
{
  return ((const WAVE_GLOBAL PackedU32*)p)->v;
}
__device__ __forceinline__ u32x4 gload_u32x4_aligned(const uint8_t* p)
{
  return *(const WAVE_GLOBAL u32x4*)p;
}
__device__ __forceinline__ u32x4 gload_u32x4(const uint8_t* p) /* any alignment (global_load_dwordx4) */
{
  const WAVE_GLOBAL PackedU32x4* q = (const WAVE_GLOBAL PackedU32x4*)p;
  u32x4 r = {q->v[0], q->v[1], q->v[2], q->v[3]};
  return r;
}

struct __attribute__((packed)) PackedU32x3
{
  uint32_t v[3];
};
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ u32x3 gload_u32x3(const uint8_t* p) /* any alignment (global_load_dwordx3) */
{
  const WAVE_GLOBAL PackedU32x3* q = (const WAVE_GLOBAL PackedU32x3*)p;
  u32x3 r = {q->v[0], q->v[1], q->v[2]};
  return r;
}

__device__ __forceinline__ uint64_t gload_u64(const uint8_t* p) /* any alignment (global_load_dwordx2) */
{
  const WAVE_GLOBAL PackedU32x2* q = (const WAVE_GLOBAL PackedU32x2*)p;
  return ((uint64_t)q->v[1] << 32) | q->v[0];
}
__device__ __forceinline__ uint32_t gload_u16(const uint16_t* p)
{
  return *(const WAVE_GLOBAL uint16_t*)p;
}
__device__ __forceinline__ void gstore_u8(uint8_t* p, uint32_t v)
{
  *(WAVE_GLOBAL uint8_t*)p = (uint8_t)v;
}
__device__ __forceinline__ void gstore_u32x4_aligned(uint8_t* p, u32x4 v)
{
  *(WAVE_GLOBAL u32x4*)p = v;
}
/* the same, non-temporal: for output that nothing on the card reads again soon (a streamed literal run) */
__device__ __forceinline__ void gstore_u32x4_aligned_nt(uint8_t* p, u32x4 v)
{
  __builtin_nontemporal_store(v, (WAVE_GLOBAL u32x4*)p);
}
__device__ __forceinline__ void gstore_u32x4(uint8_t* p, u32x4 v) /* any alignment (global_store_dwordx4) */
{
  WAVE_GLOBAL PackedU32x4* q = (WAVE_GLOBAL PackedU32x4*)p;
  q->v[0] = v.x, q->v[1] = v.y, q->v[2] = v.z, q->v[3] = v.w;
}
__device__ __forceinline__ void gstore_u32(uint8_t* p, uint32_t v) /* any alignment */
{
  ((WAVE_GLOBAL PackedU32*)p)->v = v;
}

__device__ __forceinline__ int lane_id()
{
  return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

/* The same value, computed where it is asked for. The compiler treats lane_id() as one loop-invariant value and hoists
 * everything derived from it (16 * lane, lds + 4 * lane, 1ull << lane, ...) to the top of the kernel: a dozen
 * registers held across the whole decode loop of the LZ kernels, enough to push them into scratch -- and a scratch
 * reload is a vector memory load: the s_waitcnt vmcnt(0) in front of its first use also waits for every store the
 * previous batch's flush left in flight. Code that runs once per batch or less takes its lane id from here: two VALU
 * instructions, nothing kept live. */
__device__ __forceinline__ int fresh_lane_id()
{
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

/* "This value is used here": a register that a load is still filling is waited for at this point, not at its real use
 * further down (the compiler places s_waitcnt in front of the first use). */
__device__ __forceinline__ void touch(uint32_t& v)
{
  asm volatile("" : "+v"(v));
}

/* Nothing is scheduled across this point: unrolled bodies stay one after the other instead of being interleaved (and their
 * temporaries live all at once). */
__device__ __forceinline__ void sched_fence()
{
  __builtin_amdgcn_sched_barrier(0);
}

/* Is the calling lane's bit set in the wave-uniform mask? (v_cndmask with the mask as its condition: no 1ull << lane) */
__device__ __forceinline__ bool lane_in(uint64_t mask)
{
  uint32_t r;
  asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(r) : "s"(mask));
  return r != 0;
}

/* 64-bit mask of lanes whose predicate is true. */
__device__ __forceinline__ uint64_t ballot(bool pred)
{
  return __builtin_amdgcn_ballot_w64(pred);
}

/* Value of lane `lane` (wave-uniform index) broadcast to the scalar unit. */
__device__ __forceinline__ uint32_t read_lane(uint32_t v, uint32_t lane)
{
  return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane);
}

/* Tell the compiler a value is wave-uniform (it moves to an SGPR). */
__device__ __forceinline__ uint32_t uniform(uint32_t v)
{
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

__device__ __forceinline__ uint64_t uniform64(uint64_t v)
{
  const uint32_t lo = uniform((uint32_t)v);
  const uint32_t hi = uniform((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

/* A wave-uniform pointer to DEVICE memory (every chunk pointer of a batch is one): it is rebuilt through the global
 * address space, so that the compiler addresses everything derived from it with global_load / global_store. A pointer
 * read out of a pointer array is otherwise generic, and flat_* instructions pay the LDS aperture check and count on
 * lgkmcnt as well as vmcnt. */
template <typename T>
__device__ __forceinline__ T* uniform_ptr(T* p)
{
  return (T*)(WAVE_GLOBAL T*)uniform64((uint64_t)p);
}

/* The kernel's argument block, read where it is USED. A kernel whose single parameter is the struct T gets it in the
 * kernarg segment; the compiler normally loads every field once and keeps all of them in SGPRs for the kernel's lifetime --
 * in a persistent kernel that is a dozen scalar registers held across the whole decode loop for values needed once per
 * chunk (the LZ4 window kernel went from 7 to 34 spilled SGPRs and from 6 to 15 spilled VGPRs that way). The pointer
 * returned here is opaque to the optimiser at every call, so a field read through it is an s_load at that point.
 * `first_param` must be the kernel's only parameter. */
#define WAVE_CONSTANT __attribute__((address_space(4)))
template <typename T>
__device__ __forceinline__ const WAVE_CONSTANT T* kernel_args(const T& first_param)
{
  (void)first_param;
  uint64_t p = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return (const WAVE_CONSTANT T*)p;
}

/* `vec` with lane `lane` (wave-uniform) replaced by the uniform value `val`. */
__device__ __forceinline__ uint32_t write_lane(uint32_t vec, uint32_t val, uint32_t lane)
{
  return ((uint32_t)lane_id() == lane) ? val : vec;
}

/* Same through v_writelane_b32: one VALU op instead of three. A VALU op may read
 * only one SGPR, so the lane select travels in M0 (SALU write of M0, one wait
 * state, then the lane write). M0 is a reserved register the compiler only uses
 * around LDS-DMA / s_movrel / message instructions, none of which this library emits. */
__device__ __forceinline__ uint32_t write_lane_scalar(uint32_t vec, uint32_t val, uint32_t lane)
{
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(vec) : "s"(val), "s"(lane) : "m0");
  return vec;
}

/*
 * Walk a chain of relative steps held one per lane: starting at position r (< limit <= 64),
 * repeat { rec[k++] = r; r += step[r]; } while (r < limit). This is the serial critical
 * path of the LZ decoders (one iteration per token), so it is written out: 2 VALU
 * (v_readlane, v_writelane) + 4 SALU per iteration, the lane counter living in M0. The
 * CU's single scalar ALU is the resource the decoders run out of first.
 * Requires k + (iterations) <= 64. r and k are updated in place.
 */
__device__ __forceinline__ void chain_walk(uint32_t step, uint32_t limit, uint32_t& r, uint32_t& k, uint32_t& rec)
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
      : "scc", "m0");
}

/*
 * Greedy walk over a mask of candidate lanes, each with a length below 64 in `len`: from lane `start` (< 64, and at least
 * one candidate at or above it) repeat { f = the lowest candidate at or above the current lane; take it; go on at
 * f + len[f] }. Returns the taken lanes in `taken` and where the walk stopped (f + len[f] of the last one taken) in `end`.
 * The serial part of the LZ compressors' match selection, one iteration per selected match, written out: 7 SALU + 1 VALU
 * an iteration (the compiler's version of the same loop: 16).
 */
__device__ __forceinline__ void select_walk(uint64_t candidates, uint32_t len, uint32_t start, uint64_t& taken, uint32_t& end)
{
  uint64_t r;
  uint32_t t, l;
  asm volatile(
      "s_lshr_b64 %[r], %[cand], %[start]\n\t"
      "s_mov_b32 %[pos], %[start]\n\t"
      "s_mov_b64 %[taken], 0\n"
      "1:\n\t"
      "s_ff1_i32_b64 %[t], %[r]\n\t"
      "s_add_u32 %[pos], %[pos], %[t]\n\t"
      "s_lshr_b64 %[r], %[r], %[t]\n\t"
      "v_readlane_b32 %[l], %[len], %[pos]\n\t"
      "s_bitset1_b64 %[taken], %[pos]\n\t"
      "s_add_u32 %[pos], %[pos], %[l]\n\t"
      "s_lshr_b64 %[r], %[r], %[l]\n\t"
      "s_cbranch_scc1 1b"
      : [taken] "=&s"(taken), [pos] "=&s"(end), [r] "=&s"(r), [t] "=&s"(t), [l] "=&s"(l)
      : [cand] "s"(candidates), [len] "v"(len), [start] "s"(start)
      : "scc");
}

/* Per-lane gather: lane i receives v of lane src_lane(i) (ds_bpermute_b32). */
__device__ __forceinline__ uint32_t shuffle(uint32_t v, uint32_t src_lane)
{
  return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v);
}

/* Per-lane scatter: lane i's v goes to lane dst_lane(i) (ds_permute_b32). dst_lane must be a permutation of the wave (a lane
 * that nobody writes to reads as 0 on the hardware; the callers do not rely on it). */
__device__ __forceinline__ uint32_t permute_to(uint32_t v, uint32_t dst_lane)
{
  return (uint32_t)__builtin_amdgcn_ds_permute((int)(dst_lane << 2), (int)v);
}

/* v of the lane below (lane 0: 0): one DPP move across the whole wave (wave_shr:1), no LDS crossbar round trip. */
__device__ __forceinline__ uint32_t prev_lane(uint32_t v)
{
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true);
}

/* v of the lane above (lane 63: 0): wave_shl:1. */
__device__ __forceinline__ uint32_t next_lane(uint32_t v)
{
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);
}

/* Inclusive prefix sum across the wave: 4 row_shr steps inside each row of 16
 * lanes, then row_bcast:15 / row_bcast:31 to carry across rows (DPP, no LDS). */
__device__ __forceinline__ uint32_t scan_add_inclusive(uint32_t v)
{
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); /* row_shr:1 */
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); /* row_shr:2 */
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); /* row_shr:4 */
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); /* row_shr:8 */
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); /* row_bcast:15 */
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); /* row_bcast:31 */
  return v;
}

__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b)
{
  return a > b ? a : b;
}

/* Wave-wide unsigned maximum, returned as a uniform value. Same DPP ladder as
 * the scan; lane 63 ends up holding the reduction. */
__device__ __forceinline__ uint32_t reduce_max(uint32_t v)
{
  v = umax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
  v = umax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
  v = umax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));
  v = umax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));
  v = umax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
  v = umax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
  return read_lane(v, 63);
}

/* Inclusive running maximum across the wave (same DPP ladder). */
__device__ __forceinline__ uint32_t scan_max_inclusive(uint32_t v)
{
  v = umax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
  v = umax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
  v = umax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));
  v = umax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));
  v = umax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
  v = umax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
  return v;
}

/* Inclusive "last writer" scan across the wave, bit by bit: lane k ends up with, for every bit position, the value bit
 * of the highest lane j <= k whose mask has that bit set, and with the OR of the masks of lanes 0 .. k. Value bits are
 * kept zero where the mask is zero. Same DPP ladder as the sums (the operator is associative, not commutative: a lane
 * combines what it receives from below with its own). The run executor of the LZ decoders carries the 16-byte pattern
 * of a run from sequence to sequence with it (common/lz_window.hip.h: execute_run_batch). */
__device__ __forceinline__ void scan_last_writer(uint32_t& val, uint32_t& mask)
{
  val &= mask;
#define NVCOMP_WAVE_LW_STEP(ctrl, rows)                                                                      \
  {                                                                                                          \
    const uint32_t pm = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mask, ctrl, rows, 0xf, false);         \
    const uint32_t pv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)val, ctrl, rows, 0xf, false);          \
    val = (val & mask) | (pv & ~mask);                                                                       \
    mask |= pm;                                                                                              \
  }
  NVCOMP_WAVE_LW_STEP(0x111, 0xf) /* row_shr:1 */
  NVCOMP_WAVE_LW_STEP(0x112, 0xf) /* row_shr:2 */
  NVCOMP_WAVE_LW_STEP(0x114, 0xf) /* row_shr:4 */
  NVCOMP_WAVE_LW_STEP(0x118, 0xf) /* row_shr:8 */
  NVCOMP_WAVE_LW_STEP(0x142, 0xa) /* row_bcast:15 */
  NVCOMP_WAVE_LW_STEP(0x143, 0xc) /* row_bcast:31 */
#undef NVCOMP_WAVE_LW_STEP
}

__device__ __forceinline__ uint32_t reduce_add(uint32_t v)
{
  return read_lane(scan_add_inclusive(v), 63);
}

/*
 * Order this wave's earlier memory writes (LDS or global) before its later
 * reads, across lanes. One wave's LDS and vector-memory operations are served
 * in issue order by the CU's LDS and L1, so no s_waitcnt is needed for a
 * same-wave cross-lane read-after-write; what has to be stopped is the
 * compiler moving a load above a store it believes cannot alias.
 */
/* The fences are at WAVEFRONT scope (round 5): what is asked for is the compiler's order -- the hardware keeps a wave's
 * LDS and vector-memory operations in issue order by itself. At workgroup scope (rounds 1-4) every release drained the
 * wave's LDS queue (s_waitcnt lgkmcnt(0)) in front of each of these points: measured on the decoders with the same sources,
 * mix 654 -> 668 GB/s, 4 096 chunks 305 -> 316, sorted-key column 1 668 -> 1 720 (gpurun r5b). Waves of one workgroup
 * that hand data to each other do it through lds_store_release / lds_load_acquire or a workgroup barrier, which carry
 * their own scope. NVCOMP_WAVE_SYNC_WORKGROUP=1: the old behaviour (A/B build). */
#if defined(NVCOMP_WAVE_SYNC_WORKGROUP) && NVCOMP_WAVE_SYNC_WORKGROUP
#define NVCOMP_WAVE_SYNC_SCOPE "workgroup"
#else
#define NVCOMP_WAVE_SYNC_SCOPE "wavefront"
#endif
__device__ __forceinline__ void sync()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, NVCOMP_WAVE_SYNC_SCOPE);
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, NVCOMP_WAVE_SYNC_SCOPE);
}

/* The same, always at wavefront scope (the compressors' per-wave LDS: nothing another wave reads). */
__device__ __forceinline__ void sync_wave()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* LDS word |= bits, no return value (ds_or_b32): lanes of one instruction may hit the same word. */
__device__ __forceinline__ void lds_or(uint32_t* p, uint32_t bits)
{
  __hip_atomic_fetch_or(p, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

/* LDS word += / -= v, no return value (ds_add_u32 / ds_sub_u32): lanes of one instruction, and several waves, may hit the same word. */
__device__ __forceinline__ void lds_add(uint32_t* p, uint32_t v)
{
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_sub(uint32_t* p, uint32_t v)
{
  __hip_atomic_fetch_sub(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

/* ---- two waves of one workgroup handing work to each other through LDS (producer / consumer) ----
 * A flag word is written after, and read before, the data it guards; both sides are single instruction streams whose
 * LDS operations are served in issue order, the release / acquire pair stops the compiler and drains the counters. */
__device__ __forceinline__ void lds_store_release(uint32_t* p, uint32_t v)
{
  __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t lds_load_acquire(const uint32_t* p)
{
  return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
/* A flag word other waves of the workgroup poll, written / read without the memory-model ceremony: one wave's LDS
 * operations are served in issue order, so data written in front of the flag is in LDS before it, and a reader that saw the
 * flag reads the data behind it; wave::sync() around these keeps the compiler from reordering. */
__device__ __forceinline__ void lds_store_relaxed(uint32_t* p, uint32_t v)
{
  *(volatile uint32_t*)p = v;
}
__device__ __forceinline__ uint32_t lds_load_relaxed(const uint32_t* p)
{
  return *(const volatile uint32_t*)p;
}
/* Give the SIMD to the other waves for a few hundred cycles while polling. */
__device__ __forceinline__ void nap()
{
  __builtin_amdgcn_s_sleep(4);
}
/* ... for 64 cycles */
__device__ __forceinline__ void nap_short()
{
  __builtin_amdgcn_s_sleep(1);
}

/* Trailing-zero / popcount helpers on ballots. */
__device__ __forceinline__ uint32_t ctz64(uint64_t m)
{
  return (uint32_t)__builtin_ctzll(m);
}

__device__ __forceinline__ uint32_t popc64(uint64_t m)
{
  return (uint32_t)__builtin_popcountll(m);
}

/* a * b for operands below 2^24 (v_mul_u32_u24: full rate; a 32-bit v_mul_lo_u32 takes four times as long, and
 * __umul24 compiles to a mask and that). */
__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b)
{
  uint32_t r;
  asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

/* (hi:lo) >> shift, low 32 bits, shift = 0 ... 31 (v_alignbit_b32). */
__device__ __forceinline__ uint32_t align_bits(uint32_t hi, uint32_t lo, uint32_t shift)
{
  return __builtin_amdgcn_alignbit(hi, lo, shift);
}

/* Bytes [shift, shift + 4) of the 8-byte value hi:lo, shift in 0..3 (v_alignbyte_b32). */
__device__ __forceinline__ uint32_t align_bytes(uint32_t hi, uint32_t lo, uint32_t shift)
{
  return __builtin_amdgcn_alignbyte(hi, lo, shift);
}

/* Two 16-bit lanes per dword: min(a + b, 255) in each (v_pk_add_u16 + v_pk_min_u16). */
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_add_sat255(uint32_t a, uint32_t b)
{
  u16x2 x = __builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b);
  const u16x2 cap = {255, 255};
  x = __builtin_elementwise_min(x, cap);
  return __builtin_bit_cast(uint32_t, x);
}

/* v_bfrev_b32 */
__device__ __forceinline__ uint32_t bit_reverse(uint32_t v)
{
  return __builtin_bitreverse32(v);
}

/* Byte permute (v_perm_b32): result byte i = byte sel[i] of the 8-byte value hi:lo (0-3 = lo, 4-7 = hi). */
__device__ __forceinline__ uint32_t perm_bytes(uint32_t hi, uint32_t lo, uint32_t sel)
{
  return __builtin_amdgcn_perm(hi, lo, sel);
}

/* Number of set bits of m below the calling lane (v_mbcnt_lo/hi). */
__device__ __forceinline__ uint32_t prefix_popc(uint64_t m)
{
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

} // namespace wave

/* Algorithm statistics hook (rounds, path counts): compiled out on the device; the
 * host emulation (tests/emu/common/wave.h) turns it into counters for design studies. */
#define LZ_STAT(name, n) ((void)0)

/* The workgroup's dynamically sized LDS segment (size given at launch). */
#define WAVE_DYNAMIC_LDS(name) extern __shared__ __attribute__((aligned(16))) uint8_t name[]
