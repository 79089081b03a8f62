// Microbenchmark (measurement, not product): cycles a wave64 instruction of the kinds the LZ decoders are made of holds its
// SIMD on gfx950, at 1 / 2 / 4 / 7 / 8 waves per SIMD.  Answers VERDICT r5 "Next round" 1(a): MI355X_MICROARCH.md:52-54,430 says
// 2 cycles (SIMD-32), the decode model of rounds 3-5 assumed 4 (16 lanes a cycle).
//
// Every kernel runs ITERS x 64 instructions of ONE kind over 8 independent register chains (no dependent-issue stall can
// hide the rate), takes s_memtime (shader clock) and s_memrealtime (100 MHz) at both ends, and records where it ran
// (HW_REG_HW_ID, HW_REG_XCC_ID).  The host groups the waves by SIMD: cycles per instruction per SIMD =
// (last end - first start) / instructions issued there.  Wall time by HIP events beside it.
//
// build: hipcc -O3 --offload-arch=gfx950 scripts/microbench/valu_issue.hip -o scripts/microbench/valu_issue
// run:   scripts/microbench/valu_issue [iters] [b] > profiles/r06_valu_issue_{a,b}.jsonl   (b: the second table of ops)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

struct Rec {
  uint64_t t0, t1, r0, r1;
  uint32_t hw_id, xcc_id, pad0, pad1;
};

#define OP8(I) I(a0) I(a1) I(a2) I(a3) I(a4) I(a5) I(a6) I(a7)
#define BLOCK64(I) OP8(I) OP8(I) OP8(I) OP8(I) OP8(I) OP8(I) OP8(I) OP8(I)

#define I_ADD(x) "v_add_u32 %[" #x "], %[" #x "], %[b]\n\t"
#define I_AND(x) "v_and_b32 %[" #x "], %[" #x "], %[b]\n\t"
#define I_LSHL(x) "v_lshlrev_b32 %[" #x "], 1, %[" #x "]\n\t"
#define I_LSHLV(x) "v_lshlrev_b32 %[" #x "], %[c], %[" #x "]\n\t"
#define I_CNDMASK(x) "v_cndmask_b32 %[" #x "], %[" #x "], %[b], vcc\n\t"
#define I_ALIGNBYTE(x) "v_alignbyte_b32 %[" #x "], %[" #x "], %[b], %[c]\n\t"
#define I_ALIGNBIT(x) "v_alignbit_b32 %[" #x "], %[" #x "], %[b], %[c]\n\t"
#define I_PERM(x) "v_perm_b32 %[" #x "], %[" #x "], %[b], %[c]\n\t"
#define I_DPP(x) "v_mov_b32_dpp %[" #x "], %[" #x "] row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_DPPWS(x) "v_mov_b32_dpp %[" #x "], %[" #x "] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_ADD_DPP(x) "v_add_u32_dpp %[" #x "], %[" #x "], %[" #x "] row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_READLANE(x) "v_readlane_b32 s20, %[" #x "], 5\n\t"
#define I_READFIRST(x) "v_readfirstlane_b32 s20, %[" #x "]\n\t"
#define I_ADD3(x) "v_add3_u32 %[" #x "], %[" #x "], %[b], %[c]\n\t"
#define I_LSHLADD(x) "v_lshl_add_u32 %[" #x "], %[" #x "], 2, %[b]\n\t"
#define I_BFE(x) "v_bfe_u32 %[" #x "], %[" #x "], 3, 8\n\t"
#define I_MAD24(x) "v_mad_u32_u24 %[" #x "], %[" #x "], %[b], %[c]\n\t"
#define I_MULLO(x) "v_mul_lo_u32 %[" #x "], %[" #x "], %[b]\n\t"
#define I_PKADD16(x) "v_pk_add_u16 %[" #x "], %[" #x "], %[b]\n\t"
#define I_CMP(x) "v_cmp_lt_u32 vcc, %[" #x "], %[b]\n\t"
#define I_CMP_S(x) "v_cmp_lt_u32 s[22:23], %[" #x "], %[b]\n\t"
#define I_MIN(x) "v_min_u32 %[" #x "], %[" #x "], %[b]\n\t"
#define I_FMA(x) "v_fma_f32 %[" #x "], %[" #x "], %[b], %[c]\n\t"
#define I_PKFMA(x) "v_pk_fma_f32 %[" #x "], %[" #x "], %[p], %[p]\n\t"
#define I_MBCNT(x) "v_mbcnt_lo_u32_b32 %[" #x "], %[b], %[" #x "]\n\t"
#define I_BPERM(x) "ds_bpermute_b32 %[" #x "], %[c], %[" #x "]\n\ts_waitcnt lgkmcnt(4)\n\t"
#define I_SADD(x) "s_add_u32 s20, s20, 3\n\t"
#define I_SAND64(x) "s_and_b64 s[22:23], s[22:23], exec\n\t"
#define I_SBCNT(x) "s_bcnt1_i32_b64 s20, s[22:23]\n\t"
/* pairs: a vector and a scalar instruction alternating (the decoders are 8.7 vector + 6.1 scalar per sequence) */
#define I_VS(x) "v_add_u32 %[" #x "], %[" #x "], %[b]\n\ts_add_u32 s20, s20, 3\n\t"
#define I_VSS(x) "v_add_u32 %[" #x "], %[" #x "], %[b]\n\ts_add_u32 s20, s20, 3\n\ts_and_b32 s21, s21, s20\n\t"
/* dependent chain: every instruction reads the one before (latency of a dependent issue) */
#define I_DEP(x) "v_add_u32 %[a0], %[a0], %[b]\n\t"
#define I_DEP_S(x) "s_add_u32 s20, s20, 3\n\t"


/* second pass (r06_valu_issue_b): which VOP2 ops run at the 2-cycle rate, what an SGPR / vcc / literal operand costs,
 * SDWA forms (the compiler's byte extracts), half-empty exec masks, LDS byte accesses */
#define I_OR(x) "v_or_b32 %[" #x "], %[" #x "], %[b]\n\t"
#define I_XOR(x) "v_xor_b32 %[" #x "], %[" #x "], %[b]\n\t"
#define I_SUB(x) "v_sub_u32 %[" #x "], %[" #x "], %[b]\n\t"
#define I_MOV(x) "v_mov_b32 %[" #x "], %[b]\n\t"
#define I_LSHR(x) "v_lshrrev_b32 %[" #x "], 1, %[" #x "]\n\t"
#define I_MAX(x) "v_max_u32 %[" #x "], %[" #x "], %[b]\n\t"
#define I_ANDOR(x) "v_and_or_b32 %[" #x "], %[" #x "], %[b], %[c]\n\t"
#define I_OR3(x) "v_or3_b32 %[" #x "], %[" #x "], %[b], %[c]\n\t"
#define I_BFI(x) "v_bfi_b32 %[" #x "], %[" #x "], %[b], %[c]\n\t"
#define I_ADDCO(x) "v_add_co_u32 %[" #x "], vcc, %[" #x "], %[b]\n\t"
#define I_ADDC(x) "v_addc_co_u32 %[" #x "], vcc, %[" #x "], %[b], vcc\n\t"
#define I_MUL24(x) "v_mul_u32_u24 %[" #x "], %[" #x "], %[b]\n\t"
#define I_ADD16(x) "v_add_u16 %[" #x "], %[" #x "], %[b]\n\t"
#define I_ADD_E64(x) "v_add_u32_e64 %[" #x "], %[" #x "], %[b]\n\t"
#define I_ADD_SGPR(x) "v_add_u32 %[" #x "], s20, %[" #x "]\n\t"
#define I_ADD_LIT(x) "v_add_u32 %[" #x "], 0x12345, %[" #x "]\n\t"
#define I_ADD_INL(x) "v_add_u32 %[" #x "], 7, %[" #x "]\n\t"
#define I_AND_SGPR(x) "v_and_b32 %[" #x "], s20, %[" #x "]\n\t"
#define I_LSHL_SGPR(x) "v_lshlrev_b32 %[" #x "], s20, %[" #x "]\n\t"
#define I_CND_S(x) "v_cndmask_b32 %[" #x "], %[" #x "], %[b], s[22:23]\n\t"
#define I_CND_3(x) "v_cndmask_b32 %[" #x "], %[c], %[b], vcc\n\t"
#define I_CND_EXEC(x) "v_cndmask_b32 %[" #x "], %[" #x "], %[b], exec\n\t"
#define I_CND_1_3(x) "v_cndmask_b32 %[" #x "], %[" #x "], %[b], vcc\n\tv_add_u32 %[" #x "], %[" #x "], %[b]\n\tv_and_b32 %[" #x "], %[" #x "], %[c]\n\tv_add_u32 %[" #x "], %[" #x "], %[c]\n\t"
#define I_CND_E64_VCC(x) "v_cndmask_b32_e64 %[" #x "], %[" #x "], %[b], vcc\n\t"
#define I_CND_2_2(x) "v_cndmask_b32 %[" #x "], %[" #x "], %[b], vcc\n\tv_cndmask_b32 %[" #x "], %[" #x "], %[c], vcc\n\tv_add_u32 %[" #x "], %[" #x "], %[b]\n\tv_and_b32 %[" #x "], %[" #x "], %[c]\n\t"
#define I_CND_1_1(x) "v_cndmask_b32 %[" #x "], %[" #x "], %[b], vcc\n\tv_add_u32 %[" #x "], %[" #x "], %[b]\n\t"
#define I_CND_VS(x) "v_cndmask_b32 %[" #x "], %[" #x "], %[b], vcc\n\tv_cndmask_b32 %[" #x "], %[" #x "], %[c], s[22:23]\n\t"
#define I_CND_SNOP(x) "v_cndmask_b32 %[" #x "], %[" #x "], %[b], vcc\n\ts_nop 0\n\t"
#define I_CND_SALU(x) "v_cndmask_b32 %[" #x "], %[" #x "], %[b], vcc\n\ts_add_u32 s20, s20, 3\n\t"
#define I_ADDC_2(x) "v_addc_co_u32 %[" #x "], s[22:23], %[" #x "], %[b], s[22:23]\n\t"
#define I_LSHL2(x) "v_lshlrev_b32 %[" #x "], 2, %[" #x "]\n\t"
#define I_LSHL_E64(x) "v_lshlrev_b32_e64 %[" #x "], 1, %[" #x "]\n\t"
#define I_ASHR(x) "v_ashrrev_i32 %[" #x "], 1, %[" #x "]\n\t"
#define I_LSHRV(x) "v_lshrrev_b32 %[" #x "], %[c], %[" #x "]\n\t"
#define I_MUL_F32(x) "v_mul_f32 %[" #x "], %[" #x "], %[b]\n\t"
#define I_ADD_F32(x) "v_add_f32 %[" #x "], %[" #x "], %[b]\n\t"
#define I_NOT(x) "v_not_b32 %[" #x "], %[" #x "]\n\t"
#define I_BFREV(x) "v_bfrev_b32 %[" #x "], %[" #x "]\n\t"
#define I_FFBH(x) "v_ffbh_u32 %[" #x "], %[" #x "]\n\t"
#define I_SUBREV(x) "v_subrev_u32 %[" #x "], %[" #x "], %[b]\n\t"
#define I_SLOW_FAST(x) "v_lshlrev_b32 %[" #x "], 1, %[" #x "]\n\tv_add_u32 %[" #x "], %[" #x "], %[b]\n\t"
#define I_SLOW_2FAST(x) "v_perm_b32 %[" #x "], %[" #x "], %[b], %[c]\n\tv_add_u32 %[" #x "], %[" #x "], %[b]\n\tv_and_b32 %[" #x "], %[" #x "], %[c]\n\t"
#define I_SLOW_3FAST(x) "v_perm_b32 %[" #x "], %[" #x "], %[b], %[c]\n\tv_add_u32 %[" #x "], %[" #x "], %[b]\n\tv_and_b32 %[" #x "], %[" #x "], %[c]\n\tv_xor_b32 %[" #x "], %[" #x "], %[b]\n\t"
#define I_2SLOW_FAST(x) "v_perm_b32 %[" #x "], %[" #x "], %[b], %[c]\n\tv_lshlrev_b32 %[" #x "], 1, %[" #x "]\n\tv_add_u32 %[" #x "], %[" #x "], %[b]\n\t"
#define I_FAST_LDS(x) "v_add_u32 %[" #x "], %[" #x "], %[b]\n\tds_read_b32 v40, %[c]\n\tv_and_b32 %[" #x "], %[" #x "], %[b]\n\tv_xor_b32 %[" #x "], %[" #x "], %[c]\n\ts_waitcnt lgkmcnt(4)\n\t"
#define I_SLOW_SALU(x) "v_lshlrev_b32 %[" #x "], 1, %[" #x "]\n\ts_add_u32 s20, s20, 3\n\t"
#define I_CMP_CND(x) "v_cmp_lt_u32 vcc, %[" #x "], %[b]\n\tv_cndmask_b32 %[" #x "], %[" #x "], %[b], vcc\n\t"
#define I_ADD_SDWA(x) "v_add_u32_sdwa %[" #x "], %[" #x "], %[b] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t"
#define I_AND_SDWA(x) "v_and_b32_sdwa %[" #x "], %[" #x "], %[b] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n\t"
#define I_MOV_SDWA(x) "v_mov_b32_sdwa %[" #x "], %[b] dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0\n\t"
#define I_LSHL_SDWA(x) "v_lshlrev_b32_sdwa %[" #x "], %[c], %[" #x "] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
#define I_DSR32(x) "ds_read_b32 %[" #x "], %[c]\n\ts_waitcnt lgkmcnt(4)\n\t"
#define I_DSR8(x) "ds_read_u8 %[" #x "], %[c]\n\ts_waitcnt lgkmcnt(4)\n\t"
#define I_DSR128(x) "ds_read_b128 v[40:43], %[c]\n\ts_waitcnt lgkmcnt(4)\n\t"
#define I_DSW8(x) "ds_write_b8 %[c], %[" #x "]\n\ts_waitcnt lgkmcnt(4)\n\t"
#define I_DSW32(x) "ds_write_b32 %[c], %[" #x "]\n\ts_waitcnt lgkmcnt(4)\n\t"
#define I_SWAITCNT(x) "s_waitcnt lgkmcnt(0)\n\t"
#define I_SNOP(x) "s_nop 0\n\t"
#define I_SMOV64(x) "s_mov_b64 s[22:23], exec\n\t"
#define I_SAVEEXEC(x) "s_and_saveexec_b64 s[22:23], vcc\n\ts_mov_b64 exec, s[22:23]\n\t"
#define I_SCSEL(x) "s_cselect_b32 s20, s20, s21\n\t"
#define I_SLSHL(x) "s_lshl_b32 s20, s20, 1\n\t"
#define I_SFF1(x) "s_ff1_i32_b64 s20, s[22:23]\n\t"

#define KERNEL(NAME, INS)                                                                                          \
  __global__ void __launch_bounds__(256) k_##NAME(Rec* out, int iters, uint32_t* sink)                             \
  {                                                                                                                \
    extern __shared__ uint32_t lds[];                                                                              \
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6,       \
             a7 = a0 + 7;                                                                                          \
    uint32_t b = threadIdx.x * 3 + 1, c = (threadIdx.x & 3) * 4;                                                   \
    typedef float f2 __attribute__((ext_vector_type(2)));                                                          \
    f2 p = {1.0f, 0.5f};                                                                                           \
    f2 q0 = p, q1 = p, q2 = p, q3 = p, q4 = p, q5 = p, q6 = p, q7 = p;                                             \
    (void)q0; (void)q1; (void)q2; (void)q3; (void)q4; (void)q5; (void)q6; (void)q7;                                \
    if (threadIdx.x == 100000) lds[0] = 1;                                                                         \
    __syncthreads();                                                                                               \
    uint64_t t0, t1, r0, r1;                                                                                       \
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0));                 \
    PROLOGUE                                                                                                       \
    for (int it = 0; it < iters; ++it) {                                                                           \
      asm volatile(BLOCK64(INS)                                                                                    \
                   : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [a4] "+v"(a4), [a5] "+v"(a5),     \
                     [a6] "+v"(a6), [a7] "+v"(a7)                                                                  \
                   : [b] "v"(b), [c] "v"(c), [p] "v"(p)                                                            \
                   : "vcc", "s20", "s21", "s22", "s23", "scc", "v40", "v41", "v42", "v43", "memory");                                                    \
    }                                                                                                              \
    asm volatile("s_mov_b64 exec, -1");                                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)"               \
                 : "=s"(t1), "=s"(r1));                                                                            \
    uint32_t hw, xcc;                                                                                              \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc)); \
    if ((threadIdx.x & 63) == 0) {                                                                                 \
      Rec r = {t0, t1, r0, r1, hw, xcc, 0, 0};                                                                     \
      out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = r;                                                       \
    }                                                                                                              \
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345678u) sink[0] = a0;                                        \
  }

#define PROLOGUE
KERNEL(add, I_ADD)
KERNEL(and_, I_AND)
KERNEL(lshl, I_LSHL)
KERNEL(lshlv, I_LSHLV)
KERNEL(cndmask, I_CNDMASK)
KERNEL(alignbyte, I_ALIGNBYTE)
KERNEL(alignbit, I_ALIGNBIT)
KERNEL(perm, I_PERM)
KERNEL(dpp_row_shr, I_DPP)
KERNEL(dpp_wave_shr, I_DPPWS)
KERNEL(add_dpp, I_ADD_DPP)
KERNEL(readlane, I_READLANE)
KERNEL(readfirstlane, I_READFIRST)
KERNEL(add3, I_ADD3)
KERNEL(lshl_add, I_LSHLADD)
KERNEL(bfe, I_BFE)
KERNEL(mad_u32_u24, I_MAD24)
KERNEL(mul_lo_u32, I_MULLO)
KERNEL(pk_add_u16, I_PKADD16)
KERNEL(cmp_vcc, I_CMP)
KERNEL(cmp_sgpr, I_CMP_S)
KERNEL(min_u32, I_MIN)
KERNEL(fma_f32, I_FMA)
KERNEL(mbcnt, I_MBCNT)
KERNEL(ds_bpermute, I_BPERM)
KERNEL(s_add, I_SADD)
KERNEL(s_and_b64, I_SAND64)
KERNEL(s_bcnt1, I_SBCNT)
KERNEL(valu_salu_1_1, I_VS)
KERNEL(valu_salu_1_2, I_VSS)
KERNEL(dep_v_add, I_DEP)
KERNEL(dep_s_add, I_DEP_S)

KERNEL(or_, I_OR)
KERNEL(xor_, I_XOR)
KERNEL(sub, I_SUB)
KERNEL(mov, I_MOV)
KERNEL(lshr, I_LSHR)
KERNEL(max_u32, I_MAX)
KERNEL(and_or, I_ANDOR)
KERNEL(or3, I_OR3)
KERNEL(bfi, I_BFI)
KERNEL(add_co, I_ADDCO)
KERNEL(addc_co, I_ADDC)
KERNEL(mul_u32_u24, I_MUL24)
KERNEL(add_u16, I_ADD16)
KERNEL(add_e64, I_ADD_E64)
KERNEL(add_sgpr, I_ADD_SGPR)
KERNEL(add_lit, I_ADD_LIT)
KERNEL(add_inl, I_ADD_INL)
KERNEL(and_sgpr, I_AND_SGPR)
KERNEL(lshl_sgpr, I_LSHL_SGPR)
KERNEL(cnd_sgpr, I_CND_S)
KERNEL(cnd_3, I_CND_3)
KERNEL(cnd_exec, I_CND_EXEC)
KERNEL(cnd_1_3, I_CND_1_3)
KERNEL(cmp_cnd, I_CMP_CND)
KERNEL(slow_fast, I_SLOW_FAST)
KERNEL(slow_2fast, I_SLOW_2FAST)
KERNEL(slow_3fast, I_SLOW_3FAST)
KERNEL(slow2_fast, I_2SLOW_FAST)
KERNEL(fast_lds, I_FAST_LDS)
KERNEL(slow_salu, I_SLOW_SALU)
KERNEL(cnd_e64_vcc, I_CND_E64_VCC)
KERNEL(cnd_2_2, I_CND_2_2)
KERNEL(cnd_1_1, I_CND_1_1)
KERNEL(cnd_vs, I_CND_VS)
KERNEL(cnd_snop, I_CND_SNOP)
KERNEL(cnd_salu, I_CND_SALU)
KERNEL(addc_sgpr, I_ADDC_2)
KERNEL(lshl2, I_LSHL2)
KERNEL(lshl_e64, I_LSHL_E64)
KERNEL(ashr, I_ASHR)
KERNEL(lshrv, I_LSHRV)
KERNEL(mul_f32, I_MUL_F32)
KERNEL(add_f32, I_ADD_F32)
KERNEL(not_, I_NOT)
KERNEL(bfrev, I_BFREV)
KERNEL(ffbh, I_FFBH)
KERNEL(subrev, I_SUBREV)
KERNEL(add_sdwa, I_ADD_SDWA)
KERNEL(and_sdwa, I_AND_SDWA)
KERNEL(mov_sdwa, I_MOV_SDWA)
KERNEL(lshl_sdwa, I_LSHL_SDWA)
KERNEL(ds_read_b32, I_DSR32)
KERNEL(ds_read_u8, I_DSR8)
KERNEL(ds_read_b128, I_DSR128)
KERNEL(ds_write_b8, I_DSW8)
KERNEL(ds_write_b32, I_DSW32)
KERNEL(s_waitcnt, I_SWAITCNT)
KERNEL(s_nop, I_SNOP)
KERNEL(s_mov_b64, I_SMOV64)
KERNEL(saveexec, I_SAVEEXEC)
KERNEL(s_cselect, I_SCSEL)
KERNEL(s_lshl, I_SLSHL)
KERNEL(s_ff1, I_SFF1)
#undef PROLOGUE
#define PROLOGUE asm volatile("s_mov_b64 exec, 0xffffffff");
KERNEL(lshl_exec_lo32, I_LSHL)
KERNEL(add_exec_lo32, I_ADD)
#undef PROLOGUE
#define PROLOGUE asm volatile("s_mov_b64 exec, 1");
KERNEL(lshl_exec_1, I_LSHL)
#undef PROLOGUE
#define PROLOGUE asm volatile("s_mov_b32 exec_lo, 0\n\ts_mov_b32 exec_hi, 0xffff");
KERNEL(lshl_exec_hi16, I_LSHL)
#undef PROLOGUE
#define PROLOGUE asm volatile("s_mov_b32 exec_lo, 0x0000ffff\n\ts_mov_b32 exec_hi, 0x0000ffff");
KERNEL(lshl_exec_2x16, I_LSHL)
#undef PROLOGUE
#define PROLOGUE

struct K {
  const char* name;
  void (*fn)(Rec*, int, uint32_t*);
  int per_block; /* instructions of a BLOCK64 */
  const char* kind;
};


const K ks_b[] = {
      {"v_add_u32", k_add, 64, "valu"},
      {"v_or_b32", k_or_, 64, "valu"},
      {"v_xor_b32", k_xor_, 64, "valu"},
      {"v_sub_u32", k_sub, 64, "valu"},
      {"v_mov_b32", k_mov, 64, "valu"},
      {"v_lshrrev_b32 (imm)", k_lshr, 64, "valu"},
      {"v_max_u32", k_max_u32, 64, "valu"},
      {"v_and_or_b32", k_and_or, 64, "valu"},
      {"v_or3_b32", k_or3, 64, "valu"},
      {"v_bfi_b32", k_bfi, 64, "valu"},
      {"v_add_co_u32 (vcc out)", k_add_co, 64, "valu"},
      {"v_addc_co_u32 (vcc in, out)", k_addc_co, 64, "valu"},
      {"v_mul_u32_u24", k_mul_u32_u24, 64, "valu"},
      {"v_add_u16", k_add_u16, 64, "valu"},
      {"v_add_u32_e64 (VOP3 encoding)", k_add_e64, 64, "valu"},
      {"v_add_u32 sgpr source", k_add_sgpr, 64, "valu"},
      {"v_add_u32 literal source", k_add_lit, 64, "valu"},
      {"v_add_u32 inline constant", k_add_inl, 64, "valu"},
      {"v_and_b32 sgpr source", k_and_sgpr, 64, "valu"},
      {"v_lshlrev_b32 sgpr shift", k_lshl_sgpr, 64, "valu"},
      {"v_cndmask_b32 sgpr-pair mask (VOP3)", k_cnd_sgpr, 64, "valu"},
      {"v_cndmask_b32 vcc, three registers", k_cnd_3, 64, "valu"},
      {"v_cndmask_b32 exec as mask", k_cnd_exec, 64, "valu"},
      {"v_cndmask vcc + add + and + add (per 4)", k_cnd_1_3, 64, "quad"},
      {"v_cmp -> vcc + v_cndmask vcc (per pair)", k_cmp_cnd, 64, "pair"},
      {"v_add_u32_sdwa src0 BYTE_1", k_add_sdwa, 64, "valu"},
      {"v_and_b32_sdwa src0 BYTE_2", k_and_sdwa, 64, "valu"},
      {"v_mov_b32_sdwa dst BYTE_1 preserve", k_mov_sdwa, 64, "valu"},
      {"v_lshlrev_b32_sdwa", k_lshl_sdwa, 64, "valu"},
      {"ds_read_b32 (one address per lane, stride 16 B)", k_ds_read_b32, 64, "lds"},
      {"ds_read_u8", k_ds_read_u8, 64, "lds"},
      {"ds_read_b128", k_ds_read_b128, 64, "lds"},
      {"ds_write_b8", k_ds_write_b8, 64, "lds"},
      {"ds_write_b32", k_ds_write_b32, 64, "lds"},
      {"s_waitcnt lgkmcnt(0) (nothing pending)", k_s_waitcnt, 64, "salu"},
      {"s_nop 0", k_s_nop, 64, "salu"},
      {"s_mov_b64 from exec", k_s_mov_b64, 64, "salu"},
      {"s_and_saveexec_b64 + s_mov exec (per pair)", k_saveexec, 64, "pair"},
      {"s_cselect_b32", k_s_cselect, 64, "salu"},
      {"s_lshl_b32", k_s_lshl, 64, "salu"},
      {"s_ff1_i32_b64", k_s_ff1, 64, "salu"},
      {"v_lshlrev_b32, exec = low 32 lanes", k_lshl_exec_lo32, 64, "valu"},
      {"v_add_u32, exec = low 32 lanes", k_add_exec_lo32, 64, "valu"},
      {"v_lshlrev_b32, exec = 1 lane", k_lshl_exec_1, 64, "valu"},
      {"v_lshlrev_b32, exec = lanes 32-47", k_lshl_exec_hi16, 64, "valu"},
      {"v_lshlrev_b32, exec = lanes 0-15 and 32-47", k_lshl_exec_2x16, 64, "valu"},
};

const K ks_d[] = { /* the pipes: do 4-cycle and 2-cycle operations of different waves overlap? + PMC calibration set */
      {"v_add_u32", k_add, 64, "valu"},
      {"v_lshlrev_b32 (imm)", k_lshl, 64, "valu"},
      {"v_lshlrev + v_add (per pair)", k_slow_fast, 64, "pair"},
      {"v_perm + v_add + v_and (per triple)", k_slow_2fast, 64, "triple"},
      {"v_perm + v_add + v_and + v_xor (per quad)", k_slow_3fast, 64, "quad"},
      {"v_perm + v_lshlrev + v_add (per triple)", k_slow2_fast, 64, "triple"},
      {"v_add + ds_read_b32 + v_and + v_xor (per quad)", k_fast_lds, 64, "quad"},
      {"v_lshlrev + s_add_u32 (per pair)", k_slow_salu, 64, "pair"},
      {"s_add_u32", k_s_add, 64, "salu"},
};
const K ks_c[] = {
      {"v_cndmask_b32 vcc (VOP2), back to back", k_cndmask, 64, "valu"},
      {"v_cndmask_b32_e64 vcc (VOP3), back to back", k_cnd_e64_vcc, 64, "valu"},
      {"2 x v_cndmask vcc + add + and (per 4)", k_cnd_2_2, 64, "quad"},
      {"v_cndmask vcc + v_add_u32 (per pair)", k_cnd_1_1, 64, "pair"},
      {"v_cndmask vcc + v_cndmask sgpr pair (per pair)", k_cnd_vs, 64, "pair"},
      {"v_cndmask vcc + s_nop 0 (per pair)", k_cnd_snop, 64, "pair"},
      {"v_cndmask vcc + s_add_u32 (per pair)", k_cnd_salu, 64, "pair"},
      {"v_addc_co_u32 sgpr pair in, out", k_addc_sgpr, 64, "valu"},
      {"v_lshlrev_b32 (imm 2)", k_lshl2, 64, "valu"},
      {"v_lshlrev_b32_e64 (imm 1)", k_lshl_e64, 64, "valu"},
      {"v_ashrrev_i32 (imm)", k_ashr, 64, "valu"},
      {"v_lshrrev_b32 (vgpr)", k_lshrv, 64, "valu"},
      {"v_mul_f32", k_mul_f32, 64, "valu"},
      {"v_add_f32", k_add_f32, 64, "valu"},
      {"v_not_b32", k_not_, 64, "valu"},
      {"v_bfrev_b32", k_bfrev, 64, "valu"},
      {"v_ffbh_u32", k_ffbh, 64, "valu"},
      {"v_subrev_u32", k_subrev, 64, "valu"},
};

int main(int argc, char** argv)
{
  const int iters = argc > 1 ? atoi(argv[1]) : 2048;
  const bool second = argc > 2 && std::string(argv[2]) == "b"; /* the second table */
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const K ks_a[] = {
      {"v_add_u32", k_add, 64, "valu"},
      {"v_and_b32", k_and_, 64, "valu"},
      {"v_lshlrev_b32 (imm)", k_lshl, 64, "valu"},
      {"v_lshlrev_b32 (vgpr)", k_lshlv, 64, "valu"},
      {"v_cndmask_b32", k_cndmask, 64, "valu"},
      {"v_alignbyte_b32", k_alignbyte, 64, "valu"},
      {"v_alignbit_b32", k_alignbit, 64, "valu"},
      {"v_perm_b32", k_perm, 64, "valu"},
      {"v_mov_b32 dpp row_shr:1", k_dpp_row_shr, 64, "valu"},
      {"v_mov_b32 dpp wave_shr:1", k_dpp_wave_shr, 64, "valu"},
      {"v_add_u32 dpp row_shr:1", k_add_dpp, 64, "valu"},
      {"v_readlane_b32", k_readlane, 64, "valu"},
      {"v_readfirstlane_b32", k_readfirstlane, 64, "valu"},
      {"v_add3_u32", k_add3, 64, "valu"},
      {"v_lshl_add_u32", k_lshl_add, 64, "valu"},
      {"v_bfe_u32", k_bfe, 64, "valu"},
      {"v_mad_u32_u24", k_mad_u32_u24, 64, "valu"},
      {"v_mul_lo_u32", k_mul_lo_u32, 64, "valu"},
      {"v_pk_add_u16", k_pk_add_u16, 64, "valu"},
      {"v_cmp_lt_u32 -> vcc", k_cmp_vcc, 64, "valu"},
      {"v_cmp_lt_u32 -> sgpr pair", k_cmp_sgpr, 64, "valu"},
      {"v_min_u32", k_min_u32, 64, "valu"},
      {"v_fma_f32", k_fma_f32, 64, "valu"},
      {"v_mbcnt_lo_u32_b32", k_mbcnt, 64, "valu"},
      {"ds_bpermute_b32", k_ds_bpermute, 64, "lds"},
      {"s_add_u32", k_s_add, 64, "salu"},
      {"s_and_b64", k_s_and_b64, 64, "salu"},
      {"s_bcnt1_i32_b64", k_s_bcnt1, 64, "salu"},
      {"v_add_u32 + s_add_u32 alternating (per pair)", k_valu_salu_1_1, 64, "pair"},
      {"v_add_u32 + 2 salu alternating (per triple)", k_valu_salu_1_2, 64, "triple"},
      {"v_add_u32 dependent chain", k_dep_v_add, 64, "valu-dep"},
      {"s_add_u32 dependent chain", k_dep_s_add, 64, "salu-dep"},
  };
  const int only_wps = argc > 3 ? atoi(argv[3]) : 0; /* PMC passes: one occupancy, one dispatch pair per op */
  const int wps_all[] = {1, 2, 4, 7, 8};
  std::vector<int> wps_list(wps_all, wps_all + 5);
  if (only_wps) wps_list.assign(1, only_wps);
  const int max_waves = cus * 4 * 8;
  Rec* d_out;
  uint32_t* d_sink;
  CK(hipMalloc(&d_out, sizeof(Rec) * max_waves));
  CK(hipMalloc(&d_sink, 64));
  std::vector<Rec> h(max_waves);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_rate_khz\": %d, \"iters\": %d, \"instructions_per_wave\": %lld}\n",
         prop.gcnArchName, cus, prop.clockRate, iters, (long long)iters * 64);
  const bool third = argc > 2 && std::string(argv[2]) == "c";
  const bool fourth = argc > 2 && std::string(argv[2]) == "d";
  const K* ks = fourth ? ks_d : third ? ks_c : second ? ks_b : ks_a;
  const size_t nk = fourth ? sizeof(ks_d) / sizeof(K) : third ? sizeof(ks_c) / sizeof(K) : second ? sizeof(ks_b) / sizeof(K) : sizeof(ks_a) / sizeof(K);
  for (size_t ki = 0; ki < nk; ++ki) {
    const K& k = ks[ki];
    for (int wps : wps_list) {
      /* a workgroup is 4 waves (one per SIMD); LDS so that exactly `wps` workgroups fit a CU */
      const int lds = (160 * 1024 / wps) & ~255;
      CK(hipFuncSetAttribute((const void*)k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      const int blocks = cus * wps, waves = blocks * 4;
      for (int rep = 0; rep < 2; ++rep) { /* the first run warms the clock */
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), lds, 0, d_out, iters, d_sink);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
      }
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipMemcpy(h.data(), d_out, sizeof(Rec) * waves, hipMemcpyDeviceToHost));
      /* group by SIMD: xcc, se, sh, cu, simd */
      struct G {
        uint64_t lo = ~0ull, hi = 0, n = 0;
      };
      std::map<uint64_t, G> simds;
      std::vector<double> per_wave, ghz;
      for (int w = 0; w < waves; ++w) {
        const Rec& r = h[w];
        const uint64_t key = ((uint64_t)(r.xcc_id & 0xf) << 32) | (r.hw_id & 0xff30u);
        G& g = simds[key];
        g.lo = std::min(g.lo, r.t0), g.hi = std::max(g.hi, r.t1), g.n += 1;
        per_wave.push_back(double(r.t1 - r.t0) / (double(iters) * k.per_block));
        if (r.r1 > r.r0) ghz.push_back(double(r.t1 - r.t0) / double(r.r1 - r.r0) * 0.1);
      }
      std::vector<double> per_simd, occ;
      for (auto& kv : simds) {
        per_simd.push_back(double(kv.second.hi - kv.second.lo) / (double(kv.second.n) * iters * k.per_block));
        occ.push_back(double(kv.second.n));
      }
      auto med = [](std::vector<double>& v) {
        if (v.empty()) return 0.0;
        std::sort(v.begin(), v.end());
        return v[v.size() / 2];
      };
      const double total_instr = double(waves) * iters * k.per_block;
      const double mghz = med(ghz);
      std::sort(occ.begin(), occ.end());
      printf("{\"op\": \"%s\", \"kind\": \"%s\", \"waves_per_simd_launched\": %d, \"simds_seen\": %zu, "
             "\"waves_per_simd_seen_min_med_max\": [%.0f, %.0f, %.0f], "
             "\"cycles_per_instr_per_simd_median\": %.3f, \"cycles_per_instr_one_wave_median\": %.3f, "
             "\"memtime_ghz_median\": %.3f, \"wall_ms\": %.4f, "
             "\"cycles_per_instr_per_simd_from_wall\": %.3f}\n",
             k.name, k.kind, wps, simds.size(), occ.front(), occ[occ.size() / 2], occ.back(), med(per_simd), med(per_wave),
             mghz, ms, ms * 1e-3 * mghz * 1e9 * (cus * 4) / total_instr);
      fflush(stdout);
    }
  }
  return 0;
}
