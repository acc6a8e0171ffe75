// valu_calib.hip -- VALU issue calibration for gfx950 (MI355X): how many shader cycles one SIMD needs per wave64
// VALU instruction, alone and with several waves resident, for the instruction kinds the traversal kernel is made of.
//
//   v_fma_f32            plain fp32 FMA
//   v_pk_fma_f32         packed fp32 FMA (two FMAs per lane; the slab arithmetic of the BVH8 node test)
//   v_cvt_f32_ubyte0     byte -> float conversion (the quantised box planes)
//   v_max3_f32 / v_min_f32 / v_and_b32 / v_cndmask_b32 / v_lshlrev_b32   the rest of the node test's mix
//
// Each wave runs ITERS iterations of 32 instructions on 8 independent accumulators (dependency distance 8) between
// two s_memtime stamps.  Workgroups of 256 threads (one wave per SIMD of a CU) with enough dynamic LDS that exactly
// W of them fit on a CU give W waves per SIMD on every SIMD of the chip: cycles per instruction per SIMD =
// median(stamp delta) / (W x instructions per wave).  Under `rocprofv3 --pmc` the same launches give the counter
// value per instruction (SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE).
//
// Build: hipcc --offload-arch=gfx950 -O2 tools/calib/valu_calib.hip -o tools/calib/valu_calib
// Run:   tools/calib/valu_calib > gpurun_out/valu_calib.json
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kIters = 2048;
constexpr int kPerIter = 32;

struct Stamp { unsigned long long cycles, realtime; };

#define REP8(OP) OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#define REP32(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP)

#define KERNEL(NAME, DECL, BODY, SINK)                                                                  \
    __global__ void __launch_bounds__(256) NAME(Stamp *out, float *sink, float x, float y)              \
    {                                                                                                   \
        extern __shared__ unsigned char lds_pad[];                                                      \
        DECL                                                                                            \
        unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();    \
        for (int it = 0; it < kIters; ++it) { BODY }                                                    \
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();    \
        if ((threadIdx.x & 63) == 0) {                                                                  \
            Stamp s; s.cycles = t1 - t0; s.realtime = r1 - r0;                                          \
            out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = s;                                      \
        }                                                                                               \
        SINK                                                                                            \
        if (x == 12345.678f) lds_pad[threadIdx.x] = 1;                                                  \
    }

#define DECL_F float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
#define SINK_F if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0.123f) sink[threadIdx.x] = a0;
typedef float pk2 __attribute__((ext_vector_type(2)));
#define DECL_P unsigned ux = __float_as_uint(x); pk2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f; pk2 px = {x, x}, py = {y, y};
#define SINK_P { pk2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; if (s.x + s.y == 0.123f) sink[threadIdx.x] = s.x; }
#define DECL_U unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; unsigned ux = __float_as_uint(x), uy = __float_as_uint(y); unsigned long long smask = __ballot(threadIdx.x & 1);
#define SINK_U if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) sink[threadIdx.x] = (float)a0;

#define OP_FMA(a) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
#define OP_PKFMA(a) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(px), "v"(py));
#define OP_CVT(a) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(a));
#define OP_MAX3(a) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
#define OP_MIN(a) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a) : "v"(x));
#define OP_AND(a) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(ux));
#define OP_LSHL(a) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a));
#define OP_CNDMASK(a) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a) : "v"(ux), "s"(smask));
#define OP_ADDF(a) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(x));
#define OP_MULF(a) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(x));
#define OP_MAXF(a) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(x));
#define OP_MIN3(a) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
#define OP_MED3(a) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
#define OP_FMAC(a) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
#define OP_FMAMIX_LO(a) asm volatile("v_fma_mix_f32 %0, %1, %0, %2 op_sel_hi:[1,0,0]" : "+v"(a) : "v"(x), "v"(y));
#define OP_FMAMIX_HI(a) asm volatile("v_fma_mix_f32 %0, %1, %0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a) : "v"(x), "v"(y));
#define OP_PKMUL(a) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a) : "v"(px));
#define OP_PKADD(a) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(px));
#define OP_OR(a) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a) : "v"(ux));
#define OP_XOR(a) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(ux));
#define OP_ADDU(a) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(ux));
#define OP_SUBU(a) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a) : "v"(ux));
#define OP_LSHR(a) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a));
#define OP_ANDOR(a) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a) : "v"(ux), "v"(uy));
#define OP_LSHLOR(a) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a) : "v"(ux));
#define OP_OR3(a) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a) : "v"(ux), "v"(uy));
#define OP_PERM(a) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a) : "v"(ux), "v"(uy));
#define OP_CVTU(a) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a));
#define OP_CVTUB2(a) asm volatile("v_cvt_f32_ubyte2 %0, %0" : "+v"(a));
#define OP_MOV(a) asm volatile("v_mov_b32 %0, %1" : "+v"(a) : "v"(ux));
#define OP_CMP64(a) asm volatile("v_cmp_le_f32_e64 s[20:21], %0, %1" : : "v"(a), "v"(x) : "s20", "s21");
#define OP_MAD24(a) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a) : "v"(ux), "v"(uy));
#define OP_BCNT(a) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a) : "v"(ux));
#define OP_FFBH(a) asm volatile("v_ffbh_u32 %0, %0" : "+v"(a));
#define OP_ADD3(a) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a) : "v"(ux), "v"(uy));
#define OP_LSHLADD(a) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a) : "v"(ux));
#define OP_DPP(a) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a));
#define OP_ADDDPP(a) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a));
#define OP_BPERM(a) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a) : "v"(ux));
#define OP_READLANE(a) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(a) : "s20");
#define OP_SDWA(a) asm volatile("v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "+v"(a));
#define OP_FMA_SGPR(a) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "s"(x), "v"(y));
#define OP_CVTPKFP8(a) asm volatile("v_cvt_pk_f32_fp8 %0, %1" : "+v"(a) : "v"(ux));
#define OP_BFE(a) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(a));
#define OP_MULLO(a) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(ux));
#define OP_RCP(a) asm volatile("v_rcp_f32 %0, %0" : "+v"(a));
#define OP_CMP(a) asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(a), "v"(x) : "vcc");

KERNEL(calib_v_fma_f32, DECL_F, REP32(OP_FMA), SINK_F)
KERNEL(calib_v_pk_fma_f32, DECL_P, REP32(OP_PKFMA), SINK_P)
KERNEL(calib_v_cvt_f32_ubyte0, DECL_U, REP32(OP_CVT), SINK_U)
KERNEL(calib_v_max3_f32, DECL_F, REP32(OP_MAX3), SINK_F)
KERNEL(calib_v_min_f32, DECL_F, REP32(OP_MIN), SINK_F)
KERNEL(calib_v_and_b32, DECL_U, REP32(OP_AND), SINK_U)
KERNEL(calib_v_lshlrev_b32, DECL_U, REP32(OP_LSHL), SINK_U)
KERNEL(calib_v_cndmask_b32, DECL_U, REP32(OP_CNDMASK), SINK_U)
KERNEL(calib_v_bfe_u32, DECL_U, REP32(OP_BFE), SINK_U)
KERNEL(calib_v_mul_lo_u32, DECL_U, REP32(OP_MULLO), SINK_U)
KERNEL(calib_v_rcp_f32, DECL_F, REP32(OP_RCP), SINK_F)
KERNEL(calib_v_cmp_le_f32, DECL_F, REP32(OP_CMP), SINK_F)
KERNEL(calib_v_add_f32, DECL_F, REP32(OP_ADDF), SINK_F)
KERNEL(calib_v_mul_f32, DECL_F, REP32(OP_MULF), SINK_F)
KERNEL(calib_v_max_f32, DECL_F, REP32(OP_MAXF), SINK_F)
KERNEL(calib_v_min3_f32, DECL_F, REP32(OP_MIN3), SINK_F)
KERNEL(calib_v_med3_f32, DECL_F, REP32(OP_MED3), SINK_F)
KERNEL(calib_v_fmac_f32, DECL_F, REP32(OP_FMAC), SINK_F)
KERNEL(calib_v_fma_mix_f32_lo, DECL_F, REP32(OP_FMAMIX_LO), SINK_F)
KERNEL(calib_v_fma_mix_f32_hi, DECL_F, REP32(OP_FMAMIX_HI), SINK_F)
KERNEL(calib_v_fma_f32_sgpr, DECL_F, REP32(OP_FMA_SGPR), SINK_F)
KERNEL(calib_v_pk_mul_f32, DECL_P, REP32(OP_PKMUL), SINK_P)
KERNEL(calib_v_pk_add_f32, DECL_P, REP32(OP_PKADD), SINK_P)
KERNEL(calib_v_or_b32, DECL_U, REP32(OP_OR), SINK_U)
KERNEL(calib_v_xor_b32, DECL_U, REP32(OP_XOR), SINK_U)
KERNEL(calib_v_add_u32, DECL_U, REP32(OP_ADDU), SINK_U)
KERNEL(calib_v_sub_u32, DECL_U, REP32(OP_SUBU), SINK_U)
KERNEL(calib_v_lshrrev_b32, DECL_U, REP32(OP_LSHR), SINK_U)
KERNEL(calib_v_and_or_b32, DECL_U, REP32(OP_ANDOR), SINK_U)
KERNEL(calib_v_lshl_or_b32, DECL_U, REP32(OP_LSHLOR), SINK_U)
KERNEL(calib_v_or3_b32, DECL_U, REP32(OP_OR3), SINK_U)
KERNEL(calib_v_perm_b32, DECL_U, REP32(OP_PERM), SINK_U)
KERNEL(calib_v_cvt_f32_u32, DECL_U, REP32(OP_CVTU), SINK_U)
KERNEL(calib_v_cvt_f32_ubyte2, DECL_U, REP32(OP_CVTUB2), SINK_U)
KERNEL(calib_v_mov_b32, DECL_U, REP32(OP_MOV), SINK_U)
KERNEL(calib_v_cmp_le_f32_e64, DECL_F, REP32(OP_CMP64), SINK_F)
KERNEL(calib_v_mad_u32_u24, DECL_U, REP32(OP_MAD24), SINK_U)
KERNEL(calib_v_bcnt_u32_b32, DECL_U, REP32(OP_BCNT), SINK_U)
KERNEL(calib_v_ffbh_u32, DECL_U, REP32(OP_FFBH), SINK_U)
KERNEL(calib_v_add3_u32, DECL_U, REP32(OP_ADD3), SINK_U)
KERNEL(calib_v_lshl_add_u32, DECL_U, REP32(OP_LSHLADD), SINK_U)
KERNEL(calib_v_mov_b32_dpp, DECL_U, REP32(OP_DPP), SINK_U)
KERNEL(calib_v_add_u32_dpp, DECL_U, REP32(OP_ADDDPP), SINK_U)
KERNEL(calib_ds_bpermute_b32, DECL_U, REP32(OP_BPERM), SINK_U)
KERNEL(calib_v_readlane_b32, DECL_U, REP32(OP_READLANE), SINK_U)
KERNEL(calib_v_cvt_f32_u32_sdwa, DECL_U, REP32(OP_SDWA), SINK_U)
KERNEL(calib_v_cvt_pk_f32_fp8, DECL_P, REP32(OP_CVTPKFP8), SINK_P)

// mixes of a full-rate and a half-rate instruction, alternating: do the two classes share one pipe (costs add: 6 cycles
// per pair) or run side by side from different waves (4 per pair)?
#define MIX8(OPA, OPB) OPA(a0) OPB(a1) OPA(a2) OPB(a3) OPA(a4) OPB(a5) OPA(a6) OPB(a7)
#define MIX32(OPA, OPB) MIX8(OPA, OPB) MIX8(OPA, OPB) MIX8(OPA, OPB) MIX8(OPA, OPB)
#define DECL_M float a0 = threadIdx.x, a2 = a0 + 2, a4 = a0 + 4, a6 = a0 + 6; unsigned a1 = threadIdx.x + 1, a3 = a1 + 2, a5 = a1 + 4, a7 = a1 + 6; unsigned ux = __float_as_uint(x), uy = __float_as_uint(y);
#define SINK_M if (a0 + a2 + a4 + a6 == 0.123f || (a1 ^ a3 ^ a5 ^ a7) == 0x12345u) sink[threadIdx.x] = a0;
KERNEL(calib_mix_fma_cvt, DECL_M, MIX32(OP_FMA, OP_CVT), SINK_M)
KERNEL(calib_mix_fma_bfe, DECL_M, MIX32(OP_FMA, OP_BFE), SINK_M)
KERNEL(calib_mix_fma_and, DECL_M, MIX32(OP_FMA, OP_AND), SINK_M)
#define DECL_MF float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
KERNEL(calib_mix_fma_max3, DECL_MF, MIX32(OP_FMA, OP_MAX3), SINK_F)
KERNEL(calib_mix_add_min, DECL_MF, MIX32(OP_ADDF, OP_MIN), SINK_F)

KERNEL(calib_mix_cvt_bfe, DECL_U, MIX32(OP_CVT, OP_BFE), SINK_U)
#define MIX8_31(OPA, OPB) OPA(a0) OPA(a1) OPA(a2) OPB(a3) OPA(a4) OPA(a5) OPA(a6) OPB(a7)
#define DECL_M31 float a3 = threadIdx.x, a7 = a3 + 4; unsigned a0 = threadIdx.x + 1, a1 = a0 + 1, a2 = a0 + 2, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6; unsigned ux = __float_as_uint(x), uy = __float_as_uint(y);
#define SINK_M31 if (a3 + a7 == 0.123f || (a0 ^ a1 ^ a2 ^ a4 ^ a5 ^ a6) == 0x12345u) sink[threadIdx.x] = a3;
KERNEL(calib_mix_3cvt_1fma, DECL_M31, MIX8_31(OP_CVT, OP_FMA) MIX8_31(OP_CVT, OP_FMA) MIX8_31(OP_CVT, OP_FMA) MIX8_31(OP_CVT, OP_FMA), SINK_M31)
KERNEL(calib_mix_3and_1fma, DECL_M31, MIX8_31(OP_AND, OP_FMA) MIX8_31(OP_AND, OP_FMA) MIX8_31(OP_AND, OP_FMA) MIX8_31(OP_AND, OP_FMA), SINK_M31)
#define OP_ASHR(a) asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(a));
#define OP_LSHLV(a) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a) : "v"(ux));
#define OP_LSHRV(a) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a) : "v"(ux));
#define OP_SUBF(a) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a) : "v"(x));
#define OP_NOT(a) asm volatile("v_not_b32 %0, %0" : "+v"(a));
#define OP_BFI(a) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(a) : "v"(ux), "v"(uy));
#define OP_BITOP3(a) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x30" : "+v"(a) : "v"(ux), "v"(uy));
#define OP_MINU(a) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a) : "v"(ux));
#define OP_MULU24(a) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a) : "v"(ux));
#define OP_ADDLIT(a) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(a));
#define OP_ANDLIT(a) asm volatile("v_and_b32 %0, 0x7f8000, %0" : "+v"(a));
#define OP_ORLIT(a) asm volatile("v_or_b32 %0, 0x43800000, %0" : "+v"(a));
#define OP_FMA_LIT(a) asm volatile("v_fma_f32 %0, %0, %1, 0x3f000000" : "+v"(a) : "v"(x));
#define OP_ADDF_SGPR(a) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a) : "s"(x));
#define OP_ADDCO(a) asm volatile("v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(a) : : "vcc");
#define OP_FFBL(a) asm volatile("v_ffbl_b32 %0, %0" : "+v"(a));
#define OP_DIVSCALE(a) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a) : "v"(x) : "vcc");
#define OP_DIVFMAS(a) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y) : "vcc");
#define OP_DIVFIXUP(a) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
#define OP_SQRT(a) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a));
#define OP_MULF64LIKE(a) asm volatile("v_mul_legacy_f32 %0, %0, %1" : "+v"(a) : "v"(x));
KERNEL(calib_v_ashrrev_i32, DECL_U, REP32(OP_ASHR), SINK_U)
KERNEL(calib_v_lshlrev_b32_vgpr, DECL_U, REP32(OP_LSHLV), SINK_U)
KERNEL(calib_v_lshrrev_b32_vgpr, DECL_U, REP32(OP_LSHRV), SINK_U)
KERNEL(calib_v_sub_f32, DECL_F, REP32(OP_SUBF), SINK_F)
KERNEL(calib_v_not_b32, DECL_U, REP32(OP_NOT), SINK_U)
KERNEL(calib_v_bfi_b32, DECL_U, REP32(OP_BFI), SINK_U)
KERNEL(calib_v_bitop3_b32, DECL_U, REP32(OP_BITOP3), SINK_U)
KERNEL(calib_v_min_u32, DECL_U, REP32(OP_MINU), SINK_U)
KERNEL(calib_v_mul_u32_u24, DECL_U, REP32(OP_MULU24), SINK_U)
KERNEL(calib_v_add_u32_literal, DECL_U, REP32(OP_ADDLIT), SINK_U)
KERNEL(calib_v_and_b32_literal, DECL_U, REP32(OP_ANDLIT), SINK_U)
KERNEL(calib_v_or_b32_literal, DECL_U, REP32(OP_ORLIT), SINK_U)
KERNEL(calib_v_fma_f32_literal, DECL_F, REP32(OP_FMA_LIT), SINK_F)
KERNEL(calib_v_add_f32_sgpr, DECL_F, REP32(OP_ADDF_SGPR), SINK_F)
KERNEL(calib_v_addc_co_u32, DECL_U, REP32(OP_ADDCO), SINK_U)
KERNEL(calib_v_ffbl_b32, DECL_U, REP32(OP_FFBL), SINK_U)
KERNEL(calib_v_div_scale_f32, DECL_F, REP32(OP_DIVSCALE), SINK_F)
KERNEL(calib_v_div_fmas_f32, DECL_F, REP32(OP_DIVFMAS), SINK_F)
KERNEL(calib_v_div_fixup_f32, DECL_F, REP32(OP_DIVFIXUP), SINK_F)
KERNEL(calib_v_sqrt_f32, DECL_F, REP32(OP_SQRT), SINK_F)

typedef void (*kern_t)(Stamp *, float *, float, float);
struct Entry { const char *name; kern_t fn; };

int main(int argc, char **argv)
{
    const Entry kernels[] = {
        {"v_fma_f32", calib_v_fma_f32}, {"v_pk_fma_f32", calib_v_pk_fma_f32}, {"v_cvt_f32_ubyte0", calib_v_cvt_f32_ubyte0},
        {"v_max3_f32", calib_v_max3_f32}, {"v_min_f32", calib_v_min_f32}, {"v_and_b32", calib_v_and_b32},
        {"v_lshlrev_b32", calib_v_lshlrev_b32}, {"v_cndmask_b32", calib_v_cndmask_b32}, {"v_bfe_u32", calib_v_bfe_u32},
        {"v_mul_lo_u32", calib_v_mul_lo_u32}, {"v_rcp_f32", calib_v_rcp_f32}, {"v_cmp_le_f32", calib_v_cmp_le_f32},
        {"v_add_f32", calib_v_add_f32}, {"v_mul_f32", calib_v_mul_f32}, {"v_max_f32", calib_v_max_f32}, {"v_min3_f32", calib_v_min3_f32},
        {"v_med3_f32", calib_v_med3_f32}, {"v_fmac_f32", calib_v_fmac_f32}, {"v_fma_mix_f32_lo", calib_v_fma_mix_f32_lo},
        {"v_fma_mix_f32_hi", calib_v_fma_mix_f32_hi}, {"v_fma_f32_sgpr", calib_v_fma_f32_sgpr}, {"v_pk_mul_f32", calib_v_pk_mul_f32}, {"v_pk_add_f32", calib_v_pk_add_f32},
        {"v_or_b32", calib_v_or_b32}, {"v_xor_b32", calib_v_xor_b32}, {"v_add_u32", calib_v_add_u32}, {"v_sub_u32", calib_v_sub_u32},
        {"v_lshrrev_b32", calib_v_lshrrev_b32}, {"v_and_or_b32", calib_v_and_or_b32}, {"v_lshl_or_b32", calib_v_lshl_or_b32},
        {"v_or3_b32", calib_v_or3_b32}, {"v_perm_b32", calib_v_perm_b32}, {"v_cvt_f32_u32", calib_v_cvt_f32_u32},
        {"v_cvt_f32_ubyte2", calib_v_cvt_f32_ubyte2}, {"v_mov_b32", calib_v_mov_b32}, {"v_cmp_le_f32_e64", calib_v_cmp_le_f32_e64},
        {"v_mad_u32_u24", calib_v_mad_u32_u24}, {"v_bcnt_u32_b32", calib_v_bcnt_u32_b32}, {"v_ffbh_u32", calib_v_ffbh_u32},
        {"v_add3_u32", calib_v_add3_u32}, {"v_lshl_add_u32", calib_v_lshl_add_u32}, {"v_mov_b32_dpp", calib_v_mov_b32_dpp},
        {"v_add_u32_dpp", calib_v_add_u32_dpp}, {"ds_bpermute_b32", calib_ds_bpermute_b32}, {"v_readlane_b32", calib_v_readlane_b32},
        {"v_cvt_f32_u32_sdwa", calib_v_cvt_f32_u32_sdwa}, {"v_cvt_pk_f32_fp8", calib_v_cvt_pk_f32_fp8},
        {"mix_fma_cvt", calib_mix_fma_cvt}, {"mix_fma_bfe", calib_mix_fma_bfe}, {"mix_fma_and", calib_mix_fma_and},
        {"mix_fma_max3", calib_mix_fma_max3}, {"mix_add_min", calib_mix_add_min},
        {"mix_cvt_bfe", calib_mix_cvt_bfe}, {"mix_3cvt_1fma", calib_mix_3cvt_1fma}, {"mix_3and_1fma", calib_mix_3and_1fma},
        {"v_ashrrev_i32", calib_v_ashrrev_i32}, {"v_lshlrev_b32_vgpr", calib_v_lshlrev_b32_vgpr}, {"v_lshrrev_b32_vgpr", calib_v_lshrrev_b32_vgpr},
        {"v_sub_f32", calib_v_sub_f32}, {"v_not_b32", calib_v_not_b32}, {"v_bfi_b32", calib_v_bfi_b32}, {"v_bitop3_b32", calib_v_bitop3_b32},
        {"v_min_u32", calib_v_min_u32}, {"v_mul_u32_u24", calib_v_mul_u32_u24}, {"v_add_u32_literal", calib_v_add_u32_literal},
        {"v_and_b32_literal", calib_v_and_b32_literal}, {"v_or_b32_literal", calib_v_or_b32_literal}, {"v_fma_f32_literal", calib_v_fma_f32_literal},
        {"v_add_f32_sgpr", calib_v_add_f32_sgpr}, {"v_addc_co_u32", calib_v_addc_co_u32}, {"v_ffbl_b32", calib_v_ffbl_b32},
        {"v_div_scale_f32", calib_v_div_scale_f32}, {"v_div_fmas_f32", calib_v_div_fmas_f32}, {"v_div_fixup_f32", calib_v_div_fixup_f32},
        {"v_sqrt_f32", calib_v_sqrt_f32},
    };
    const int waves_list[] = {1, 2, 4, 8};
    int only_waves = argc > 1 ? atoi(argv[1]) : 0;      // 0: all
    const char *only_prefix = argc > 2 ? argv[2] : nullptr;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t lds_total = 160 * 1024;
    Stamp *d_out; float *d_sink;
    CHECK(hipMalloc((void **)&d_out, sizeof(Stamp) * cus * 8 * 4));
    CHECK(hipMalloc((void **)&d_sink, 4096));
    std::vector<Stamp> h(cus * 8 * 4);
    printf("{\"device\": \"%s\", \"cus\": %d, \"iters\": %d, \"instr_per_iter\": %d, \"results\": [\n", prop.gcnArchName, cus, kIters, kPerIter);
    bool first = true;
    for (const Entry &k : kernels)
        for (int w : waves_list) {
            if (only_waves && w != only_waves) continue;
            if (only_prefix && !strstr(k.name, only_prefix)) continue;
            // W workgroups of 4 waves per CU: dynamic LDS sized so that W fit and W + 1 do not
            size_t lds = (lds_total / w) & ~(size_t)255;
            if (lds > 64 * 1024) CHECK(hipFuncSetAttribute((const void *)k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const int blocks = cus * w;
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), lds, 0, d_out, d_sink, 1.0000001f, 0.5f);      // warm-up (clock ramp)
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), lds, 0, d_out, d_sink, 1.0000001f, 0.5f);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipDeviceSynchronize());
            float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const int n_waves = blocks * 4;
            CHECK(hipMemcpy(h.data(), d_out, sizeof(Stamp) * n_waves, hipMemcpyDeviceToHost));
            std::vector<double> cyc(n_waves), rt(n_waves);
            for (int i = 0; i < n_waves; ++i) { cyc[i] = (double)h[i].cycles; rt[i] = (double)h[i].realtime; }
            std::sort(cyc.begin(), cyc.end()); std::sort(rt.begin(), rt.end());
            const double n_inst = (double)kIters * kPerIter;
            const double med = cyc[n_waves / 2], med_rt = rt[n_waves / 2];
            printf("%s  {\"instr\": \"%s\", \"waves_per_simd\": %d, \"blocks\": %d, \"wall_ms\": %.4f, \"median_wave_cycles\": %.0f, "
                   "\"p05_wave_cycles\": %.0f, \"p95_wave_cycles\": %.0f, \"cycles_per_instr_per_simd\": %.4f, "
                   "\"cycles_per_instr_one_wave\": %.4f, \"clock_mhz\": %.1f, \"valu_instr_per_wave\": %.0f}",
                   first ? "" : ",\n", k.name, w, blocks, ms, med, cyc[n_waves / 20], cyc[n_waves - 1 - n_waves / 20],
                   med / (w * n_inst), med / n_inst, med_rt > 0 ? med / med_rt * 100.0 : 0.0, n_inst);
            first = false;
            CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
        }
    printf("\n]}\n");
    return 0;
}
