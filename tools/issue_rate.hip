// tools/issue_rate.hip -- VALU / LDS issue-rate microbenchmark for gfx950 (MI355X).
//
// Question it answers: how many shader cycles does ONE wave64 instruction of each class
// occupy a SIMD's issue for, with 1, 2, 4, 6, 8 waves resident per SIMD?  (the kernels' "VALU busy" figure multiplies
// SQ_INSTS_VALU by that constant).
//
// Method: every wave runs `iters` trips of a 128-instruction body: 8 independent register streams x 16 rounds of the
// same instruction (inline asm, volatile: the compiler neither removes nor reorders them), so neither a dependency
// chain (distance 8) nor the loop's scalar tail limits the rate.  Workgroups are 256 threads (4 waves = one per
// SIMD: the dispatcher deals a workgroup's waves round-robin over the CU's SIMDs) and reserve 160 KB / W of LDS, so
// exactly W of them are resident per CU and 256 x W fill the chip at W waves per SIMD; HW_ID / XCC_ID of every
// wave are recorded and the placement is verified on the host.  Time: s_memtime (shader cycles) around the loop per
// wave; the 100 MHz s_memrealtime beside it gives the clock the chip actually held.
//
// Result: cycles_per_inst_per_simd = launch time (HIP events) x measured clock / (iters x 128 x W): what one SIMD sustains with
// W waves resident; wave_cycles_per_own_inst_median = a wave's own elapsed cycles / its instructions (issue is arbitrated
// oldest-first, so the waves of a SIMD do not progress at equal rates: the SIMD figure is the robust one).
//
// Build + run (GPU box): tools/issue_rate.sh   -> gpurun_out/issue_rates.json (copied to profiles/r03_issue_rates.json)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

enum Op {
    OP_FMA_F32, OP_ADD_F32, OP_MUL_F32, OP_ADD_U32, OP_XOR_B32, OP_BITOP3, OP_LSHRREV_B64, OP_MUL_LO_U32, OP_MUL_HI_U32,
    OP_MAD_U64_U32, OP_SQRT_F32, OP_RCP_F32, OP_RSQ_F32, OP_DIV_SCALE_F32, OP_DIV_FMAS_F32, OP_DIV_FIXUP_F32,
    OP_FMA_F64, OP_ADD_F64, OP_MUL_F64, OP_CNDMASK, OP_CMP_F32, OP_PK_FMA_F32, OP_PK_MUL_F32, OP_PK_ADD_F32,
    OP_CVT_F64_F32, OP_CVT_F32_F64, OP_MOV_B32, OP_MAX3_F32, OP_READLANE, OP_MBCNT,
    OP_DS_READ_B32, OP_DS_READ_B128, OP_DS_WRITE_B32, OP_DS_ADD_F64, OP_DS_ADD_U32, OP_DS_BPERMUTE,
    OP_MIX_FMA_SALU,
    OP_CNDMASK_E64, OP_CNDMASK_VCC_ONES, OP_CMP_CNDMASK, OP_CMP_E64, OP_MIN_F32, OP_MAX_I32, OP_MED3_F32, OP_BFI_B32, OP_AND_OR_B32,
    OP_LSHL_ADD_U32, OP_ADD3_U32, OP_FMAC_F32, OP_SUB_F32, OP_FLOOR_F32, OP_CVT_I32_F32, OP_LSHL_ADD_U64, OP_ADD_CO_U32, OP_BCNT,
    OP_WRITELANE, OP_PERM_B32, OP_MOV_DPP, OP_MUL_U32_U24, OP_MAD_U32_U24, OP_MUL_F32_NEG, OP_LSHLREV_B32, OP_AND_B32, OP_SQRT_F64, OP_RCP_F64,
    OP_MIX_FMA_TRANS, OP_MIX_FMA_CND,
    OP_SALU_ONLY, OP_MIX_HALF_SALU, OP_MIX_FMA_NOP, OP_MIX_FMA_LDS, OP_MIX_FMA_BRANCH, OP_MIX_FMA_SAVEEXEC, OP_MIX_FMA_2SALU, OP_MIX_FMA_WAITCNT,
    OP_COUNT
};

static const char* kOpName[OP_COUNT] = {
    "v_fma_f32", "v_add_f32", "v_mul_f32", "v_add_u32", "v_xor_b32", "v_bitop3_b32", "v_lshrrev_b64", "v_mul_lo_u32", "v_mul_hi_u32",
    "v_mad_u64_u32", "v_sqrt_f32", "v_rcp_f32", "v_rsq_f32", "v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32",
    "v_fma_f64", "v_add_f64", "v_mul_f64", "v_cndmask_b32", "v_cmp_lt_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32",
    "v_cvt_f64_f32", "v_cvt_f32_f64", "v_mov_b32", "v_max3_f32", "v_readlane_b32", "v_mbcnt_lo_u32_b32",
    "ds_read_b32", "ds_read_b128", "ds_write_b32", "ds_add_f64", "ds_add_u32", "ds_bpermute_b32",
    "v_fma_f32+s_add_u32 (1:1)",
    "v_cndmask_b32_e64 (sgpr pair)", "v_cndmask_b32 vcc=-1", "v_cmp_lt_f32+v_cndmask_b32 (1:1)", "v_cmp_lt_f32_e64 (sgpr dst)", "v_min_f32", "v_max_i32", "v_med3_f32",
    "v_bfi_b32", "v_and_or_b32", "v_lshl_add_u32", "v_add3_u32", "v_fmac_f32", "v_sub_f32", "v_floor_f32", "v_cvt_i32_f32", "v_lshl_add_u64",
    "v_add_co_u32", "v_bcnt_u32_b32", "v_writelane_b32", "v_perm_b32", "v_mov_b32_dpp row_shr:1", "v_mul_u32_u24", "v_mad_u32_u24",
    "v_mul_f32_e64 neg", "v_lshlrev_b32", "v_and_b32", "v_sqrt_f64", "v_rcp_f64",
    "v_fma_f32+v_rcp_f32 (7:1)", "v_fma_f32+v_cndmask_b32 (7:1)",
    "s_add_u32 only (per SALU inst)", "v_lshlrev_b32+s_add_u32 (1:1, per VALU)", "v_fma_f32+s_nop (1:1, per VALU)", "v_fma_f32+ds_read_b32 (7:1, per inst)",
    "v_fma_f32+s_cbranch_scc0 not taken (1:1, per VALU)", "v_fma_f32+s_and_saveexec/s_or exec (4:2, per VALU)", "v_fma_f32+2 s_add_u32 (1:2, per VALU)",
    "v_fma_f32+s_waitcnt (1:1, per VALU)"
};

struct Rec { uint64_t t0, t1, r0, r1; uint32_t hwid, xcc; uint32_t pad[2]; };

#define R2(x) x x
#define R4(x) R2(x) R2(x)
#define R16(x) R4(x) R4(x) R4(x) R4(x)

// one round = the instruction once on each of the 8 streams
#define ROUND_F32(INS)                                                                                                    \
    asm volatile(INS " %0, %0, %8, %9\n" INS " %1, %1, %8, %9\n" INS " %2, %2, %8, %9\n" INS " %3, %3, %8, %9\n"          \
                 INS " %4, %4, %8, %9\n" INS " %5, %5, %8, %9\n" INS " %6, %6, %8, %9\n" INS " %7, %7, %8, %9\n"          \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
#define ROUND_F32_2(INS)                                                                                                  \
    asm volatile(INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n"                            \
                 INS " %4, %4, %8\n" INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8\n"                            \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
#define ROUND_F32_1(INS)                                                                                                  \
    asm volatile(INS " %0, %0\n" INS " %1, %1\n" INS " %2, %2\n" INS " %3, %3\n"                                            \
                 INS " %4, %4\n" INS " %5, %5\n" INS " %6, %6\n" INS " %7, %7\n"                                            \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
#define ROUND_D3(INS)                                                                                                     \
    asm volatile(INS " %0, %0, %8, %9\n" INS " %1, %1, %8, %9\n" INS " %2, %2, %8, %9\n" INS " %3, %3, %8, %9\n"          \
                 INS " %4, %4, %8, %9\n" INS " %5, %5, %8, %9\n" INS " %6, %6, %8, %9\n" INS " %7, %7, %8, %9\n"          \
                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(db), "v"(dc));
#define ROUND_D2(INS)                                                                                                     \
    asm volatile(INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n"                            \
                 INS " %4, %4, %8\n" INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8\n"                            \
                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(db));

template <int OP>
__global__ void __launch_bounds__(256) k_issue(Rec* __restrict__ rec, float* __restrict__ sink, int iters, float seedf) {
    extern __shared__ unsigned char lds_raw[];
    const int lane = threadIdx.x & 63;
    float a0 = seedf + lane, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    float b = 1.0000001f, c = 1e-9f;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7, db = 1.00000001, dc = 1e-12;
    uint32_t laddr = (threadIdx.x * 4u) & 0x3fffu;               // conflict-free dword per lane
    uint32_t laddr16 = (threadIdx.x * 16u) & 0x3fffu;
    uint32_t laddr8 = (threadIdx.x * 8u) & 0x3fffu;
    uint32_t bperm = ((lane * 7 + 3) & 63) * 4;
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 q0, q1, q2, q3, q4, q5, q6, q7;
    if ((OP >= OP_DS_READ_B32 && OP <= OP_DS_BPERMUTE) || OP == OP_MIX_FMA_LDS) {
        for (int i = threadIdx.x; i < 4096 + 1024; i += 256) reinterpret_cast<float*>(lds_raw)[i] = 0.0f;
    }
    __syncthreads();
    uint64_t t0, t1, r0, r1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\ns_memtime %0\ns_memrealtime %1\ns_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0));
    for (int it = 0; it < iters; ++it) {
        if constexpr (OP == OP_FMA_F32) { R16(ROUND_F32("v_fma_f32")) }
        else if constexpr (OP == OP_ADD_F32) { R16(ROUND_F32_2("v_add_f32")) }
        else if constexpr (OP == OP_MUL_F32) { R16(ROUND_F32_2("v_mul_f32")) }
        else if constexpr (OP == OP_ADD_U32) { R16(ROUND_F32_2("v_add_u32")) }
        else if constexpr (OP == OP_XOR_B32) { R16(ROUND_F32_2("v_xor_b32")) }
        else if constexpr (OP == OP_BITOP3) {
            R16(asm volatile("v_bitop3_b32 %0, %0, %8, %9 bitop3:0x6c\nv_bitop3_b32 %1, %1, %8, %9 bitop3:0x6c\n"
                             "v_bitop3_b32 %2, %2, %8, %9 bitop3:0x6c\nv_bitop3_b32 %3, %3, %8, %9 bitop3:0x6c\n"
                             "v_bitop3_b32 %4, %4, %8, %9 bitop3:0x6c\nv_bitop3_b32 %5, %5, %8, %9 bitop3:0x6c\n"
                             "v_bitop3_b32 %6, %6, %8, %9 bitop3:0x6c\nv_bitop3_b32 %7, %7, %8, %9 bitop3:0x6c\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        }
        else if constexpr (OP == OP_LSHRREV_B64) {
            R16(asm volatile("v_lshrrev_b64 %0, 1, %0\nv_lshrrev_b64 %1, 1, %1\nv_lshrrev_b64 %2, 1, %2\nv_lshrrev_b64 %3, 1, %3\n"
                             "v_lshrrev_b64 %4, 1, %4\nv_lshrrev_b64 %5, 1, %5\nv_lshrrev_b64 %6, 1, %6\nv_lshrrev_b64 %7, 1, %7\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));)
        }
        else if constexpr (OP == OP_MUL_LO_U32) { R16(ROUND_F32_2("v_mul_lo_u32")) }
        else if constexpr (OP == OP_MUL_HI_U32) { R16(ROUND_F32_2("v_mul_hi_u32")) }
        else if constexpr (OP == OP_MAD_U64_U32) {
            R16(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\nv_mad_u64_u32 %1, vcc, %8, %9, %1\n"
                             "v_mad_u64_u32 %2, vcc, %8, %9, %2\nv_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                             "v_mad_u64_u32 %4, vcc, %8, %9, %4\nv_mad_u64_u32 %5, vcc, %8, %9, %5\n"
                             "v_mad_u64_u32 %6, vcc, %8, %9, %6\nv_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(b), "v"(c) : "vcc");)
        }
        else if constexpr (OP == OP_SQRT_F32) { R16(ROUND_F32_1("v_sqrt_f32")) }
        else if constexpr (OP == OP_RCP_F32) { R16(ROUND_F32_1("v_rcp_f32")) }
        else if constexpr (OP == OP_RSQ_F32) { R16(ROUND_F32_1("v_rsq_f32")) }
        else if constexpr (OP == OP_DIV_SCALE_F32) {
            R16(asm volatile("v_div_scale_f32 %0, vcc, %0, %8, %0\nv_div_scale_f32 %1, vcc, %1, %8, %1\n"
                             "v_div_scale_f32 %2, vcc, %2, %8, %2\nv_div_scale_f32 %3, vcc, %3, %8, %3\n"
                             "v_div_scale_f32 %4, vcc, %4, %8, %4\nv_div_scale_f32 %5, vcc, %5, %8, %5\n"
                             "v_div_scale_f32 %6, vcc, %6, %8, %6\nv_div_scale_f32 %7, vcc, %7, %8, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
        }
        else if constexpr (OP == OP_DIV_FMAS_F32) {
            R16(asm volatile("v_div_fmas_f32 %0, %0, %8, %9\nv_div_fmas_f32 %1, %1, %8, %9\nv_div_fmas_f32 %2, %2, %8, %9\n"
                             "v_div_fmas_f32 %3, %3, %8, %9\nv_div_fmas_f32 %4, %4, %8, %9\nv_div_fmas_f32 %5, %5, %8, %9\n"
                             "v_div_fmas_f32 %6, %6, %8, %9\nv_div_fmas_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");)
        }
        else if constexpr (OP == OP_DIV_FIXUP_F32) { R16(ROUND_F32("v_div_fixup_f32")) }
        else if constexpr (OP == OP_FMA_F64) { R16(ROUND_D3("v_fma_f64")) }
        else if constexpr (OP == OP_ADD_F64) { R16(ROUND_D2("v_add_f64")) }
        else if constexpr (OP == OP_MUL_F64) { R16(ROUND_D2("v_mul_f64")) }
        else if constexpr (OP == OP_CNDMASK) {
            R16(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\n"
                             "v_cndmask_b32 %3, %3, %8, vcc\nv_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\n"
                             "v_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
        }
        else if constexpr (OP == OP_CMP_F32) {
            R16(asm volatile("v_cmp_lt_f32 vcc, %0, %8\nv_cmp_lt_f32 vcc, %1, %8\nv_cmp_lt_f32 vcc, %2, %8\nv_cmp_lt_f32 vcc, %3, %8\n"
                             "v_cmp_lt_f32 vcc, %4, %8\nv_cmp_lt_f32 vcc, %5, %8\nv_cmp_lt_f32 vcc, %6, %8\nv_cmp_lt_f32 vcc, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
        }
        else if constexpr (OP == OP_PK_FMA_F32) { R16(ROUND_D3("v_pk_fma_f32")) }
        else if constexpr (OP == OP_PK_MUL_F32) { R16(ROUND_D2("v_pk_mul_f32")) }
        else if constexpr (OP == OP_PK_ADD_F32) { R16(ROUND_D2("v_pk_add_f32")) }
        else if constexpr (OP == OP_CVT_F64_F32) {
            R16(asm volatile("v_cvt_f64_f32 %0, %8\nv_cvt_f64_f32 %1, %9\nv_cvt_f64_f32 %2, %10\nv_cvt_f64_f32 %3, %11\n"
                             "v_cvt_f64_f32 %4, %12\nv_cvt_f64_f32 %5, %13\nv_cvt_f64_f32 %6, %14\nv_cvt_f64_f32 %7, %15\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
                             : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));)
        }
        else if constexpr (OP == OP_CVT_F32_F64) {
            R16(asm volatile("v_cvt_f32_f64 %0, %8\nv_cvt_f32_f64 %1, %9\nv_cvt_f32_f64 %2, %10\nv_cvt_f32_f64 %3, %11\n"
                             "v_cvt_f32_f64 %4, %12\nv_cvt_f32_f64 %5, %13\nv_cvt_f32_f64 %6, %14\nv_cvt_f32_f64 %7, %15\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                             : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7));)
        }
        else if constexpr (OP == OP_MOV_B32) {
            R16(asm volatile("v_mov_b32 %0, %8\nv_mov_b32 %1, %8\nv_mov_b32 %2, %8\nv_mov_b32 %3, %8\n"
                             "v_mov_b32 %4, %8\nv_mov_b32 %5, %8\nv_mov_b32 %6, %8\nv_mov_b32 %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        }
        else if constexpr (OP == OP_MAX3_F32) { R16(ROUND_F32("v_max3_f32")) }
        else if constexpr (OP == OP_READLANE) {
            R16(asm volatile("v_readlane_b32 s40, %0, 3\nv_readlane_b32 s41, %1, 3\nv_readlane_b32 s42, %2, 3\nv_readlane_b32 s43, %3, 3\n"
                             "v_readlane_b32 s44, %4, 3\nv_readlane_b32 s45, %5, 3\nv_readlane_b32 s46, %6, 3\nv_readlane_b32 s47, %7, 3\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                             : : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");)
        }
        else if constexpr (OP == OP_MBCNT) {
            R16(asm volatile("v_mbcnt_lo_u32_b32 %0, -1, %0\nv_mbcnt_lo_u32_b32 %1, -1, %1\nv_mbcnt_lo_u32_b32 %2, -1, %2\n"
                             "v_mbcnt_lo_u32_b32 %3, -1, %3\nv_mbcnt_lo_u32_b32 %4, -1, %4\nv_mbcnt_lo_u32_b32 %5, -1, %5\n"
                             "v_mbcnt_lo_u32_b32 %6, -1, %6\nv_mbcnt_lo_u32_b32 %7, -1, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        }
        else if constexpr (OP == OP_DS_READ_B32) {
            R16(asm volatile("ds_read_b32 %0, %8\nds_read_b32 %1, %8 offset:256\nds_read_b32 %2, %8 offset:512\nds_read_b32 %3, %8 offset:768\n"
                             "ds_read_b32 %4, %8 offset:1024\nds_read_b32 %5, %8 offset:1280\nds_read_b32 %6, %8 offset:1536\nds_read_b32 %7, %8 offset:1792\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(laddr) : "memory");)
        }
        else if constexpr (OP == OP_DS_READ_B128) {
            R16(asm volatile("ds_read_b128 %0, %8\nds_read_b128 %1, %8 offset:16\nds_read_b128 %2, %8\nds_read_b128 %3, %8 offset:16\n"
                             "ds_read_b128 %4, %8\nds_read_b128 %5, %8 offset:16\nds_read_b128 %6, %8\nds_read_b128 %7, %8 offset:16\n"
                             : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3), "=v"(q4), "=v"(q5), "=v"(q6), "=v"(q7) : "v"(laddr16) : "memory");)
            asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
            a0 += q0.x + q1.y + q2.z + q3.w + q4.x + q5.y + q6.z + q7.w;
        }
        else if constexpr (OP == OP_DS_WRITE_B32) {
            R16(asm volatile("ds_write_b32 %8, %0\nds_write_b32 %8, %1 offset:256\nds_write_b32 %8, %2 offset:512\nds_write_b32 %8, %3 offset:768\n"
                             "ds_write_b32 %8, %4 offset:1024\nds_write_b32 %8, %5 offset:1280\nds_write_b32 %8, %6 offset:1536\nds_write_b32 %8, %7 offset:1792\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(laddr) : "memory");)
        }
        else if constexpr (OP == OP_DS_ADD_F64) {
            R16(asm volatile("ds_add_f64 %8, %0\nds_add_f64 %8, %1 offset:512\nds_add_f64 %8, %2 offset:1024\nds_add_f64 %8, %3 offset:1536\n"
                             "ds_add_f64 %8, %4 offset:2048\nds_add_f64 %8, %5 offset:2560\nds_add_f64 %8, %6 offset:3072\nds_add_f64 %8, %7 offset:3584\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(laddr8) : "memory");)
        }
        else if constexpr (OP == OP_DS_ADD_U32) {
            R16(asm volatile("ds_add_u32 %8, %0\nds_add_u32 %8, %1 offset:256\nds_add_u32 %8, %2 offset:512\nds_add_u32 %8, %3 offset:768\n"
                             "ds_add_u32 %8, %4 offset:1024\nds_add_u32 %8, %5 offset:1280\nds_add_u32 %8, %6 offset:1536\nds_add_u32 %8, %7 offset:1792\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(laddr) : "memory");)
        }
        else if constexpr (OP == OP_DS_BPERMUTE) {
            R16(asm volatile("ds_bpermute_b32 %0, %8, %0\nds_bpermute_b32 %1, %8, %1\nds_bpermute_b32 %2, %8, %2\nds_bpermute_b32 %3, %8, %3\n"
                             "ds_bpermute_b32 %4, %8, %4\nds_bpermute_b32 %5, %8, %5\nds_bpermute_b32 %6, %8, %6\nds_bpermute_b32 %7, %8, %7\n"
                             "s_waitcnt lgkmcnt(0)\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(bperm) : "memory");)
        }
        else if constexpr (OP == OP_MIX_FMA_SALU) {
            // 8 VALU + 8 SALU per round: does scalar issue ride beside the vector stream of the same wave / other waves?
            R16(asm volatile("v_fma_f32 %0, %0, %8, %9\ns_add_u32 s40, s40, 1\nv_fma_f32 %1, %1, %8, %9\ns_add_u32 s41, s41, 1\n"
                             "v_fma_f32 %2, %2, %8, %9\ns_add_u32 s42, s42, 1\nv_fma_f32 %3, %3, %8, %9\ns_add_u32 s43, s43, 1\n"
                             "v_fma_f32 %4, %4, %8, %9\ns_add_u32 s44, s44, 1\nv_fma_f32 %5, %5, %8, %9\ns_add_u32 s45, s45, 1\n"
                             "v_fma_f32 %6, %6, %8, %9\ns_add_u32 s46, s46, 1\nv_fma_f32 %7, %7, %8, %9\ns_add_u32 s47, s47, 1\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)
                             : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "scc");)
        }

        else if constexpr (OP == OP_CNDMASK_E64) {
            R16(asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[40:41]\nv_cndmask_b32_e64 %1, %1, %8, s[40:41]\nv_cndmask_b32_e64 %2, %2, %8, s[40:41]\n"
                             "v_cndmask_b32_e64 %3, %3, %8, s[40:41]\nv_cndmask_b32_e64 %4, %4, %8, s[40:41]\nv_cndmask_b32_e64 %5, %5, %8, s[40:41]\n"
                             "v_cndmask_b32_e64 %6, %6, %8, s[40:41]\nv_cndmask_b32_e64 %7, %7, %8, s[40:41]\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s40", "s41");)
        }
        else if constexpr (OP == OP_CNDMASK_VCC_ONES) {
            asm volatile("s_mov_b64 vcc, -1" : : : "vcc");
            R16(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\n"
                             "v_cndmask_b32 %3, %3, %8, vcc\nv_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\n"
                             "v_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
        }
        else if constexpr (OP == OP_CMP_CNDMASK) {
            R16(asm volatile("v_cmp_lt_f32 vcc, %0, %8\nv_cndmask_b32 %1, %1, %8, vcc\nv_cmp_lt_f32 vcc, %2, %8\nv_cndmask_b32 %3, %3, %8, vcc\n"
                             "v_cmp_lt_f32 vcc, %4, %8\nv_cndmask_b32 %5, %5, %8, vcc\nv_cmp_lt_f32 vcc, %6, %8\nv_cndmask_b32 %7, %7, %8, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
        }
        else if constexpr (OP == OP_CMP_E64) {
            R16(asm volatile("v_cmp_lt_f32_e64 s[40:41], %0, %8\nv_cmp_lt_f32_e64 s[42:43], %1, %8\nv_cmp_lt_f32_e64 s[44:45], %2, %8\nv_cmp_lt_f32_e64 s[46:47], %3, %8\n"
                             "v_cmp_lt_f32_e64 s[40:41], %4, %8\nv_cmp_lt_f32_e64 s[42:43], %5, %8\nv_cmp_lt_f32_e64 s[44:45], %6, %8\nv_cmp_lt_f32_e64 s[46:47], %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)
                             : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");)
        }
        else if constexpr (OP == OP_MIN_F32) { R16(ROUND_F32_2("v_min_f32")) }
        else if constexpr (OP == OP_MAX_I32) { R16(ROUND_F32_2("v_max_i32")) }
        else if constexpr (OP == OP_MED3_F32) { R16(ROUND_F32("v_med3_f32")) }
        else if constexpr (OP == OP_BFI_B32) { R16(ROUND_F32("v_bfi_b32")) }
        else if constexpr (OP == OP_AND_OR_B32) { R16(ROUND_F32("v_and_or_b32")) }
        else if constexpr (OP == OP_LSHL_ADD_U32) {
            R16(asm volatile("v_lshl_add_u32 %0, %0, 1, %8\nv_lshl_add_u32 %1, %1, 1, %8\nv_lshl_add_u32 %2, %2, 1, %8\nv_lshl_add_u32 %3, %3, 1, %8\n"
                             "v_lshl_add_u32 %4, %4, 1, %8\nv_lshl_add_u32 %5, %5, 1, %8\nv_lshl_add_u32 %6, %6, 1, %8\nv_lshl_add_u32 %7, %7, 1, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        }
        else if constexpr (OP == OP_ADD3_U32) { R16(ROUND_F32("v_add3_u32")) }
        else if constexpr (OP == OP_FMAC_F32) { R16(ROUND_F32_2("v_fmac_f32")) }
        else if constexpr (OP == OP_SUB_F32) { R16(ROUND_F32_2("v_sub_f32")) }
        else if constexpr (OP == OP_FLOOR_F32) { R16(ROUND_F32_1("v_floor_f32")) }
        else if constexpr (OP == OP_CVT_I32_F32) { R16(ROUND_F32_1("v_cvt_i32_f32")) }
        else if constexpr (OP == OP_LSHL_ADD_U64) {
            R16(asm volatile("v_lshl_add_u64 %0, %0, 1, %8\nv_lshl_add_u64 %1, %1, 1, %8\nv_lshl_add_u64 %2, %2, 1, %8\nv_lshl_add_u64 %3, %3, 1, %8\n"
                             "v_lshl_add_u64 %4, %4, 1, %8\nv_lshl_add_u64 %5, %5, 1, %8\nv_lshl_add_u64 %6, %6, 1, %8\nv_lshl_add_u64 %7, %7, 1, %8\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(db));)
        }
        else if constexpr (OP == OP_ADD_CO_U32) {
            R16(asm volatile("v_add_co_u32 %0, vcc, %0, %8\nv_add_co_u32 %1, vcc, %1, %8\nv_add_co_u32 %2, vcc, %2, %8\nv_add_co_u32 %3, vcc, %3, %8\n"
                             "v_add_co_u32 %4, vcc, %4, %8\nv_add_co_u32 %5, vcc, %5, %8\nv_add_co_u32 %6, vcc, %6, %8\nv_add_co_u32 %7, vcc, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
        }
        else if constexpr (OP == OP_BCNT) { R16(ROUND_F32_2("v_bcnt_u32_b32")) }
        else if constexpr (OP == OP_WRITELANE) {
            R16(asm volatile("v_writelane_b32 %0, s40, 3\nv_writelane_b32 %1, s40, 3\nv_writelane_b32 %2, s40, 3\nv_writelane_b32 %3, s40, 3\n"
                             "v_writelane_b32 %4, s40, 3\nv_writelane_b32 %5, s40, 3\nv_writelane_b32 %6, s40, 3\nv_writelane_b32 %7, s40, 3\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "s40");)
        }
        else if constexpr (OP == OP_PERM_B32) { R16(ROUND_F32("v_perm_b32")) }
        else if constexpr (OP == OP_MOV_DPP) {
            R16(asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        }
        else if constexpr (OP == OP_MUL_U32_U24) { R16(ROUND_F32_2("v_mul_u32_u24")) }
        else if constexpr (OP == OP_MAD_U32_U24) { R16(ROUND_F32("v_mad_u32_u24")) }
        else if constexpr (OP == OP_MUL_F32_NEG) {
            R16(asm volatile("v_mul_f32_e64 %0, -%0, |%8|\nv_mul_f32_e64 %1, -%1, |%8|\nv_mul_f32_e64 %2, -%2, |%8|\nv_mul_f32_e64 %3, -%3, |%8|\n"
                             "v_mul_f32_e64 %4, -%4, |%8|\nv_mul_f32_e64 %5, -%5, |%8|\nv_mul_f32_e64 %6, -%6, |%8|\nv_mul_f32_e64 %7, -%7, |%8|\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        }
        else if constexpr (OP == OP_LSHLREV_B32) {
            R16(asm volatile("v_lshlrev_b32 %0, 1, %0\nv_lshlrev_b32 %1, 1, %1\nv_lshlrev_b32 %2, 1, %2\nv_lshlrev_b32 %3, 1, %3\n"
                             "v_lshlrev_b32 %4, 1, %4\nv_lshlrev_b32 %5, 1, %5\nv_lshlrev_b32 %6, 1, %6\nv_lshlrev_b32 %7, 1, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        }
        else if constexpr (OP == OP_AND_B32) { R16(ROUND_F32_2("v_and_b32")) }
        else if constexpr (OP == OP_SQRT_F64) {
            R16(asm volatile("v_sqrt_f64 %0, %0\nv_sqrt_f64 %1, %1\nv_sqrt_f64 %2, %2\nv_sqrt_f64 %3, %3\nv_sqrt_f64 %4, %4\nv_sqrt_f64 %5, %5\nv_sqrt_f64 %6, %6\nv_sqrt_f64 %7, %7\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));)
        }
        else if constexpr (OP == OP_RCP_F64) {
            R16(asm volatile("v_rcp_f64 %0, %0\nv_rcp_f64 %1, %1\nv_rcp_f64 %2, %2\nv_rcp_f64 %3, %3\nv_rcp_f64 %4, %4\nv_rcp_f64 %5, %5\nv_rcp_f64 %6, %6\nv_rcp_f64 %7, %7\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));)
        }
        else if constexpr (OP == OP_MIX_FMA_TRANS) {
            R16(asm volatile("v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\nv_rcp_f32 %3, %3\n"
                             "v_fma_f32 %4, %4, %8, %9\nv_fma_f32 %5, %5, %8, %9\nv_fma_f32 %6, %6, %8, %9\nv_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        }
        else if constexpr (OP == OP_MIX_FMA_CND) {
            R16(asm volatile("v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\nv_cndmask_b32 %3, %3, %8, vcc\n"
                             "v_fma_f32 %4, %4, %8, %9\nv_fma_f32 %5, %5, %8, %9\nv_fma_f32 %6, %6, %8, %9\nv_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");)
        }

        else if constexpr (OP == OP_SALU_ONLY) {
            R16(asm volatile("s_add_u32 s40, s40, 1\ns_add_u32 s41, s41, 1\ns_add_u32 s42, s42, 1\ns_add_u32 s43, s43, 1\n"
                             "s_add_u32 s44, s44, 1\ns_add_u32 s45, s45, 1\ns_add_u32 s46, s46, 1\ns_add_u32 s47, s47, 1\n"
                             : : : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "scc");)
        }
        else if constexpr (OP == OP_MIX_HALF_SALU) {
            R16(asm volatile("v_lshlrev_b32 %0, 1, %0\ns_add_u32 s40, s40, 1\nv_lshlrev_b32 %1, 1, %1\ns_add_u32 s41, s41, 1\n"
                             "v_lshlrev_b32 %2, 1, %2\ns_add_u32 s42, s42, 1\nv_lshlrev_b32 %3, 1, %3\ns_add_u32 s43, s43, 1\n"
                             "v_lshlrev_b32 %4, 1, %4\ns_add_u32 s44, s44, 1\nv_lshlrev_b32 %5, 1, %5\ns_add_u32 s45, s45, 1\n"
                             "v_lshlrev_b32 %6, 1, %6\ns_add_u32 s46, s46, 1\nv_lshlrev_b32 %7, 1, %7\ns_add_u32 s47, s47, 1\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :
                             : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "scc");)
        }
        else if constexpr (OP == OP_MIX_FMA_NOP) {
            R16(asm volatile("v_fma_f32 %0, %0, %8, %9\ns_nop 0\nv_fma_f32 %1, %1, %8, %9\ns_nop 0\nv_fma_f32 %2, %2, %8, %9\ns_nop 0\nv_fma_f32 %3, %3, %8, %9\ns_nop 0\n"
                             "v_fma_f32 %4, %4, %8, %9\ns_nop 0\nv_fma_f32 %5, %5, %8, %9\ns_nop 0\nv_fma_f32 %6, %6, %8, %9\ns_nop 0\nv_fma_f32 %7, %7, %8, %9\ns_nop 0\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        }
        else if constexpr (OP == OP_MIX_FMA_LDS) {
            R16(asm volatile("v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\nds_read_b32 %3, %10\n"
                             "v_fma_f32 %4, %4, %8, %9\nv_fma_f32 %5, %5, %8, %9\nv_fma_f32 %6, %6, %8, %9\nv_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "v"(laddr) : "memory");)
            asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
        }
        else if constexpr (OP == OP_MIX_FMA_BRANCH) {
            R16(asm volatile("s_cmp_eq_u32 s40, s40\n"
                             "v_fma_f32 %0, %0, %8, %9\ns_cbranch_scc0 1f\nv_fma_f32 %1, %1, %8, %9\ns_cbranch_scc0 1f\nv_fma_f32 %2, %2, %8, %9\ns_cbranch_scc0 1f\n"
                             "v_fma_f32 %3, %3, %8, %9\ns_cbranch_scc0 1f\nv_fma_f32 %4, %4, %8, %9\ns_cbranch_scc0 1f\nv_fma_f32 %5, %5, %8, %9\ns_cbranch_scc0 1f\n"
                             "v_fma_f32 %6, %6, %8, %9\ns_cbranch_scc0 1f\nv_fma_f32 %7, %7, %8, %9\ns_cbranch_scc0 1f\n1:\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "s40", "scc");)
        }
        else if constexpr (OP == OP_MIX_FMA_SAVEEXEC) {
            // the compiler's divergent-if skeleton around four VALU instructions, twice per round
            R16(asm volatile("s_mov_b64 s[40:41], exec\n"
                             "s_and_saveexec_b64 s[42:43], s[40:41]\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\nv_fma_f32 %3, %3, %8, %9\ns_or_b64 exec, exec, s[42:43]\n"
                             "s_and_saveexec_b64 s[42:43], s[40:41]\nv_fma_f32 %4, %4, %8, %9\nv_fma_f32 %5, %5, %8, %9\nv_fma_f32 %6, %6, %8, %9\nv_fma_f32 %7, %7, %8, %9\ns_or_b64 exec, exec, s[42:43]\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "s40", "s41", "s42", "s43", "scc");)
        }
        else if constexpr (OP == OP_MIX_FMA_2SALU) {
            R16(asm volatile("v_fma_f32 %0, %0, %8, %9\ns_add_u32 s40, s40, 1\ns_add_u32 s41, s41, 1\nv_fma_f32 %1, %1, %8, %9\ns_add_u32 s42, s42, 1\ns_add_u32 s43, s43, 1\n"
                             "v_fma_f32 %2, %2, %8, %9\ns_add_u32 s44, s44, 1\ns_add_u32 s45, s45, 1\nv_fma_f32 %3, %3, %8, %9\ns_add_u32 s46, s46, 1\ns_add_u32 s47, s47, 1\n"
                             "v_fma_f32 %4, %4, %8, %9\ns_add_u32 s40, s40, 1\ns_add_u32 s41, s41, 1\nv_fma_f32 %5, %5, %8, %9\ns_add_u32 s42, s42, 1\ns_add_u32 s43, s43, 1\n"
                             "v_fma_f32 %6, %6, %8, %9\ns_add_u32 s44, s44, 1\ns_add_u32 s45, s45, 1\nv_fma_f32 %7, %7, %8, %9\ns_add_u32 s46, s46, 1\ns_add_u32 s47, s47, 1\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)
                             : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "scc");)
        }
        else if constexpr (OP == OP_MIX_FMA_WAITCNT) {
            R16(asm volatile("v_fma_f32 %0, %0, %8, %9\ns_waitcnt vmcnt(0)\nv_fma_f32 %1, %1, %8, %9\ns_waitcnt lgkmcnt(0)\nv_fma_f32 %2, %2, %8, %9\ns_waitcnt vmcnt(0)\n"
                             "v_fma_f32 %3, %3, %8, %9\ns_waitcnt lgkmcnt(0)\nv_fma_f32 %4, %4, %8, %9\ns_waitcnt vmcnt(0)\nv_fma_f32 %5, %5, %8, %9\ns_waitcnt lgkmcnt(0)\n"
                             "v_fma_f32 %6, %6, %8, %9\ns_waitcnt vmcnt(0)\nv_fma_f32 %7, %7, %8, %9\ns_waitcnt lgkmcnt(0)\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "memory");)
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\ns_memtime %0\ns_memrealtime %1\ns_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) : : "memory");
    uint32_t hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\ns_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hwid), "=s"(xcc));
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (lane == 0) {
        Rec r; r.t0 = t0; r.t1 = t1; r.r0 = r0; r.r1 = r1; r.hwid = hwid; r.xcc = xcc; r.pad[0] = r.pad[1] = 0;
        rec[wave] = r;
    }
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
    if (s == 123.456f) sink[threadIdx.x] = s;
}

typedef void (*KernelFn)(Rec*, float*, int, float);
template <int OP> struct Table { static void fill(KernelFn* t) { t[OP] = k_issue<OP>; Table<OP + 1>::fill(t); } };
template <> struct Table<OP_COUNT> { static void fill(KernelFn*) {} };

static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; }

int main(int argc, char** argv) {
    const char* out_path = argc > 1 ? argv[1] : "issue_rates.json";
    int iters = argc > 2 ? atoi(argv[2]) : 2048;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    KernelFn table[OP_COUNT];
    Table<0>::fill(table);
    const int Ws[] = {1, 2, 4, 6, 8};
    const int max_waves = n_cu * 8 * 4;
    Rec* d_rec; float* d_sink;
    CHECK(hipMalloc(&d_rec, sizeof(Rec) * max_waves));
    CHECK(hipMalloc(&d_sink, 4096));
    std::vector<Rec> h(max_waves);
    FILE* fo = fopen(out_path, "w");
    if (!fo) { perror(out_path); return 1; }
    fprintf(fo, "{\n \"device\": \"%s\", \"gcn_arch\": \"%s\", \"compute_units\": %d, \"clock_rate_khz\": %d,\n", prop.name, prop.gcnArchName, n_cu, prop.clockRate);
    fprintf(fo, " \"method\": \"tools/issue_rate.hip: 256-thread workgroups (one wave per SIMD), W workgroups resident per CU (LDS reservation), "
                "8 independent streams x 16 rounds per trip, %d trips; cycles = s_memtime ticks around the loop; "
                "cycles_per_inst_per_simd = launch time x measured clock / (instructions per wave x W)\",\n", iters);
    fprintf(fo, " \"instructions_per_wave\": %d,\n \"rates\": {\n", iters * 128);
    const int first_op = argc > 3 ? atoi(argv[3]) : 0;
    for (int op = first_op; op < OP_COUNT; ++op) {
        fprintf(fo, "  \"%s\": {", kOpName[op]);
        for (int wi = 0; wi < 5; ++wi) {
            const int W = Ws[wi];
            size_t lds = (size_t)(160 * 1024 / W) & ~(size_t)1023;
            if (lds < 24 * 1024) lds = (size_t)(160 * 1024 / W) & ~(size_t)255;
            CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(table[op]), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const int grid = n_cu * W;
            const int n_waves = grid * 4;
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            hipLaunchKernelGGL(table[op], dim3(grid), dim3(256), lds, 0, d_rec, d_sink, iters / 8, 1.0f);   // warm
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(table[op], dim3(grid), dim3(256), lds, 0, d_rec, d_sink, iters, 1.0f);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(h.data(), d_rec, sizeof(Rec) * n_waves, hipMemcpyDeviceToHost));
            // placement: waves per (xcc, se, cu, simd)
            std::map<uint32_t, int> per_simd;
            for (int w = 0; w < n_waves; ++w) {
                uint32_t id = h[w].hwid;
                uint32_t key = ((h[w].xcc & 0xf) << 16) | (((id >> 13) & 7) << 12) | (((id >> 8) & 0xf) << 4) | ((id >> 4) & 3);
                per_simd[key]++;
            }
            int wmin = 1 << 30, wmax = 0;
            for (auto& kv : per_simd) { wmin = std::min(wmin, kv.second); wmax = std::max(wmax, kv.second); }
            std::vector<double> cyc, ghz;
            for (int w = 0; w < n_waves; ++w) {
                double el = (double)(h[w].t1 - h[w].t0);
                uint32_t id = h[w].hwid;
                uint32_t key = ((h[w].xcc & 0xf) << 16) | (((id >> 13) & 7) << 12) | (((id >> 8) & 0xf) << 4) | ((id >> 4) & 3);
                cyc.push_back(el / ((double)iters * 128.0));
                double rt = (double)(h[w].r1 - h[w].r0);          // 100 MHz ticks
                if (rt > 0) ghz.push_back(el / (rt * 10.0));
            }
            // residency actually reached: per SIMD, the number of waves whose [t0, t1) contains the SIMD's median start..end midpoint
            std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> spans;
            for (int w = 0; w < n_waves; ++w) {
                uint32_t id = h[w].hwid;
                uint32_t key = ((h[w].xcc & 0xf) << 16) | (((id >> 13) & 7) << 12) | (((id >> 8) & 0xf) << 4) | ((id >> 4) & 3);
                spans[key].push_back(std::make_pair(h[w].r0, h[w].r1));          // 100 MHz clock: comparable across the chip
            }
            std::vector<double> conc;
            for (auto& kv : spans) {
                // largest number of spans that overlap at one instant (sweep)
                std::vector<std::pair<uint64_t, int>> ev;
                for (auto& sp : kv.second) { ev.push_back(std::make_pair(sp.first, 1)); ev.push_back(std::make_pair(sp.second, -1)); }
                std::sort(ev.begin(), ev.end());
                int cur = 0, best = 0;
                for (auto& e : ev) { cur += e.second; best = std::max(best, cur); }
                conc.push_back((double)best);
            }
            const double resident = median(conc);
            const double c = median(cyc), g = median(ghz);
            // event-based cross-check: all waves of a SIMD issue W x iters x 128 instructions within the launch
            const double c_evt = (ms * 1e-3 * g * 1e9) / ((double)iters * 128.0 * W);
            fprintf(fo, "%s\"w%d\": {\"wave_cycles_per_own_inst_median\": %.3f, \"cycles_per_inst_per_simd\": %.3f, \"launch_ms\": %.4f, \"memtime_ghz\": %.3f, "
                        "\"simds_used\": %d, \"waves_per_simd_min\": %d, \"waves_per_simd_max\": %d, \"resident_waves_per_simd\": %.0f}",
                    wi ? ", " : "", W, c, c_evt, ms, g, (int)per_simd.size(), wmin, wmax, resident);
            printf("%-32s W=%d  wave: %.3f cyc per own inst | SIMD: %.3f cyc/inst  %.3f ms  %.3f GHz  simds %d  waves/simd %d..%d resident %.0f\n",
                   kOpName[op], W, c, c_evt, ms, g, (int)per_simd.size(), wmin, wmax, resident);
            fflush(stdout);
            CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
        }
        fprintf(fo, "}%s\n", op + 1 < OP_COUNT ? "," : "");
    }
    fprintf(fo, " }\n}\n");
    fclose(fo);
    return 0;
}
