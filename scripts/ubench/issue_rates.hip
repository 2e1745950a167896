// Micro-benchmark (design aid, not product): what one gfx950 SIMD issues per cycle for the instruction classes the
// literal-coder kernels are made of, at 1..8 waves per SIMD, and what one CU's LDS pipeline accepts.
// Settles whether a wave64 integer VALU op occupies its SIMD for 2 or 4 cycles.
//
//   hipcc --offload-arch=gfx950 -O3 -o issue_rates issue_rates.hip && ./issue_rates
//
// Every kernel runs ITERS iterations of UNROLL copies of one instruction on independent registers (or one dependent
// chain), 256-thread workgroups (= one wave per SIMD), grid = 256 CUs x waves-per-SIMD.  Reported: SIMD cycles per
// wave-instruction = shader cycles of the whole kernel (s_memtime of one wave) * waves-per-SIMD-share ... simply
// (kernel time * clock) / (ITERS * UNROLL * waves_per_simd).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <string>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int ITERS = 8192;
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));

// 8 independent destination registers, the asm text uses %0..%7 as dst (also src where a chain is wanted), %8/%9 as extra sources
#define BODY8(TXT)                                                                                                  \
    asm volatile(TXT(0) TXT(1) TXT(2) TXT(3) TXT(4) TXT(5) TXT(6) TXT(7)                                              \
                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)                     \
                 : "v"(a), "v"(b) : "vcc", "s20", "s21");

#define DEF_KERNEL(NAME, TXT)                                                                                       \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint64_t* ticks, uint32_t seed) {                    \
        uint32_t r0 = threadIdx.x + seed, r1 = r0 * 3u, r2 = r0 * 5u, r3 = r0 * 7u, r4 = r0 ^ 11u, r5 = r0 + 13u, r6 = r0 | 17u, r7 = r0 + 19u; \
        uint32_t a = seed | 1u, b = threadIdx.x | 3u;                                                               \
        uint64_t t0 = __builtin_readcyclecounter();                                                                 \
        for (int i = 0; i < ITERS; ++i) { BODY8(TXT) BODY8(TXT) }                                                   \
        uint64_t t1 = __builtin_readcyclecounter();                                                                 \
        if ((r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7) == 0xdeadbeefu) out[0] = r0;                                    \
        if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;                                                \
    }
constexpr int UNROLL = 16;

#define T_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define T_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define T_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 3, %9\n"
#define T_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define T_MULHI(i) "v_mul_hi_u32 %" #i ", %" #i ", %8\n"
#define T_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define T_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define T_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define T_CVTFU(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define T_CVTUF(i) "v_cvt_u32_f32 %" #i ", %" #i "\n"
#define T_MULF(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define T_FMAF(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define T_DPPB(i) "v_mov_b32_dpp %" #i ", %" #i " row_newbcast:15 row_mask:0xf bank_mask:0xf\n"
#define T_DPPS(i) "v_mov_b32_dpp %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define T_ADDDPP(i) "v_add_u32_dpp %" #i ", %" #i ", %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define T_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define T_CMP(i) "v_cmp_ge_u32 vcc, %" #i ", %8\n"
#define T_CMPS(i) "v_cmp_ge_u32 s[20:21], %" #i ", %8\n"
#define T_BCNT(i) "v_bcnt_u32_b32 %" #i ", %" #i ", %8\n"
#define T_PKADD(i) "v_pk_add_u16 %" #i ", %" #i ", %8\n"
#define T_PKSUB(i) "v_pk_sub_i16 %" #i ", %" #i ", %8\n"
#define T_PKLSHR(i) "v_pk_lshrrev_b16 %" #i ", 2, %" #i "\n"
#define T_PKMULLO(i) "v_pk_mul_lo_u16 %" #i ", %" #i ", %8\n"
#define T_PKMAX(i) "v_pk_max_u16 %" #i ", %" #i ", %8\n"
#define T_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define T_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 3, 9\n"
#define T_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define T_ALIGNBIT(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 15\n"
#define T_LSHR(i) "v_lshrrev_b32 %" #i ", 3, %" #i "\n"
#define T_SAD(i) "v_sad_u16 %" #i ", %" #i ", %8, %9\n"
#define T_READLANE(i) "v_readlane_b32 s20, %" #i ", 5\n"
#define T_READFIRST(i) "v_readfirstlane_b32 s20, %" #i "\n"
#define T_SNOP(i) "s_add_u32 s20, s20, 1\n"
#define T_BPERM(i) "ds_bpermute_b32 %" #i ", %8, %" #i "\n"
#define T_SWIZ(i) "ds_swizzle_b32 %" #i ", %" #i " offset:0x8000\n"
#define T_PERMLANE(i) "v_permlane32_swap %" #i ", %" #i "\n"
#define T_FFBL(i) "v_ffbl_b32 %" #i ", %" #i "\n"
#define T_FFBH(i) "v_ffbh_u32 %" #i ", %" #i "\n"

DEF_KERNEL(k_add, T_ADD)
DEF_KERNEL(k_add3, T_ADD3)
DEF_KERNEL(k_lshlor, T_LSHLOR)
DEF_KERNEL(k_mullo, T_MULLO)
DEF_KERNEL(k_mulhi, T_MULHI)
DEF_KERNEL(k_mul24, T_MUL24)
DEF_KERNEL(k_mad24, T_MAD24)
DEF_KERNEL(k_rcp, T_RCP)
DEF_KERNEL(k_cvtfu, T_CVTFU)
DEF_KERNEL(k_cvtuf, T_CVTUF)
DEF_KERNEL(k_mulf, T_MULF)
DEF_KERNEL(k_fmaf, T_FMAF)
DEF_KERNEL(k_dppb, T_DPPB)
DEF_KERNEL(k_dpps, T_DPPS)
DEF_KERNEL(k_adddpp, T_ADDDPP)
DEF_KERNEL(k_cndmask, T_CNDMASK)
DEF_KERNEL(k_cmp, T_CMP)
DEF_KERNEL(k_cmps, T_CMPS)
DEF_KERNEL(k_bcnt, T_BCNT)
DEF_KERNEL(k_pkadd, T_PKADD)
DEF_KERNEL(k_pksub, T_PKSUB)
DEF_KERNEL(k_pklshr, T_PKLSHR)
DEF_KERNEL(k_pkmullo, T_PKMULLO)
DEF_KERNEL(k_pkmax, T_PKMAX)
DEF_KERNEL(k_perm, T_PERM)
DEF_KERNEL(k_bfe, T_BFE)
DEF_KERNEL(k_andor, T_ANDOR)
DEF_KERNEL(k_alignbit, T_ALIGNBIT)
DEF_KERNEL(k_lshr, T_LSHR)
DEF_KERNEL(k_sad, T_SAD)
DEF_KERNEL(k_readlane, T_READLANE)
DEF_KERNEL(k_readfirst, T_READFIRST)
DEF_KERNEL(k_salu, T_SNOP)
DEF_KERNEL(k_bperm, T_BPERM)
DEF_KERNEL(k_swizzle, T_SWIZ)
DEF_KERNEL(k_ffbl, T_FFBL)
DEF_KERNEL(k_ffbh, T_FFBH)


#define T_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define T_OR(i) "v_or_b32 %" #i ", %" #i ", %8\n"
#define T_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define T_SUB(i) "v_sub_u32 %" #i ", %" #i ", %8\n"
#define T_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define T_NOT(i) "v_not_b32 %" #i ", %" #i "\n"
#define T_MAXU(i) "v_max_u32 %" #i ", %" #i ", %8\n"
#define T_MINI(i) "v_min_i32 %" #i ", %" #i ", %8\n"
#define T_MAXF(i) "v_max_f32 %" #i ", %" #i ", %8\n"
#define T_ADDF(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define T_FMAC(i) "v_fmac_f32 %" #i ", %8, %9\n"
#define T_LSHL(i) "v_lshlrev_b32 %" #i ", 3, %" #i "\n"
#define T_ASHR(i) "v_ashrrev_i32 %" #i ", 3, %" #i "\n"
#define T_ADDCO(i) "v_add_co_u32 %" #i ", vcc, %" #i ", %8\n"
#define T_ADDC(i) "v_addc_co_u32 %" #i ", vcc, %" #i ", %8, vcc\n"
#define T_CNDE64(i) "v_cndmask_b32 %" #i ", %" #i ", %8, s[20:21]\n"
#define T_CNDVCC(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define T_FLOOR(i) "v_floor_f32 %" #i ", %" #i "\n"
#define T_TRUNC(i) "v_trunc_f32 %" #i ", %" #i "\n"
#define T_CVTI(i) "v_cvt_f32_i32 %" #i ", %" #i "\n"
#define T_CMPF(i) "v_cmp_ge_f32 vcc, %" #i ", %8\n"
#define T_MED3(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n"
#define T_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 3, %9\n"
#define T_ADDLSHL(i) "v_add_lshl_u32 %" #i ", %" #i ", %8, 2\n"
#define T_XAD(i) "v_xad_u32 %" #i ", %" #i ", %8, %9\n"
#define T_MADF(i) "v_mad_u32_u16 %" #i ", %" #i ", %8, %9\n"
#define T_MULI24(i) "v_mul_i32_i24 %" #i ", %" #i ", %8\n"
#define T_BITOP3(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0x48\n"
#define T_PKMAD(i) "v_pk_mad_u16 %" #i ", %" #i ", %8, %9\n"
#define T_PKMULF(i) "v_pk_mul_f32 %" #i ", %" #i ", %" #i "\n"
#define T_DOT2(i) "v_dot2_u32_u16 %" #i ", %" #i ", %8, %9\n"
#define T_DOT4(i) "v_dot4_u32_u8 %" #i ", %" #i ", %8, %9\n"
#define T_MBCNT(i) "v_mbcnt_lo_u32_b32 %" #i ", %" #i ", %8\n"
#define T_SDWA(i) "v_add_u32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0\n"
DEF_KERNEL(k_and, T_AND) DEF_KERNEL(k_or, T_OR) DEF_KERNEL(k_xor, T_XOR) DEF_KERNEL(k_sub, T_SUB) DEF_KERNEL(k_mov, T_MOV) DEF_KERNEL(k_not, T_NOT)
DEF_KERNEL(k_maxu, T_MAXU) DEF_KERNEL(k_mini, T_MINI) DEF_KERNEL(k_maxf, T_MAXF) DEF_KERNEL(k_addf, T_ADDF) DEF_KERNEL(k_fmac, T_FMAC)
DEF_KERNEL(k_lshl, T_LSHL) DEF_KERNEL(k_ashr, T_ASHR) DEF_KERNEL(k_addco, T_ADDCO) DEF_KERNEL(k_addc, T_ADDC) DEF_KERNEL(k_cnde64, T_CNDE64)
DEF_KERNEL(k_cndvcc, T_CNDVCC) DEF_KERNEL(k_floor, T_FLOOR) DEF_KERNEL(k_trunc, T_TRUNC) DEF_KERNEL(k_cvti, T_CVTI) DEF_KERNEL(k_cmpf, T_CMPF)
DEF_KERNEL(k_med3, T_MED3) DEF_KERNEL(k_lshladd, T_LSHLADD) DEF_KERNEL(k_addlshl, T_ADDLSHL) DEF_KERNEL(k_xad, T_XAD) DEF_KERNEL(k_madu16, T_MADF)
DEF_KERNEL(k_muli24, T_MULI24) DEF_KERNEL(k_bitop3, T_BITOP3) DEF_KERNEL(k_pkmad, T_PKMAD) DEF_KERNEL(k_dot2, T_DOT2) DEF_KERNEL(k_dot4, T_DOT4)
DEF_KERNEL(k_mbcnt, T_MBCNT) DEF_KERNEL(k_sdwa, T_SDWA)

// 64-bit forms need register pairs
__global__ __launch_bounds__(256) void k_pairs(uint32_t* out, uint64_t* ticks, uint32_t seed, int which) {
    uint64_t r0 = threadIdx.x + seed, r1 = r0 * 3u, r2 = r0 * 5u, r3 = r0 * 7u, r4 = r0 ^ 11u, r5 = r0 + 13u, r6 = r0 | 17u, r7 = r0 + 19u;
    uint32_t a = seed | 1u, b = threadIdx.x | 3u;
    uint64_t t0 = __builtin_readcyclecounter();
#define P8(TXT) asm volatile(TXT(0) TXT(1) TXT(2) TXT(3) TXT(4) TXT(5) TXT(6) TXT(7) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b) : "vcc", "s20", "s21");
#define T_LSHL64(i) "v_lshlrev_b64 %" #i ", 3, %" #i "\n"
#define T_LSHR64(i) "v_lshrrev_b64 %" #i ", 15, %" #i "\n"
#define T_MAD64(i) "v_mad_u64_u32 %" #i ", vcc, %8, %9, %" #i "\n"
#define T_MULF64(i) "v_mul_f64 %" #i ", %" #i ", %" #i "\n"
#define T_FMAF64(i) "v_fma_f64 %" #i ", %" #i ", %" #i ", %" #i "\n"
#define T_CVTF64(i) "v_cvt_f64_u32 %" #i ", %8\n"
#define T_RCPF64(i) "v_rcp_f64 %" #i ", %" #i "\n"
#define T_PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %" #i ", %" #i "\n"
#define T_CMPU64(i) "v_cmp_gt_u64 vcc, %" #i ", %" #i "\n"
#define T_MOV64(i) "v_mov_b64 %" #i ", %" #i "\n"
#define T_LSHLADD64(i) "v_lshl_add_u64 %" #i ", %" #i ", 0, %" #i "\n"
#define T_LSHR64V(i) "v_lshrrev_b64 %" #i ", %8, %" #i "\n"
    if (which == 0) for (int i = 0; i < ITERS; ++i) { P8(T_LSHL64) P8(T_LSHL64) }
    else if (which == 1) for (int i = 0; i < ITERS; ++i) { P8(T_LSHR64) P8(T_LSHR64) }
    else if (which == 2) for (int i = 0; i < ITERS; ++i) { P8(T_MAD64) P8(T_MAD64) }
    else if (which == 3) for (int i = 0; i < ITERS; ++i) { P8(T_MULF64) P8(T_MULF64) }
    else if (which == 4) for (int i = 0; i < ITERS; ++i) { P8(T_FMAF64) P8(T_FMAF64) }
    else if (which == 5) for (int i = 0; i < ITERS; ++i) { P8(T_CVTF64) P8(T_CVTF64) }
    else if (which == 6) for (int i = 0; i < ITERS; ++i) { P8(T_RCPF64) P8(T_RCPF64) }
    else if (which == 7) for (int i = 0; i < ITERS; ++i) { P8(T_PKFMA) P8(T_PKFMA) }
    else if (which == 8) for (int i = 0; i < ITERS; ++i) { P8(T_CMPU64) P8(T_CMPU64) }
    else if (which == 9) for (int i = 0; i < ITERS; ++i) { P8(T_MOV64) P8(T_MOV64) }
    else if (which == 10) for (int i = 0; i < ITERS; ++i) { P8(T_LSHLADD64) P8(T_LSHLADD64) }
    else for (int i = 0; i < ITERS; ++i) { P8(T_LSHR64V) P8(T_LSHR64V) }
    uint64_t t1 = __builtin_readcyclecounter();
    if ((r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7) == 0xdeadbeefull) out[0] = (uint32_t)r0;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

// dependent chain: every instruction reads the previous one's result
__global__ __launch_bounds__(256) void k_chain(uint32_t* out, uint64_t* ticks, uint32_t seed, int which) {
    uint32_t r0 = threadIdx.x + seed, a = seed | 1u, b = threadIdx.x | 3u;
    uint64_t t0 = __builtin_readcyclecounter();
#define C16(TXT) asm volatile(TXT(0) TXT(0) TXT(0) TXT(0) TXT(0) TXT(0) TXT(0) TXT(0) TXT(0) TXT(0) TXT(0) TXT(0) TXT(0) TXT(0) TXT(0) TXT(0) : "+v"(r0) : "v"(a), "v"(b) : "vcc", "s20", "s21");
#define U_ADD(i) "v_add_u32 %0, %0, %1\n"
#define U_MULLO(i) "v_mul_lo_u32 %0, %0, %1\n"
#define U_DPPB(i) "v_mov_b32_dpp %0, %0 row_newbcast:15 row_mask:0xf bank_mask:0xf\ns_nop 1\n"
#define U_RCP(i) "v_rcp_f32 %0, %0\n"
#define U_BPERM(i) "ds_bpermute_b32 %0, %1, %0\ns_waitcnt lgkmcnt(0)\n"
#define U_CVT(i) "v_cvt_f32_u32 %0, %0\n"
    if (which == 0) for (int i = 0; i < ITERS; ++i) { C16(U_ADD) }
    else if (which == 1) for (int i = 0; i < ITERS; ++i) { C16(U_MULLO) }
    else if (which == 2) for (int i = 0; i < ITERS; ++i) { C16(U_DPPB) }
    else if (which == 3) for (int i = 0; i < ITERS; ++i) { C16(U_RCP) }
    else if (which == 4) for (int i = 0; i < ITERS; ++i) { C16(U_BPERM) }
    else for (int i = 0; i < ITERS; ++i) { C16(U_CVT) }
    uint64_t t1 = __builtin_readcyclecounter();
    if (r0 == 0xdeadbeefu) out[0] = r0;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

// LDS: 16 independent ops per iteration per wave, conflict-free addressing (lane-linear)
__global__ __launch_bounds__(256) void k_lds(uint32_t* out, uint64_t* ticks, uint32_t seed, int which) {
    __shared__ __attribute__((aligned(16))) uint32_t buf[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) buf[i] = i * seed;
    __syncthreads();
    uint32_t acc = 0;
    const uint32_t lane = threadIdx.x;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        if (which == 0) {          // ds_read_u16, 2 bytes per lane, 16 lanes = one 32-byte row
#pragma unroll
            for (int k = 0; k < 16; ++k) acc += ((volatile uint16_t*)buf)[lane + 512 * k];
        } else if (which == 1) {   // ds_write_b16
#pragma unroll
            for (int k = 0; k < 16; ++k) ((volatile uint16_t*)buf)[lane + 512 * k] = (uint16_t)(acc + k);
        } else if (which == 2) {   // ds_read_b32
#pragma unroll
            for (int k = 0; k < 16; ++k) acc += ((volatile uint32_t*)buf)[lane + 256 * k];
        } else if (which == 3) {   // ds_write_b32
#pragma unroll
            for (int k = 0; k < 16; ++k) ((volatile uint32_t*)buf)[lane + 256 * k] = acc + k;
        } else if (which == 4) {   // ds_read_b128
#pragma unroll
            for (int k = 0; k < 8; ++k) { v4u v = ((volatile v4u*)buf)[lane + 256 * k]; acc += v.x + v.w; }
        } else if (which == 5) {   // ds_write_b128
#pragma unroll
            for (int k = 0; k < 8; ++k) { v4u v = {acc, acc + 1, acc + 2, acc + k}; ((volatile v4u*)buf)[lane + 256 * k] = v; }
        } else if (which == 6) {   // ds_read_b64
#pragma unroll
            for (int k = 0; k < 16; ++k) { v2u v = ((volatile v2u*)buf)[lane + 256 * k]; acc += v.x + v.y; }
        } else {                   // ds_read_u16 with the row-cache pattern: each 16-lane row reads its own 32-byte row at a random-ish offset
#pragma unroll
            for (int k = 0; k < 16; ++k) acc += ((volatile uint16_t*)buf)[(lane & 15) + 16 * (((lane >> 4) * 37 + k * 11 + (acc & 1)) & 1023)];
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (acc == 0xdeadbeefu) out[0] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

typedef void (*K3)(uint32_t*, uint64_t*, uint32_t);

int main() {
    uint32_t* out; uint64_t* ticks;
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&ticks, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs, clockRate %d kHz\n", prop.name, ncu, prop.clockRate);
    auto report = [&](const char* name, int wps, float ms, int per_iter) {
        uint64_t tk; CK(hipMemcpy(&tk, ticks, 8, hipMemcpyDeviceToHost));
        // s_memtime counts a constant-rate clock on gfx9 (100 MHz); report wall-clock based SIMD cycles at the nominal 2.4 GHz too
        const double insts_per_simd = (double)ITERS * per_iter * wps;
        printf("%-22s waves/SIMD %d : %7.3f ms  %6.2f ns per wave-inst per SIMD = %5.2f cycles @2.4GHz   (memtime ticks %llu)\n", name, wps, ms,
               ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4, (unsigned long long)tk);
    };
    struct V { const char* name; K3 k; } vs[] = {
        {"v_and_b32", k_and}, {"v_or_b32", k_or}, {"v_xor_b32", k_xor}, {"v_sub_u32", k_sub}, {"v_mov_b32", k_mov}, {"v_not_b32", k_not},
        {"v_max_u32", k_maxu}, {"v_min_i32", k_mini}, {"v_max_f32", k_maxf}, {"v_add_f32", k_addf}, {"v_fmac_f32", k_fmac}, {"v_lshlrev_b32", k_lshl},
        {"v_ashrrev_i32", k_ashr}, {"v_add_co_u32", k_addco}, {"v_addc_co_u32", k_addc}, {"v_cndmask e64 sgpr", k_cnde64}, {"v_cndmask vcc (2)", k_cndvcc},
        {"v_floor_f32", k_floor}, {"v_trunc_f32", k_trunc}, {"v_cvt_f32_i32", k_cvti}, {"v_cmp_ge_f32", k_cmpf}, {"v_med3_f32", k_med3},
        {"v_lshl_add_u32", k_lshladd}, {"v_add_lshl_u32", k_addlshl}, {"v_xad_u32", k_xad}, {"v_mad_u32_u16", k_madu16}, {"v_mul_i32_i24", k_muli24},
        {"v_bitop3_b32", k_bitop3}, {"v_pk_mad_u16", k_pkmad}, {"v_dot2_u32_u16", k_dot2}, {"v_dot4_u32_u8", k_dot4}, {"v_mbcnt_lo", k_mbcnt}, {"v_add_u32_sdwa", k_sdwa},
        {"v_add_u32", k_add}, {"v_add3_u32", k_add3}, {"v_lshl_or_b32", k_lshlor}, {"v_mul_lo_u32", k_mullo}, {"v_mul_hi_u32", k_mulhi},
        {"v_mul_u32_u24", k_mul24}, {"v_mad_u32_u24", k_mad24}, {"v_rcp_f32", k_rcp}, {"v_cvt_f32_u32", k_cvtfu}, {"v_cvt_u32_f32", k_cvtuf},
        {"v_mul_f32", k_mulf}, {"v_fma_f32", k_fmaf}, {"v_mov_dpp newbcast", k_dppb}, {"v_mov_dpp row_shr", k_dpps}, {"v_add_u32_dpp", k_adddpp},
        {"v_cndmask_b32", k_cndmask}, {"v_cmp vcc", k_cmp}, {"v_cmp sgpr", k_cmps}, {"v_bcnt", k_bcnt}, {"v_pk_add_u16", k_pkadd},
        {"v_pk_sub_i16", k_pksub}, {"v_pk_lshrrev_b16", k_pklshr}, {"v_pk_mul_lo_u16", k_pkmullo}, {"v_pk_max_u16", k_pkmax}, {"v_perm_b32", k_perm},
        {"v_bfe_u32", k_bfe}, {"v_and_or_b32", k_andor}, {"v_alignbit_b32", k_alignbit}, {"v_lshrrev_b32", k_lshr}, {"v_sad_u16", k_sad},
        {"ds_bpermute_b32", k_bperm}, {"ds_swizzle_b32", k_swizzle},
        {"v_ffbl_b32", k_ffbl}, {"v_ffbh_u32", k_ffbh},
    };
    setvbuf(stdout, nullptr, _IOLBF, 0);
    for (auto& v : vs) {
        for (int wps : {1, 4, 8}) {
            v.k<<<ncu * wps, 256>>>(out, ticks, 1u);
            CK(hipEventRecord(e0));
            v.k<<<ncu * wps, 256>>>(out, ticks, 1u);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            report(v.name, wps, ms, UNROLL);
        }
    }
    const char* pn[] = {"v_lshlrev_b64", "v_lshrrev_b64", "v_mad_u64_u32", "v_mul_f64", "v_fma_f64", "v_cvt_f64_u32", "v_rcp_f64", "v_pk_fma_f32", "v_cmp_gt_u64", "v_mov_b64", "v_lshl_add_u64", "v_lshrrev_b64 vgpr shift"};
    for (int w = 0; w < 12; ++w)
        for (int wps : {1, 2, 4, 8}) {
            k_pairs<<<ncu * wps, 256>>>(out, ticks, 1u, w);
            CK(hipEventRecord(e0));
            k_pairs<<<ncu * wps, 256>>>(out, ticks, 1u, w);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            report(pn[w], wps, ms, UNROLL);
        }
    const char* cn[] = {"chain v_add_u32", "chain v_mul_lo_u32", "chain dpp bcast+nop", "chain v_rcp_f32", "chain ds_bpermute", "chain v_cvt_f32_u32"};
    for (int w = 0; w < 6; ++w)
        for (int wps : {1, 2, 4, 8}) {
            k_chain<<<ncu * wps, 256>>>(out, ticks, 1u, w);
            CK(hipEventRecord(e0));
            k_chain<<<ncu * wps, 256>>>(out, ticks, 1u, w);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            report(cn[w], wps, ms, 16);
        }
    const char* ln[] = {"ds_read_u16", "ds_write_b16", "ds_read_b32", "ds_write_b32", "ds_read_b128", "ds_write_b128", "ds_read_b64", "ds_read_u16 rows"};
    const int lper[] = {16, 16, 16, 16, 8, 8, 16, 16};
    for (int w = 0; w < 8; ++w)
        for (int wps : {1, 2, 4}) {
            k_lds<<<ncu * wps, 256>>>(out, ticks, 3u, w);
            CK(hipEventRecord(e0));
            k_lds<<<ncu * wps, 256>>>(out, ticks, 3u, w);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            // LDS is per CU: report CU cycles per wave-instruction (4 SIMDs share the pipe)
            const double insts_per_cu = (double)ITERS * lper[w] * wps * 4;
            printf("%-22s waves/SIMD %d : %7.3f ms  %6.2f CU-cycles per wave-inst @2.4GHz\n", ln[w], wps, ms, ms * 1e6 / insts_per_cu * 2.4);
        }
    return 0;
}
