// valu_rate.hip -- issue cost of VALU instructions on gfx950 (MI355X), in shader cycles per wave64 instruction per
// SIMD with 4 and 8 waves per SIMD.  GENERATED (the table of instruction forms lives in the generator that wrote this
// file, kept at its end as a comment); build: hipcc --offload-arch=gfx950 -O2 valu_rate.hip -o valu_rate
// Each test is four independent dependency chains of one instruction form, 64 instructions per loop iteration.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int OP>
__global__ __launch_bounds__(1024) void k(unsigned* out, int iters, unsigned long long* cycles) {
    unsigned a0 = 0x3f800000u + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    unsigned long long w0 = 0x3ff000003f800000ull + a0, w1 = w0 + 1, w2 = w0 + 2, w3 = w0 + 3;
    const unsigned m = 0xD2511F53u, f = 0x3fc00000u, sh = threadIdx.x & 1;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(asm volatile("v_fma_f32 %0, %0, %5, %1\n v_fma_f32 %1, %1, %5, %2\n v_fma_f32 %2, %2, %5, %3\n v_fma_f32 %3, %3, %5, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 1) { REP16(asm volatile("v_add_f32 %0, %0, %5\n v_add_f32 %1, %1, %5\n v_add_f32 %2, %2, %5\n v_add_f32 %3, %3, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 2) { REP16(asm volatile("v_sub_f32 %0, %0, %5\n v_sub_f32 %1, %1, %5\n v_sub_f32 %2, %2, %5\n v_sub_f32 %3, %3, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 3) { REP16(asm volatile("v_mul_f32 %0, %0, %5\n v_mul_f32 %1, %1, %5\n v_mul_f32 %2, %2, %5\n v_mul_f32 %3, %3, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 4) { REP16(asm volatile("v_mul_f32 %0, 0x3fc00000, %0\n v_mul_f32 %1, 0x3fc00000, %1\n v_mul_f32 %2, 0x3fc00000, %2\n v_mul_f32 %3, 0x3fc00000, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 5) { REP16(asm volatile("v_add_f32 %0, 1.0, %0\n v_add_f32 %1, 1.0, %1\n v_add_f32 %2, 1.0, %2\n v_add_f32 %3, 1.0, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 6) { REP16(asm volatile("v_fmac_f32 %0, %5, %5\n v_fmac_f32 %1, %5, %5\n v_fmac_f32 %2, %5, %5\n v_fmac_f32 %3, %5, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 7) { REP16(asm volatile("v_max_f32 %0, %0, %5\n v_max_f32 %1, %1, %5\n v_max_f32 %2, %2, %5\n v_max_f32 %3, %3, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 8) { REP16(asm volatile("v_min_f32 %0, %0, %5\n v_min_f32 %1, %1, %5\n v_min_f32 %2, %2, %5\n v_min_f32 %3, %3, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 9) { REP16(asm volatile("v_max3_f32 %0, %0, %5, %1\n v_max3_f32 %1, %1, %5, %2\n v_max3_f32 %2, %2, %5, %3\n v_max3_f32 %3, %3, %5, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 10) { REP16(asm volatile("v_med3_f32 %0, %0, %5, %1\n v_med3_f32 %1, %1, %5, %2\n v_med3_f32 %2, %2, %5, %3\n v_med3_f32 %3, %3, %5, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 11) { REP16(asm volatile("v_max_u32 %0, %0, %4\n v_max_u32 %1, %1, %4\n v_max_u32 %2, %2, %4\n v_max_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 12) { REP16(asm volatile("v_min_i32 %0, %0, %4\n v_min_i32 %1, %1, %4\n v_min_i32 %2, %2, %4\n v_min_i32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 13) { REP16(asm volatile("v_cmp_le_f32 vcc, %0, %5\n v_cmp_le_f32 vcc, %1, %5\n v_cmp_le_f32 vcc, %2, %5\n v_cmp_le_f32 vcc, %3, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : "vcc");) }
        if (OP == 14) { REP16(asm volatile("v_cmp_le_f32 s[20:21], %0, %5\n v_cmp_le_f32 s[20:21], %1, %5\n v_cmp_le_f32 s[20:21], %2, %5\n v_cmp_le_f32 s[20:21], %3, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : "s20", "s21");) }
        if (OP == 15) { REP16(asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_cmp_lt_u32 vcc, %1, %4\n v_cmp_lt_u32 vcc, %2, %4\n v_cmp_lt_u32 vcc, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : "vcc");) }
        if (OP == 16) { REP16(asm volatile("v_cmp_class_f32 vcc, %0, %4\n v_cmp_class_f32 vcc, %1, %4\n v_cmp_class_f32 vcc, %2, %4\n v_cmp_class_f32 vcc, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : "vcc");) }
        if (OP == 17) { REP16(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 18) { REP16(asm volatile("v_cndmask_b32 %0, %0, %4, s[20:21]\n v_cndmask_b32 %1, %1, %4, s[20:21]\n v_cndmask_b32 %2, %2, %4, s[20:21]\n v_cndmask_b32 %3, %3, %4, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 19) { REP16(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 20) { REP16(asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 21) { REP16(asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 22) { REP16(asm volatile("v_or_b32 %0, %0, %4\n v_or_b32 %1, %1, %4\n v_or_b32 %2, %2, %4\n v_or_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 23) { REP16(asm volatile("v_not_b32 %0, %0\n v_not_b32 %1, %1\n v_not_b32 %2, %2\n v_not_b32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 24) { REP16(asm volatile("v_bitop3_b32 %0, %0, %4, %1 bitop3:0x96\n v_bitop3_b32 %1, %1, %4, %2 bitop3:0x96\n v_bitop3_b32 %2, %2, %4, %3 bitop3:0x96\n v_bitop3_b32 %3, %3, %4, %0 bitop3:0x96" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 25) { REP16(asm volatile("v_bfi_b32 %0, %0, %4, %1\n v_bfi_b32 %1, %1, %4, %2\n v_bfi_b32 %2, %2, %4, %3\n v_bfi_b32 %3, %3, %4, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 26) { REP16(asm volatile("v_and_or_b32 %0, %0, %4, %1\n v_and_or_b32 %1, %1, %4, %2\n v_and_or_b32 %2, %2, %4, %3\n v_and_or_b32 %3, %3, %4, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 27) { REP16(asm volatile("v_or3_b32 %0, %0, %4, %1\n v_or3_b32 %1, %1, %4, %2\n v_or3_b32 %2, %2, %4, %3\n v_or3_b32 %3, %3, %4, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 28) { REP16(asm volatile("v_perm_b32 %0, %0, %4, %1\n v_perm_b32 %1, %1, %4, %2\n v_perm_b32 %2, %2, %4, %3\n v_perm_b32 %3, %3, %4, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 29) { REP16(asm volatile("v_alignbit_b32 %0, %0, %4, 5\n v_alignbit_b32 %1, %1, %4, 5\n v_alignbit_b32 %2, %2, %4, 5\n v_alignbit_b32 %3, %3, %4, 5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 30) { REP16(asm volatile("v_bfe_u32 %0, %0, 3, 9\n v_bfe_u32 %1, %1, 3, 9\n v_bfe_u32 %2, %2, 3, 9\n v_bfe_u32 %3, %3, 3, 9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 31) { REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 32) { REP16(asm volatile("v_add_u32 %0, -1, %0\n v_add_u32 %1, -1, %1\n v_add_u32 %2, -1, %2\n v_add_u32 %3, -1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 33) { REP16(asm volatile("v_sub_u32 %0, %0, %4\n v_sub_u32 %1, %1, %4\n v_sub_u32 %2, %2, %4\n v_sub_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 34) { REP16(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_add_co_u32 %1, vcc, %1, %4\n v_add_co_u32 %2, vcc, %2, %4\n v_add_co_u32 %3, vcc, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : "vcc");) }
        if (OP == 35) { REP16(asm volatile("v_addc_co_u32 %0, vcc, %0, %0, s[20:21]\n v_addc_co_u32 %1, vcc, %1, %1, s[20:21]\n v_addc_co_u32 %2, vcc, %2, %2, s[20:21]\n v_addc_co_u32 %3, vcc, %3, %3, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : "vcc");) }
        if (OP == 36) { REP16(asm volatile("v_add3_u32 %0, %0, %4, %1\n v_add3_u32 %1, %1, %4, %2\n v_add3_u32 %2, %2, %4, %3\n v_add3_u32 %3, %3, %4, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 37) { REP16(asm volatile("v_lshl_add_u32 %0, %0, 4, %1\n v_lshl_add_u32 %1, %1, 4, %2\n v_lshl_add_u32 %2, %2, 4, %3\n v_lshl_add_u32 %3, %3, 4, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 38) { REP16(asm volatile("v_add_lshl_u32 %0, %0, %1, 1\n v_add_lshl_u32 %1, %1, %2, 1\n v_add_lshl_u32 %2, %2, %3, 1\n v_add_lshl_u32 %3, %3, %0, 1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 39) { REP16(asm volatile("v_lshl_or_b32 %0, %0, 1, %1\n v_lshl_or_b32 %1, %1, 1, %2\n v_lshl_or_b32 %2, %2, 1, %3\n v_lshl_or_b32 %3, %3, 1, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 40) { REP16(asm volatile("v_xad_u32 %0, %0, %4, %1\n v_xad_u32 %1, %1, %4, %2\n v_xad_u32 %2, %2, %4, %3\n v_xad_u32 %3, %3, %4, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 41) { REP16(asm volatile("v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 42) { REP16(asm volatile("v_lshlrev_b32 %0, %6, %0\n v_lshlrev_b32 %1, %6, %1\n v_lshlrev_b32 %2, %6, %2\n v_lshlrev_b32 %3, %6, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 43) { REP16(asm volatile("v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n v_lshrrev_b32 %2, 1, %2\n v_lshrrev_b32 %3, 1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 44) { REP16(asm volatile("v_lshrrev_b32 %0, %6, %0\n v_lshrrev_b32 %1, %6, %1\n v_lshrrev_b32 %2, %6, %2\n v_lshrrev_b32 %3, %6, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 45) { REP16(asm volatile("v_ashrrev_i32 %0, 31, %0\n v_ashrrev_i32 %1, 31, %1\n v_ashrrev_i32 %2, 31, %2\n v_ashrrev_i32 %3, 31, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 46) { REP16(asm volatile("v_ffbl_b32 %0, %0\n v_ffbl_b32 %1, %1\n v_ffbl_b32 %2, %2\n v_ffbl_b32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 47) { REP16(asm volatile("v_ffbh_u32 %0, %0\n v_ffbh_u32 %1, %1\n v_ffbh_u32 %2, %2\n v_ffbh_u32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 48) { REP16(asm volatile("v_bcnt_u32_b32 %0, %0, %4\n v_bcnt_u32_b32 %1, %1, %4\n v_bcnt_u32_b32 %2, %2, %4\n v_bcnt_u32_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 49) { REP16(asm volatile("v_mbcnt_lo_u32_b32 %0, %0, %4\n v_mbcnt_lo_u32_b32 %1, %1, %4\n v_mbcnt_lo_u32_b32 %2, %2, %4\n v_mbcnt_lo_u32_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 50) { REP16(asm volatile("v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 51) { REP16(asm volatile("v_cvt_u32_f32 %0, %0\n v_cvt_u32_f32 %1, %1\n v_cvt_u32_f32 %2, %2\n v_cvt_u32_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 52) { REP16(asm volatile("v_trunc_f32 %0, %0\n v_trunc_f32 %1, %1\n v_trunc_f32 %2, %2\n v_trunc_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 53) { REP16(asm volatile("v_floor_f32 %0, %0\n v_floor_f32 %1, %1\n v_floor_f32 %2, %2\n v_floor_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 54) { REP16(asm volatile("v_ldexp_f32 %0, %0, %6\n v_ldexp_f32 %1, %1, %6\n v_ldexp_f32 %2, %2, %6\n v_ldexp_f32 %3, %3, %6" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 55) { REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 56) { REP16(asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 57) { REP16(asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 58) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 59) { REP16(asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 60) { REP16(asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 61) { REP16(asm volatile("v_mad_u32_u24 %0, %0, %4, %1\n v_mad_u32_u24 %1, %1, %4, %2\n v_mad_u32_u24 %2, %2, %4, %3\n v_mad_u32_u24 %3, %3, %4, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 62) { REP16(asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 63) { REP16(asm volatile("v_add_f32_dpp %0, %0, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %5 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : );) }
        if (OP == 64) { REP16(asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s20, %1, 3\n v_readlane_b32 s20, %2, 3\n v_readlane_b32 s20, %3, 3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(f), "v"(sh) : "s20");) }
        if (OP == 65) { REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %8, %0\n v_mad_u64_u32 %1, vcc, %4, %8, %1\n v_mad_u64_u32 %2, vcc, %4, %8, %2\n v_mad_u64_u32 %3, vcc, %4, %8, %3" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(a0) : "v"(m), "v"(f), "v"(sh), "v"(a1) : "vcc");) }
        if (OP == 66) { REP16(asm volatile("v_lshrrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 1, %1\n v_lshrrev_b64 %2, 1, %2\n v_lshrrev_b64 %3, 1, %3" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(a0) : "v"(m), "v"(f), "v"(sh), "v"(a1) : );) }
        if (OP == 67) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1\n v_pk_mul_f32 %2, %2, %2\n v_pk_mul_f32 %3, %3, %3" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(a0) : "v"(m), "v"(f), "v"(sh), "v"(a1) : );) }
        if (OP == 68) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(a0) : "v"(m), "v"(f), "v"(sh), "v"(a1) : );) }
        if (OP == 69) { REP16(asm volatile("v_pk_add_f32 %0, %0, %0\n v_pk_add_f32 %1, %1, %1\n v_pk_add_f32 %2, %2, %2\n v_pk_add_f32 %3, %3, %3" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(a0) : "v"(m), "v"(f), "v"(sh), "v"(a1) : );) }
        if (OP == 70) { REP16(asm volatile("v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(a0) : "v"(m), "v"(f), "v"(sh), "v"(a1) : );) }
        if (OP == 71) { REP16(asm volatile("v_add_f64 %0, %0, %0\n v_add_f64 %1, %1, %1\n v_add_f64 %2, %2, %2\n v_add_f64 %3, %3, %3" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(a0) : "v"(m), "v"(f), "v"(sh), "v"(a1) : );) }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ (unsigned)(w0 ^ w1 ^ w2 ^ w3);
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = (unsigned long long)(t1 - t0);
}

template <int OP>
void run(const char* name, unsigned* d_out, unsigned long long* d_cyc) {
    const int iters = 1000;
    printf("%-26s", name);
    for (int waves_per_simd : {4, 8}) {
        const int block = 64 * 4 * (waves_per_simd > 4 ? 4 : waves_per_simd);  // 4 SIMDs per CU
        const int blocks_per_cu = waves_per_simd > 4 ? waves_per_simd / 4 : 1;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(256 * blocks_per_cu), dim3(block), 0, 0, d_out, 10, d_cyc);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(256 * blocks_per_cu), dim3(block), 0, 0, d_out, iters, d_cyc);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long cyc = 0;  // shader cycles of wave 0 for the whole loop: cycles per instruction per SIMD = cyc / (W * iters * 64)
        (void)hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
        const double insts = double(waves_per_simd) * iters * 64.0;
        printf("  W=%d: %6.3f ns %5.2f cyc", waves_per_simd, ms * 1e6 / insts, double(cyc) / insts);
    }
    printf("\n");
}

int main() {
    unsigned* d_out; unsigned long long* d_cyc;
    (void)hipMalloc((void**)&d_out, 256 * 8 * 1024 * 4);
    (void)hipMalloc((void**)&d_cyc, 8);
    run<0>("v_fma_f32", d_out, d_cyc);
    run<1>("v_add_f32", d_out, d_cyc);
    run<2>("v_sub_f32", d_out, d_cyc);
    run<3>("v_mul_f32", d_out, d_cyc);
    run<4>("v_mul_f32 (literal)", d_out, d_cyc);
    run<5>("v_add_f32 (inline 1.0)", d_out, d_cyc);
    run<6>("v_fmac_f32", d_out, d_cyc);
    run<7>("v_max_f32", d_out, d_cyc);
    run<8>("v_min_f32", d_out, d_cyc);
    run<9>("v_max3_f32", d_out, d_cyc);
    run<10>("v_med3_f32", d_out, d_cyc);
    run<11>("v_max_u32", d_out, d_cyc);
    run<12>("v_min_i32", d_out, d_cyc);
    run<13>("v_cmp_le_f32 vcc", d_out, d_cyc);
    run<14>("v_cmp_le_f32 sgpr", d_out, d_cyc);
    run<15>("v_cmp_lt_u32 vcc", d_out, d_cyc);
    run<16>("v_cmp_class_f32", d_out, d_cyc);
    run<17>("v_cndmask_b32 vcc", d_out, d_cyc);
    run<18>("v_cndmask_b32 sgpr", d_out, d_cyc);
    run<19>("v_mov_b32", d_out, d_cyc);
    run<20>("v_xor_b32", d_out, d_cyc);
    run<21>("v_and_b32", d_out, d_cyc);
    run<22>("v_or_b32", d_out, d_cyc);
    run<23>("v_not_b32", d_out, d_cyc);
    run<24>("v_bitop3_b32", d_out, d_cyc);
    run<25>("v_bfi_b32", d_out, d_cyc);
    run<26>("v_and_or_b32", d_out, d_cyc);
    run<27>("v_or3_b32", d_out, d_cyc);
    run<28>("v_perm_b32", d_out, d_cyc);
    run<29>("v_alignbit_b32", d_out, d_cyc);
    run<30>("v_bfe_u32", d_out, d_cyc);
    run<31>("v_add_u32", d_out, d_cyc);
    run<32>("v_add_u32 (inline)", d_out, d_cyc);
    run<33>("v_sub_u32", d_out, d_cyc);
    run<34>("v_add_co_u32", d_out, d_cyc);
    run<35>("v_addc_co_u32", d_out, d_cyc);
    run<36>("v_add3_u32", d_out, d_cyc);
    run<37>("v_lshl_add_u32", d_out, d_cyc);
    run<38>("v_add_lshl_u32", d_out, d_cyc);
    run<39>("v_lshl_or_b32", d_out, d_cyc);
    run<40>("v_xad_u32", d_out, d_cyc);
    run<41>("v_lshlrev_b32 (inline)", d_out, d_cyc);
    run<42>("v_lshlrev_b32 (vgpr)", d_out, d_cyc);
    run<43>("v_lshrrev_b32 (inline)", d_out, d_cyc);
    run<44>("v_lshrrev_b32 (vgpr)", d_out, d_cyc);
    run<45>("v_ashrrev_i32 (inline)", d_out, d_cyc);
    run<46>("v_ffbl_b32", d_out, d_cyc);
    run<47>("v_ffbh_u32", d_out, d_cyc);
    run<48>("v_bcnt_u32_b32", d_out, d_cyc);
    run<49>("v_mbcnt_lo_u32_b32", d_out, d_cyc);
    run<50>("v_cvt_f32_u32", d_out, d_cyc);
    run<51>("v_cvt_u32_f32", d_out, d_cyc);
    run<52>("v_trunc_f32", d_out, d_cyc);
    run<53>("v_floor_f32", d_out, d_cyc);
    run<54>("v_ldexp_f32", d_out, d_cyc);
    run<55>("v_rcp_f32", d_out, d_cyc);
    run<56>("v_rsq_f32", d_out, d_cyc);
    run<57>("v_sqrt_f32", d_out, d_cyc);
    run<58>("v_mul_lo_u32", d_out, d_cyc);
    run<59>("v_mul_hi_u32", d_out, d_cyc);
    run<60>("v_mul_u32_u24", d_out, d_cyc);
    run<61>("v_mad_u32_u24", d_out, d_cyc);
    run<62>("v_mov_b32 dpp row_shr:1", d_out, d_cyc);
    run<63>("v_add_f32 dpp", d_out, d_cyc);
    run<64>("v_readlane_b32", d_out, d_cyc);
    run<65>("v_mad_u64_u32", d_out, d_cyc);
    run<66>("v_lshrrev_b64", d_out, d_cyc);
    run<67>("v_pk_mul_f32", d_out, d_cyc);
    run<68>("v_pk_fma_f32", d_out, d_cyc);
    run<69>("v_pk_add_f32", d_out, d_cyc);
    run<70>("v_fma_f64", d_out, d_cyc);
    run<71>("v_add_f64", d_out, d_cyc);
    return 0;
}
