// valu_pair.hip -- the cost of ONE instruction X when it alternates with v_add_f32 in a wave's stream on gfx950
// (F X F X F X F X; reported: ns per F+X pair minus the cost of the F, 6 and 8 waves per SIMD).  GENERATED.
// Build: hipcc --offload-arch=gfx950 -O2 valu_pair.hip -o valu_pair
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ __launch_bounds__(1024) void k(unsigned* out, int iters) {
    unsigned a0 = 0x3f800000u + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b0 = a0 + 4, b1 = a0 + 5, b2 = a0 + 6, b3 = a0 + 7;
    unsigned long long w0 = 0x3f8000003f800000ull + a0, w1 = w0 + 1, w2 = w0 + 2, w3 = w0 + 3;
    const unsigned f = 0x3fc00000u, m = 0xD2511F53u;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(asm volatile("v_add_f32 %0, %0, %12\n v_add_f32 %4, %4, %12\n v_add_f32 %1, %1, %12\n v_add_f32 %5, %5, %12\n v_add_f32 %2, %2, %12\n v_add_f32 %6, %6, %12\n v_add_f32 %3, %3, %12\n v_add_f32 %7, %7, %12" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 1) { REP16(asm volatile("v_add_f32 %0, %0, %12\n v_max_f32 %4, %4, %12\n v_add_f32 %1, %1, %12\n v_max_f32 %5, %5, %12\n v_add_f32 %2, %2, %12\n v_max_f32 %6, %6, %12\n v_add_f32 %3, %3, %12\n v_max_f32 %7, %7, %12" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 2) { REP16(asm volatile("v_add_f32 %0, %0, %12\n v_max3_f32 %4, %4, %12, %13\n v_add_f32 %1, %1, %12\n v_max3_f32 %5, %5, %12, %13\n v_add_f32 %2, %2, %12\n v_max3_f32 %6, %6, %12, %13\n v_add_f32 %3, %3, %12\n v_max3_f32 %7, %7, %12, %13" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 3) { REP16(asm volatile("v_add_f32 %0, %0, %12\n v_min3_f32 %4, %4, %12, s22\n v_add_f32 %1, %1, %12\n v_min3_f32 %5, %5, %12, s22\n v_add_f32 %2, %2, %12\n v_min3_f32 %6, %6, %12, s22\n v_add_f32 %3, %3, %12\n v_min3_f32 %7, %7, %12, s22" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 4) { REP16(asm volatile("v_add_f32 %0, %0, %12\n v_cmp_le_f32 vcc, %4, %12\n v_add_f32 %1, %1, %12\n v_cmp_le_f32 vcc, %5, %12\n v_add_f32 %2, %2, %12\n v_cmp_le_f32 vcc, %6, %12\n v_add_f32 %3, %3, %12\n v_cmp_le_f32 vcc, %7, %12" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 5) { REP16(asm volatile("v_add_f32 %0, %0, %12\n v_cmp_le_f32 s[20:21], %4, %12\n v_add_f32 %1, %1, %12\n v_cmp_le_f32 s[20:21], %5, %12\n v_add_f32 %2, %2, %12\n v_cmp_le_f32 s[20:21], %6, %12\n v_add_f32 %3, %3, %12\n v_cmp_le_f32 s[20:21], %7, %12" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 6) { REP16(asm volatile("v_add_f32 %0, %0, %12\n v_cmp_le_f32 s[20:21], %4, %12\n v_add_f32 %1, %1, %12\n v_cmp_le_f32 s[22:23], %5, %12\n v_add_f32 %2, %2, %12\n v_cmp_le_f32 s[24:25], %6, %12\n v_add_f32 %3, %3, %12\n v_cmp_le_f32 s[26:27], %7, %12" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 7) { REP16(asm volatile("v_add_f32 %0, %0, %12\n v_cmpx_ge_f32 exec, %4, %4\n v_add_f32 %1, %1, %12\n v_cmpx_ge_f32 exec, %5, %5\n v_add_f32 %2, %2, %12\n v_cmpx_ge_f32 exec, %6, %6\n v_add_f32 %3, %3, %12\n v_cmpx_ge_f32 exec, %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 8) { REP16(asm volatile("v_add_f32 %0, %0, %12\n v_cndmask_b32 %4, %4, %12, s[20:21]\n v_add_f32 %1, %1, %12\n v_cndmask_b32 %5, %5, %12, s[20:21]\n v_add_f32 %2, %2, %12\n v_cndmask_b32 %6, %6, %12, s[20:21]\n v_add_f32 %3, %3, %12\n v_cndmask_b32 %7, %7, %12, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 9) { REP16(asm volatile("v_add_f32 %0, %0, %12\n v_cndmask_b32 %4, %4, %12, vcc\n v_add_f32 %1, %1, %12\n v_cndmask_b32 %5, %5, %12, vcc\n v_add_f32 %2, %2, %12\n v_cndmask_b32 %6, %6, %12, vcc\n v_add_f32 %3, %3, %12\n v_cndmask_b32 %7, %7, %12, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 10) { REP16(asm volatile("v_add_f32 %0, %0, %12\n v_addc_co_u32 %4, vcc, %4, %4, s[20:21]\n v_add_f32 %1, %1, %12\n v_addc_co_u32 %5, vcc, %5, %5, s[20:21]\n v_add_f32 %2, %2, %12\n v_addc_co_u32 %6, vcc, %6, %6, s[20:21]\n v_add_f32 %3, %3, %12\n v_addc_co_u32 %7, vcc, %7, %7, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 11) { REP16(asm volatile("v_add_f32 %0, %0, %12\n v_addc_co_u32 %4, vcc, %4, %4, vcc\n v_add_f32 %1, %1, %12\n v_addc_co_u32 %5, vcc, %5, %5, vcc\n v_add_f32 %2, %2, %12\n v_addc_co_u32 %6, vcc, %6, %6, vcc\n v_add_f32 %3, %3, %12\n v_addc_co_u32 %7, vcc, %7, %7, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 12) { REP16(asm volatile("v_add_f32 %0, %0, %12\n v_add_co_u32 %4, vcc, %4, %12\n v_add_f32 %1, %1, %12\n v_add_co_u32 %5, vcc, %5, %12\n v_add_f32 %2, %2, %12\n v_add_co_u32 %6, vcc, %6, %12\n v_add_f32 %3, %3, %12\n v_add_co_u32 %7, vcc, %7, %12" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 13) { REP16(asm volatile("v_add_f32 %0, %0, %12\n v_add_co_u32 %4, s[20:21], %4, %12\n v_add_f32 %1, %1, %12\n v_add_co_u32 %5, s[20:21], %5, %12\n v_add_f32 %2, %2, %12\n v_add_co_u32 %6, s[20:21], %6, %12\n v_add_f32 %3, %3, %12\n v_add_co_u32 %7, s[20:21], %7, %12" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 14) { REP16(asm volatile("v_add_f32 %0, %0, %12\n v_mad_u64_u32 %8, vcc, %12, %13, %8\n v_add_f32 %1, %1, %12\n v_mad_u64_u32 %9, vcc, %12, %13, %9\n v_add_f32 %2, %2, %12\n v_mad_u64_u32 %10, vcc, %12, %13, %10\n v_add_f32 %3, %3, %12\n v_mad_u64_u32 %11, vcc, %12, %13, %11" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 15) { REP16(asm volatile("v_add_f32 %0, %0, %12\n v_mad_u64_u32 %8, vcc, %12, %4, 0\n v_add_f32 %1, %1, %12\n v_mad_u64_u32 %9, vcc, %12, %5, 0\n v_add_f32 %2, %2, %12\n v_mad_u64_u32 %10, vcc, %12, %6, 0\n v_add_f32 %3, %3, %12\n v_mad_u64_u32 %11, vcc, %12, %7, 0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 16) { REP16(asm volatile("v_add_f32 %0, %0, %12\n v_mul_hi_u32 %4, %4, %13\n v_add_f32 %1, %1, %12\n v_mul_hi_u32 %5, %5, %13\n v_add_f32 %2, %2, %12\n v_mul_hi_u32 %6, %6, %13\n v_add_f32 %3, %3, %12\n v_mul_hi_u32 %7, %7, %13" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 17) { REP16(asm volatile("v_add_f32 %0, %0, %12\n v_mul_lo_u32 %4, %4, %13\n v_add_f32 %1, %1, %12\n v_mul_lo_u32 %5, %5, %13\n v_add_f32 %2, %2, %12\n v_mul_lo_u32 %6, %6, %13\n v_add_f32 %3, %3, %12\n v_mul_lo_u32 %7, %7, %13" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 18) { REP16(asm volatile("v_add_f32 %0, %0, %12\n v_lshl_add_u32 %4, %4, 1, %13\n v_add_f32 %1, %1, %12\n v_lshl_add_u32 %5, %5, 1, %13\n v_add_f32 %2, %2, %12\n v_lshl_add_u32 %6, %6, 1, %13\n v_add_f32 %3, %3, %12\n v_lshl_add_u32 %7, %7, 1, %13" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 19) { REP16(asm volatile("v_add_f32 %0, %0, %12\n v_lshlrev_b32 %4, 4, %4\n v_add_f32 %1, %1, %12\n v_lshlrev_b32 %5, 4, %5\n v_add_f32 %2, %2, %12\n v_lshlrev_b32 %6, 4, %6\n v_add_f32 %3, %3, %12\n v_lshlrev_b32 %7, 4, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 20) { REP16(asm volatile("v_add_f32 %0, %0, %12\n v_ffbl_b32 %4, %4\n v_add_f32 %1, %1, %12\n v_ffbl_b32 %5, %5\n v_add_f32 %2, %2, %12\n v_ffbl_b32 %6, %6\n v_add_f32 %3, %3, %12\n v_ffbl_b32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 21) { REP16(asm volatile("v_add_f32 %0, %0, %12\n v_bfe_u32 %4, %4, 3, 9\n v_add_f32 %1, %1, %12\n v_bfe_u32 %5, %5, 3, 9\n v_add_f32 %2, %2, %12\n v_bfe_u32 %6, %6, 3, 9\n v_add_f32 %3, %3, %12\n v_bfe_u32 %7, %7, 3, 9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 22) { REP16(asm volatile("v_add_f32 %0, %0, %12\n v_bfi_b32 %4, %4, %12, %13\n v_add_f32 %1, %1, %12\n v_bfi_b32 %5, %5, %12, %13\n v_add_f32 %2, %2, %12\n v_bfi_b32 %6, %6, %12, %13\n v_add_f32 %3, %3, %12\n v_bfi_b32 %7, %7, %12, %13" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 23) { REP16(asm volatile("v_add_f32 %0, %0, %12\n v_bitop3_b32 %4, %4, %12, %13 bitop3:0x96\n v_add_f32 %1, %1, %12\n v_bitop3_b32 %5, %5, %12, %13 bitop3:0x96\n v_add_f32 %2, %2, %12\n v_bitop3_b32 %6, %6, %12, %13 bitop3:0x96\n v_add_f32 %3, %3, %12\n v_bitop3_b32 %7, %7, %12, %13 bitop3:0x96" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 24) { REP16(asm volatile("v_add_f32 %0, %0, %12\n v_cvt_f32_u32 %4, %4\n v_add_f32 %1, %1, %12\n v_cvt_f32_u32 %5, %5\n v_add_f32 %2, %2, %12\n v_cvt_f32_u32 %6, %6\n v_add_f32 %3, %3, %12\n v_cvt_f32_u32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 25) { REP16(asm volatile("v_add_f32 %0, %0, %12\n v_mbcnt_lo_u32_b32 %4, %4, %12\n v_add_f32 %1, %1, %12\n v_mbcnt_lo_u32_b32 %5, %5, %12\n v_add_f32 %2, %2, %12\n v_mbcnt_lo_u32_b32 %6, %6, %12\n v_add_f32 %3, %3, %12\n v_mbcnt_lo_u32_b32 %7, %7, %12" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 26) { REP16(asm volatile("v_add_f32 %0, %0, %12\n v_mov_b32_dpp %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32 %1, %1, %12\n v_mov_b32_dpp %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32 %2, %2, %12\n v_mov_b32_dpp %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32 %3, %3, %12\n v_mov_b32_dpp %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 27) { REP16(asm volatile("v_add_f32 %0, %0, %12\n v_readlane_b32 s20, %4, 3\n v_add_f32 %1, %1, %12\n v_readlane_b32 s20, %5, 3\n v_add_f32 %2, %2, %12\n v_readlane_b32 s20, %6, 3\n v_add_f32 %3, %3, %12\n v_readlane_b32 s20, %7, 3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 28) { REP16(asm volatile("v_add_f32 %0, %0, %12\n v_readfirstlane_b32 s20, %4\n v_add_f32 %1, %1, %12\n v_readfirstlane_b32 s20, %5\n v_add_f32 %2, %2, %12\n v_readfirstlane_b32 s20, %6\n v_add_f32 %3, %3, %12\n v_readfirstlane_b32 s20, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 29) { REP16(asm volatile("v_add_f32 %0, %0, %12\n v_rcp_f32 %4, %4\n v_add_f32 %1, %1, %12\n v_rcp_f32 %5, %5\n v_add_f32 %2, %2, %12\n v_rcp_f32 %6, %6\n v_add_f32 %3, %3, %12\n v_rcp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 30) { REP16(asm volatile("v_add_f32 %0, %0, %12\n v_rsq_f32 %4, %4\n v_add_f32 %1, %1, %12\n v_rsq_f32 %5, %5\n v_add_f32 %2, %2, %12\n v_rsq_f32 %6, %6\n v_add_f32 %3, %3, %12\n v_rsq_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 31) { REP16(asm volatile("v_add_f32 %0, %0, %12\n v_fma_f32 %4, %4, s22, %12\n v_add_f32 %1, %1, %12\n v_fma_f32 %5, %5, s22, %12\n v_add_f32 %2, %2, %12\n v_fma_f32 %6, %6, s22, %12\n v_add_f32 %3, %3, %12\n v_fma_f32 %7, %7, s22, %12" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 32) { REP16(asm volatile("v_add_f32 %0, %0, %12\n v_sub_f32 %4, s22, %4\n v_add_f32 %1, %1, %12\n v_sub_f32 %5, s22, %5\n v_add_f32 %2, %2, %12\n v_sub_f32 %6, s22, %6\n v_add_f32 %3, %3, %12\n v_sub_f32 %7, s22, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 33) { REP16(asm volatile("v_add_f32 %0, %0, %12\n v_pk_mul_f32 %8, %8, %8\n v_add_f32 %1, %1, %12\n v_pk_mul_f32 %9, %9, %9\n v_add_f32 %2, %2, %12\n v_pk_mul_f32 %10, %10, %10\n v_add_f32 %3, %3, %12\n v_pk_mul_f32 %11, %11, %11" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 34) { REP16(asm volatile("v_add_f32 %0, %0, %12\n s_and_b64 s[24:25], s[20:21], s[22:23]\n v_add_f32 %1, %1, %12\n s_and_b64 s[24:25], s[20:21], s[22:23]\n v_add_f32 %2, %2, %12\n s_and_b64 s[24:25], s[20:21], s[22:23]\n v_add_f32 %3, %3, %12\n s_and_b64 s[24:25], s[20:21], s[22:23]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 35) { REP16(asm volatile("v_add_f32 %0, %0, %12\n s_mov_b64 s[24:25], exec\n s_mov_b64 exec, s[24:25]\n v_add_f32 %1, %1, %12\n s_mov_b64 s[24:25], exec\n s_mov_b64 exec, s[24:25]\n v_add_f32 %2, %2, %12\n s_mov_b64 s[24:25], exec\n s_mov_b64 exec, s[24:25]\n v_add_f32 %3, %3, %12\n s_mov_b64 s[24:25], exec\n s_mov_b64 exec, s[24:25]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 36) { REP16(asm volatile("v_add_f32 %0, %0, %12\n s_nop 0\n v_add_f32 %1, %1, %12\n s_nop 0\n v_add_f32 %2, %2, %12\n s_nop 0\n v_add_f32 %3, %3, %12\n s_nop 0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(f), "v"(m) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ b0 ^ b1 ^ b2 ^ b3 ^ (unsigned)(w0 ^ w1 ^ w2 ^ w3);
}
static double base[3];
template <int OP>
void run(const char* name, unsigned* d_out) {
    const int iters = 500;
    printf("%-38s", name);
    int j = 0;
    for (int w : {4, 6, 8}) {
        const int block = w == 6 ? 768 : 64 * 4 * (w > 4 ? 4 : w);
        const int bpc = w > 4 ? 2 : 1;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(256 * bpc), dim3(block), 0, 0, d_out, 10);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(256 * bpc), dim3(block), 0, 0, d_out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        const double pair = ms * 1e6 / (double(w) * iters * 64.0);  // ns per F + X
        if (OP == 0) base[j] = pair / 2;
        printf("  W=%d: pair %5.2f ns, X = %5.2f F", w, pair, (pair - base[j]) / base[j]);
        ++j;
    }
    printf("\n");
}
int main() {
    unsigned* d_out; (void)hipMalloc((void**)&d_out, 256 * 8 * 1024 * 4);
    run<0>("v_add_f32 (F itself)", d_out);
    run<1>("v_max_f32", d_out);
    run<2>("v_max3_f32", d_out);
    run<3>("v_min3_f32 sgpr src", d_out);
    run<4>("v_cmp_le_f32 vcc", d_out);
    run<5>("v_cmp_le_f32 s[20:21]", d_out);
    run<6>("v_cmp_le_f32 4 different sgpr pairs", d_out);
    run<7>("v_cmpx_le_f32 (exec)", d_out);
    run<8>("v_cndmask_b32 s[20:21]", d_out);
    run<9>("v_cndmask_b32 vcc", d_out);
    run<10>("v_addc_co_u32 (sgpr in, vcc out)", d_out);
    run<11>("v_addc_co_u32 (vcc in, vcc out)", d_out);
    run<12>("v_add_co_u32 (vcc out)", d_out);
    run<13>("v_add_co_u32 (sgpr out)", d_out);
    run<14>("v_mad_u64_u32", d_out);
    run<15>("v_mad_u64_u32 (add 0)", d_out);
    run<16>("v_mul_hi_u32", d_out);
    run<17>("v_mul_lo_u32", d_out);
    run<18>("v_lshl_add_u32", d_out);
    run<19>("v_lshlrev_b32", d_out);
    run<20>("v_ffbl_b32", d_out);
    run<21>("v_bfe_u32", d_out);
    run<22>("v_bfi_b32", d_out);
    run<23>("v_bitop3_b32", d_out);
    run<24>("v_cvt_f32_u32", d_out);
    run<25>("v_mbcnt_lo_u32_b32", d_out);
    run<26>("v_mov_b32 dpp", d_out);
    run<27>("v_readlane_b32", d_out);
    run<28>("v_readfirstlane_b32", d_out);
    run<29>("v_rcp_f32", d_out);
    run<30>("v_rsq_f32", d_out);
    run<31>("v_fma_f32 sgpr src", d_out);
    run<32>("v_sub_f32 sgpr src", d_out);
    run<33>("v_pk_mul_f32", d_out);
    run<34>("s_and_b64 (SALU)", d_out);
    run<35>("s_and_saveexec + restore", d_out);
    run<36>("s_nop 0", d_out);
    return 0;
}
