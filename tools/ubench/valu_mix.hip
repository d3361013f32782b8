// valu_mix.hip -- which VALU instructions of gfx950 overlap?  Patterns of 8 instructions (F = v_add_f32, S = v_max_f32,
// P/Q = v_pk_mul/add_f32, M = v_mad_u64_u32, B = v_bitop3_b32, R = v_rcp_f32, C = v_cmp, D = v_cndmask, X = v_fma_f64,
// L = v_lshl_add_u32, A = v_addc_co_u32, T = v_cvt_f32_u32), 128 instructions per loop iteration, W waves per SIMD.
// GENERATED.  Build: hipcc --offload-arch=gfx950 -O2 valu_mix.hip -o valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ __launch_bounds__(1024) void k(unsigned* out, int iters) {
    unsigned a0 = 0x3f800000u + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b0 = a0 + 4, b1 = a0 + 5, b2 = a0 + 6, b3 = a0 + 7;
    unsigned long long w0 = 0x3f8000003f800000ull + a0, w1 = w0 + 1, w2 = w0 + 2, w3 = w0 + 3;
    const unsigned f = 0x3fc00000u, m = 0xD2511F53u;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(f), "v"(m), "v"(w0), "v"(w1), "v"(w2), "v"(w3) : "vcc");) }
        if (OP == 1) { REP16(asm volatile("v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(f), "v"(m), "v"(w0), "v"(w1), "v"(w2), "v"(w3) : "vcc");) }
        if (OP == 2) { REP16(asm volatile("v_pk_mul_f32 %10, %10, %10\n v_pk_mul_f32 %11, %11, %11\n v_pk_mul_f32 %12, %12, %12\n v_pk_mul_f32 %13, %13, %13\n v_pk_mul_f32 %10, %10, %10\n v_pk_mul_f32 %11, %11, %11\n v_pk_mul_f32 %12, %12, %12\n v_pk_mul_f32 %13, %13, %13" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(f), "v"(m), "v"(w0), "v"(w1), "v"(w2), "v"(w3) : "vcc");) }
        if (OP == 3) { REP16(asm volatile("v_add_f32 %0, %0, %8\n v_max_f32 %4, %4, %8\n v_add_f32 %1, %1, %8\n v_max_f32 %5, %5, %8\n v_add_f32 %2, %2, %8\n v_max_f32 %6, %6, %8\n v_add_f32 %3, %3, %8\n v_max_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(f), "v"(m), "v"(w0), "v"(w1), "v"(w2), "v"(w3) : "vcc");) }
        if (OP == 4) { REP16(asm volatile("v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_add_f32 %0, %0, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n v_add_f32 %1, %1, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(f), "v"(m), "v"(w0), "v"(w1), "v"(w2), "v"(w3) : "vcc");) }
        if (OP == 5) { REP16(asm volatile("v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_add_f32 %0, %0, %8\n v_max_f32 %7, %7, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_add_f32 %1, %1, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(f), "v"(m), "v"(w0), "v"(w1), "v"(w2), "v"(w3) : "vcc");) }
        if (OP == 6) { REP16(asm volatile("v_add_f32 %0, %0, %8\n v_pk_mul_f32 %10, %10, %10\n v_add_f32 %1, %1, %8\n v_pk_mul_f32 %11, %11, %11\n v_add_f32 %2, %2, %8\n v_pk_mul_f32 %12, %12, %12\n v_add_f32 %3, %3, %8\n v_pk_mul_f32 %13, %13, %13" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(f), "v"(m), "v"(w0), "v"(w1), "v"(w2), "v"(w3) : "vcc");) }
        if (OP == 7) { REP16(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_pk_mul_f32 %10, %10, %10\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_pk_mul_f32 %11, %11, %11\n v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(f), "v"(m), "v"(w0), "v"(w1), "v"(w2), "v"(w3) : "vcc");) }
        if (OP == 8) { REP16(asm volatile("v_max_f32 %4, %4, %8\n v_pk_mul_f32 %10, %10, %10\n v_max_f32 %5, %5, %8\n v_pk_mul_f32 %11, %11, %11\n v_max_f32 %6, %6, %8\n v_pk_mul_f32 %12, %12, %12\n v_max_f32 %7, %7, %8\n v_pk_mul_f32 %13, %13, %13" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(f), "v"(m), "v"(w0), "v"(w1), "v"(w2), "v"(w3) : "vcc");) }
        if (OP == 9) { REP16(asm volatile("v_add_f32 %0, %0, %8\n v_pk_add_f32 %10, %10, %10\n v_add_f32 %1, %1, %8\n v_pk_mul_f32 %11, %11, %11\n v_add_f32 %2, %2, %8\n v_pk_add_f32 %12, %12, %12\n v_add_f32 %3, %3, %8\n v_pk_mul_f32 %13, %13, %13" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(f), "v"(m), "v"(w0), "v"(w1), "v"(w2), "v"(w3) : "vcc");) }
        if (OP == 10) { REP16(asm volatile("v_mad_u64_u32 %10, vcc, %8, %9, %10\n v_bitop3_b32 %0, %0, %8, %9 bitop3:0x96\n v_mad_u64_u32 %11, vcc, %8, %9, %11\n v_bitop3_b32 %1, %1, %8, %9 bitop3:0x96\n v_mad_u64_u32 %12, vcc, %8, %9, %12\n v_bitop3_b32 %2, %2, %8, %9 bitop3:0x96\n v_mad_u64_u32 %13, vcc, %8, %9, %13\n v_bitop3_b32 %3, %3, %8, %9 bitop3:0x96" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(f), "v"(m), "v"(w0), "v"(w1), "v"(w2), "v"(w3) : "vcc");) }
        if (OP == 11) { REP16(asm volatile("v_mad_u64_u32 %10, vcc, %8, %9, %10\n v_mad_u64_u32 %11, vcc, %8, %9, %11\n v_bitop3_b32 %0, %0, %8, %9 bitop3:0x96\n v_bitop3_b32 %1, %1, %8, %9 bitop3:0x96\n v_mad_u64_u32 %12, vcc, %8, %9, %12\n v_mad_u64_u32 %13, vcc, %8, %9, %13\n v_bitop3_b32 %2, %2, %8, %9 bitop3:0x96\n v_bitop3_b32 %3, %3, %8, %9 bitop3:0x96" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(f), "v"(m), "v"(w0), "v"(w1), "v"(w2), "v"(w3) : "vcc");) }
        if (OP == 12) { REP16(asm volatile("v_mad_u64_u32 %10, vcc, %8, %9, %10\n v_mad_u64_u32 %11, vcc, %8, %9, %11\n v_mad_u64_u32 %12, vcc, %8, %9, %12\n v_mad_u64_u32 %13, vcc, %8, %9, %13\n v_mad_u64_u32 %10, vcc, %8, %9, %10\n v_mad_u64_u32 %11, vcc, %8, %9, %11\n v_mad_u64_u32 %12, vcc, %8, %9, %12\n v_mad_u64_u32 %13, vcc, %8, %9, %13" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(f), "v"(m), "v"(w0), "v"(w1), "v"(w2), "v"(w3) : "vcc");) }
        if (OP == 13) { REP16(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_rcp_f32 %4, %4\n v_add_f32 %3, %3, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_rcp_f32 %5, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(f), "v"(m), "v"(w0), "v"(w1), "v"(w2), "v"(w3) : "vcc");) }
        if (OP == 14) { REP16(asm volatile("v_add_f32 %0, %0, %8\n v_rcp_f32 %4, %4\n v_add_f32 %1, %1, %8\n v_rcp_f32 %5, %5\n v_add_f32 %2, %2, %8\n v_rcp_f32 %6, %6\n v_add_f32 %3, %3, %8\n v_rcp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(f), "v"(m), "v"(w0), "v"(w1), "v"(w2), "v"(w3) : "vcc");) }
        if (OP == 15) { REP16(asm volatile("v_max_f32 %4, %4, %8\n v_rcp_f32 %5, %5\n v_max_f32 %6, %6, %8\n v_rcp_f32 %7, %7\n v_max_f32 %4, %4, %8\n v_rcp_f32 %5, %5\n v_max_f32 %6, %6, %8\n v_rcp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(f), "v"(m), "v"(w0), "v"(w1), "v"(w2), "v"(w3) : "vcc");) }
        if (OP == 16) { REP16(asm volatile("v_cmp_le_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, s[20:21]\n v_cmp_le_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, s[20:21]\n v_cmp_le_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, s[20:21]\n v_cmp_le_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(f), "v"(m), "v"(w0), "v"(w1), "v"(w2), "v"(w3) : "vcc");) }
        if (OP == 17) { REP16(asm volatile("v_add_f32 %0, %0, %8\n v_fma_f64 %10, %10, %10, %10\n v_add_f32 %1, %1, %8\n v_fma_f64 %11, %11, %11, %11\n v_add_f32 %2, %2, %8\n v_fma_f64 %12, %12, %12, %12\n v_add_f32 %3, %3, %8\n v_fma_f64 %13, %13, %13, %13" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(f), "v"(m), "v"(w0), "v"(w1), "v"(w2), "v"(w3) : "vcc");) }
        if (OP == 18) { REP16(asm volatile("v_fma_f64 %10, %10, %10, %10\n v_fma_f64 %11, %11, %11, %11\n v_fma_f64 %12, %12, %12, %12\n v_fma_f64 %13, %13, %13, %13\n v_fma_f64 %10, %10, %10, %10\n v_fma_f64 %11, %11, %11, %11\n v_fma_f64 %12, %12, %12, %12\n v_fma_f64 %13, %13, %13, %13" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(f), "v"(m), "v"(w0), "v"(w1), "v"(w2), "v"(w3) : "vcc");) }
        if (OP == 19) { REP16(asm volatile("v_add_f32 %0, %0, %8\n v_lshl_add_u32 %4, %4, 1, %9\n v_add_f32 %1, %1, %8\n v_lshl_add_u32 %5, %5, 1, %9\n v_add_f32 %2, %2, %8\n v_lshl_add_u32 %6, %6, 1, %9\n v_add_f32 %3, %3, %8\n v_lshl_add_u32 %7, %7, 1, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(f), "v"(m), "v"(w0), "v"(w1), "v"(w2), "v"(w3) : "vcc");) }
        if (OP == 20) { REP16(asm volatile("v_add_f32 %0, %0, %8\n v_addc_co_u32 %4, vcc, %4, %4, s[20:21]\n v_add_f32 %1, %1, %8\n v_addc_co_u32 %5, vcc, %5, %5, s[20:21]\n v_add_f32 %2, %2, %8\n v_addc_co_u32 %6, vcc, %6, %6, s[20:21]\n v_add_f32 %3, %3, %8\n v_addc_co_u32 %7, vcc, %7, %7, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(f), "v"(m), "v"(w0), "v"(w1), "v"(w2), "v"(w3) : "vcc");) }
        if (OP == 21) { REP16(asm volatile("v_add_f32 %0, %0, %8\n v_cvt_f32_u32 %4, %4\n v_add_f32 %1, %1, %8\n v_cvt_f32_u32 %5, %5\n v_add_f32 %2, %2, %8\n v_cvt_f32_u32 %6, %6\n v_add_f32 %3, %3, %8\n v_cvt_f32_u32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(f), "v"(m), "v"(w0), "v"(w1), "v"(w2), "v"(w3) : "vcc");) }
        if (OP == 22) { REP16(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_pk_mul_f32 %10, %10, %10\n v_max_f32 %4, %4, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_pk_mul_f32 %11, %11, %11\n v_max_f32 %5, %5, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(f), "v"(m), "v"(w0), "v"(w1), "v"(w2), "v"(w3) : "vcc");) }
        if (OP == 23) { REP16(asm volatile("v_add_f32 %0, %0, %8\n v_pk_mul_f32 %10, %10, %10\n v_max_f32 %4, %4, %8\n v_add_f32 %1, %1, %8\n v_pk_mul_f32 %11, %11, %11\n v_max_f32 %5, %5, %8\n v_add_f32 %2, %2, %8\n v_pk_mul_f32 %12, %12, %12" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(f), "v"(m), "v"(w0), "v"(w1), "v"(w2), "v"(w3) : "vcc");) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ b0 ^ b1 ^ b2 ^ b3 ^ (unsigned)(w0 ^ w1 ^ w2 ^ w3);
}
template <int OP>
void run(const char* name, unsigned* d_out) {
    const int iters = 500;
    printf("%-10s", name);
    for (int w : {4, 6, 8}) {
        const int block = w == 6 ? 768 : 64 * 4 * (w > 4 ? 4 : w);
        const int bpc = w > 4 ? 2 : 1;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(256 * bpc), dim3(block), 0, 0, d_out, 10);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(256 * bpc), dim3(block), 0, 0, d_out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        const double groups = double(w) * iters * 16.0;
        printf("  W=%d: %6.2f ns per 8 instructions", w, ms * 1e6 / groups);
    }
    printf("\n");
}
int main() {
    unsigned* d_out; (void)hipMalloc((void**)&d_out, 256 * 8 * 1024 * 4);
    run<0>("FFFFFFFF", d_out);
    run<1>("SSSSSSSS", d_out);
    run<2>("PPPPPPPP", d_out);
    run<3>("FSFSFSFS", d_out);
    run<4>("SSFSSFSS", d_out);
    run<5>("SSSFSSSF", d_out);
    run<6>("FPFPFPFP", d_out);
    run<7>("FFPFFPFF", d_out);
    run<8>("SPSPSPSP", d_out);
    run<9>("FQFPFQFP", d_out);
    run<10>("MBMBMBMB", d_out);
    run<11>("MMBBMMBB", d_out);
    run<12>("MMMMMMMM", d_out);
    run<13>("FFFRFFFR", d_out);
    run<14>("FRFRFRFR", d_out);
    run<15>("SRSRSRSR", d_out);
    run<16>("CDCDCDCD", d_out);
    run<17>("FXFXFXFX", d_out);
    run<18>("XXXXXXXX", d_out);
    run<19>("FLFLFLFL", d_out);
    run<20>("FAFAFAFA", d_out);
    run<21>("FTFTFTFT", d_out);
    run<22>("FFPSFFPS", d_out);
    run<23>("FPSFPSFP", d_out);
    return 0;
}
