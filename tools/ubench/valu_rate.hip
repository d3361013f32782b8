// valu_rate.hip -- issue cost of a few VALU instructions on gfx950, in cycles per wave-instruction
// per SIMD, with W waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O2 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int OP>
__global__ __launch_bounds__(1024) void k(unsigned* out, int iters, unsigned long long* cycles) {
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 5, a5 = a0 + 7, a6 = a0 + 11, a7 = a0 + 13;
    unsigned long long w0 = a0, w1 = a1, w2 = a2, w3 = a3;
    const unsigned m = 0xD2511F53u;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %2, %2, %1\n v_xor_b32 %3, %3, %1\n v_xor_b32 %4, %4, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) :);) }
        if (OP == 1) { REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(a0), "v"(m) : "vcc");) }
        if (OP == 2) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) :);) }
        if (OP == 3) { REP16(asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) :);) }
        if (OP == 4) { REP16(asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) :);) }
        if (OP == 5) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) :);) }
        if (OP == 6) { REP16(asm volatile("v_mul_hi_u32_u24 %0, %0, %4\n v_mul_hi_u32_u24 %1, %1, %4\n v_mul_hi_u32_u24 %2, %2, %4\n v_mul_hi_u32_u24 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) :);) }
        if (OP == 7) { REP16(asm volatile("v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : :);) }
        if (OP == 8) { REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : :);) }
        if (OP == 9) { REP16(asm volatile("v_mad_u32_u24 %0, %0, %4, %0\n v_mad_u32_u24 %1, %1, %4, %1\n v_mad_u32_u24 %2, %2, %4, %2\n v_mad_u32_u24 %3, %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) :);) }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (unsigned)(w0 ^ w1 ^ w2 ^ w3);
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = (unsigned long long)(t1 - t0);
}

template <int OP>
void run(const char* name, unsigned* d_out, unsigned long long* d_cyc) {
    const int iters = 2000;
    for (int waves_per_simd : {1, 2, 4, 8}) {
        const int block = 64 * 4 * (waves_per_simd > 4 ? 4 : waves_per_simd);  // 4 SIMDs per CU
        const int blocks_per_cu = waves_per_simd > 4 ? waves_per_simd / 4 : 1;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(256 * blocks_per_cu), dim3(block), 0, 0, d_out, 10, d_cyc);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(256 * blocks_per_cu), dim3(block), 0, 0, d_out, iters, d_cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        // instructions per SIMD = waves_per_simd * iters * 64; time in ns / that, times 2.4 GHz
        const double insts = double(waves_per_simd) * iters * 64.0;
        printf("%-18s waves/SIMD %d: %.3f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", name, waves_per_simd,
               ms * 1e6 / insts, ms * 1e6 / insts * 2.4);
    }
}

int main() {
    unsigned* d_out; unsigned long long* d_cyc;
    hipMalloc((void**)&d_out, 256 * 8 * 1024 * 4);
    hipMalloc((void**)&d_cyc, 8);
    run<0>("v_xor_b32", d_out, d_cyc);
    run<5>("v_fma_f32", d_out, d_cyc);
    run<1>("v_mad_u64_u32", d_out, d_cyc);
    run<2>("v_mul_lo_u32", d_out, d_cyc);
    run<3>("v_mul_hi_u32", d_out, d_cyc);
    run<4>("v_mul_u32_u24", d_out, d_cyc);
    run<6>("v_mul_hi_u32_u24", d_out, d_cyc);
    run<9>("v_mad_u32_u24", d_out, d_cyc);
    run<7>("v_cvt_f32_u32", d_out, d_cyc);
    run<8>("v_rcp_f32", d_out, d_cyc);
    return 0;
}
