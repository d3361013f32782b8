// exec_half.hip -- does gfx950 skip the pass of a wave64 VALU instruction whose 32-lane EXEC half is all zero?
// v_fma_f32 streams under four EXEC patterns: all 64 lanes, the low 32, the high 32, every other lane, one lane.
// If a half with no active lane were skipped, "low 32" / "high 32" / "one lane" would issue at twice the rate of "all"
// and "every other lane" would not.  Build: hipcc --offload-arch=gfx950 -O2 exec_half.hip -o exec_half
#include <hip/hip_runtime.h>

#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x

__global__ __launch_bounds__(1024) void k(unsigned* out, int iters, unsigned long long mask) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    const float m = 1.0001f;
    const int lane = threadIdx.x & 63;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; ++i) {
            REP16(asm volatile("v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) :);)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = __float_as_uint(a0 + a1 + a2 + a3);
}

int main() {
    unsigned* d_out;
    hipMalloc((void**)&d_out, 256 * 8 * 1024 * 4);
    const int iters = 4000;
    struct { const char* name; unsigned long long mask; } pats[] = {
        {"all 64 lanes", ~0ull}, {"low 32 lanes", 0xffffffffull}, {"high 32 lanes", 0xffffffff00000000ull},
        {"every other lane", 0x5555555555555555ull}, {"lane 0 only", 1ull}, {"lanes 0-15", 0xffffull}, {"lanes 0-15 + 32-47", 0x0000ffff0000ffffull}};
    for (int waves_per_simd : {1, 2, 6}) {
        for (auto& p : pats) {
            const int block = 64 * 4 * (waves_per_simd == 6 ? 3 : waves_per_simd);
            const int blocks_per_cu = waves_per_simd == 6 ? 2 : 1;
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k, dim3(256 * blocks_per_cu), dim3(block), 0, 0, d_out, 10, p.mask);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256 * blocks_per_cu), dim3(block), 0, 0, d_out, iters, p.mask);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double insts = double(waves_per_simd) * iters * 64.0;
            printf("waves/SIMD %d, %-20s %.3f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", waves_per_simd, p.name,
                   ms * 1e6 / insts, ms * 1e6 / insts * 2.4);
        }
    }
    return 0;
}
