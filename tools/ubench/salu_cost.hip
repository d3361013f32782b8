// salu_cost.hip -- what does a scalar / EXEC-writing instruction cost a SIMD of gfx950 that is busy issuing VALU?
// Pattern: 8 x v_add_f32 (four chains) + X, all waves alike; reported: ns per pattern and X in units of one v_add_f32.
// GENERATED.  Build: hipcc --offload-arch=gfx950 -O2 salu_cost.hip -o salu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ __launch_bounds__(1024) void k(unsigned* out, int iters) {
    unsigned a0 = 0x3f800000u + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    const unsigned f = 0x3fc00000u;
    asm volatile("s_mov_b64 s[26:27], -1\n s_mov_b64 s[20:21], -1\n s_mov_b64 s[22:23], -1" ::: "s20", "s21", "s22", "s23", "s26", "s27");
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25");) }
        if (OP == 1) { REP16(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n s_nop 0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25");) }
        if (OP == 2) { REP16(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n s_and_b64 s[24:25], s[20:21], s[22:23]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25");) }
        if (OP == 3) { REP16(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n s_mov_b64 s[24:25], exec" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25");) }
        if (OP == 4) { REP16(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n s_mov_b64 exec, exec" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25");) }
        if (OP == 5) { REP16(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n s_or_b64 exec, exec, s[26:27]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25");) }
        if (OP == 6) { REP16(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n s_and_saveexec_b64 s[24:25], s[26:27]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25");) }
        if (OP == 7) { REP16(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n s_and_saveexec_b64 s[24:25], s[26:27]\n s_mov_b64 exec, s[24:25]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25");) }
        if (OP == 8) { REP16(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n s_and_saveexec_b64 s[24:25], s[26:27]\n s_mov_b64 exec, s[24:25]\n s_and_saveexec_b64 s[24:25], s[26:27]\n s_mov_b64 exec, s[24:25]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25");) }
        if (OP == 9) { REP16(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n s_bcnt1_i32_b64 s24, s[26:27]\n s_cmp_eq_u32 s24, 0\n s_cbranch_scc1 1f\n 1:" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25");) }
        if (OP == 10) { REP16(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n v_cmp_le_f32 s[20:21], 0, %4\n s_and_saveexec_b64 s[24:25], s[20:21]\n s_mov_b64 exec, s[24:25]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25");) }
        if (OP == 11) { REP16(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n s_cbranch_execz 1f\n 1:" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25");) }
        if (OP == 12) { REP16(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n s_and_b64 s[24:25], s[20:21], s[22:23]\n s_or_b64 s[20:21], s[24:25], s[22:23]\n s_and_b64 s[24:25], s[20:21], s[22:23]\n s_or_b64 s[20:21], s[24:25], s[22:23]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25");) }
        if (OP == 13) { REP16(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25");) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
}
static double base[3];
template <int OP>
void run(const char* name, unsigned* d_out) {
    const int iters = 500;
    printf("%-46s", name);
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
        const double pat = ms * 1e6 / (double(w) * iters * 16.0);  // ns per pattern per SIMD
        if (OP == 0) base[j] = pat;
        printf("  W=%d: %6.2f ns, X = %5.2f F", w, pat, (pat - base[j]) / (base[j] / 8));
        ++j;
    }
    printf("\n");
}
int main() {
    unsigned* d_out; (void)hipMalloc((void**)&d_out, 256 * 8 * 1024 * 4);
    run<0>("nothing (8 F)", d_out);
    run<1>("s_nop 0", d_out);
    run<2>("s_and_b64 sgpr", d_out);
    run<3>("s_mov_b64 sgpr <- exec", d_out);
    run<4>("s_mov_b64 exec <- exec", d_out);
    run<5>("s_or_b64 exec, exec, sgpr(=-1)", d_out);
    run<6>("s_and_saveexec_b64 (mask -1)", d_out);
    run<7>("saveexec + restore", d_out);
    run<8>("2 x (saveexec + restore)", d_out);
    run<9>("s_bcnt1 + s_cmp + s_cbranch (not taken)", d_out);
    run<10>("v_cmp -> sgpr, s_and_saveexec on it, restore", d_out);
    run<11>("s_cbranch_execz (not taken)", d_out);
    run<12>("4 x s_and_b64 sgpr", d_out);
    run<13>("s_waitcnt lgkmcnt(0)", d_out);
    return 0;
}
