// gather_quad.hip -- dependent random reads of 64-byte records (48 bytes used) from a table in
// L2 / Infinity Cache, the access shape of the BVH walk over a scene in HBM.
//   A: every lane reads its own record with three 16-byte loads (64 lines touched per load)
//   B: quad-cooperative: four loads, in load i the four lanes of a quad read the four 16-byte
//      pieces of the record of lane 16 i + quad (16 lines touched per load), written to a per-wave
//      LDS area, from which every lane reads its own record back
//   C: as B with LDS-DMA loads (global_load_lds_dwordx4), no staging registers
// Build: hipcc --offload-arch=gfx950 -O2 gather_quad.hip -o gather_quad
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned next_index(unsigned k, f4 a, f4 b, f4 c, unsigned mask) {
    const unsigned h = __float_as_uint(a.x) ^ __float_as_uint(b.y) ^ __float_as_uint(c.z);
    return (k * 2654435761u + h) & mask;
}

template <int MODE>
__global__ __launch_bounds__(768) void walk(const f4* __restrict__ table, unsigned mask, int iters, int active_pct, unsigned* out) {
    extern __shared__ f4 smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f4* stage = smem + 256 * wave;  // 64 records x 4 pieces
    unsigned k = (blockIdx.x * 768u + threadIdx.x) * 747796405u & mask;
    const bool active = (int)((lane * 37) % 100) < active_pct;  // lanes that walk at all
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        f4 a, b, c;
        if (MODE == 0) {
            if (active) {
                const f4* r = table + 4 * (size_t)k;
                a = r[0]; b = r[1]; c = r[2];
            }
        } else {
            for (int i = 0; i < 4; ++i) {
                const int owner = 16 * i + (lane >> 2);
                const unsigned ko = (unsigned)__builtin_amdgcn_ds_bpermute(4 * owner, (int)k);
                const bool own_active = (int)((owner * 37) % 100) < active_pct;
                if (own_active && (lane & 3) != 3) {
                    const f4* src = table + 4 * (size_t)ko + (lane & 3);
                    if (MODE == 1) {
                        stage[64 * i + lane] = *src;
                    } else {
#if defined(__HIP_DEVICE_COMPILE__)
                        typedef __attribute__((address_space(3))) void lds_void;
                        typedef __attribute__((address_space(1))) const void glb_void;
                        __builtin_amdgcn_global_load_lds((glb_void*)(uintptr_t)src, (lds_void*)(unsigned)(uintptr_t)(stage + 64 * i), 16, 0, 0);
#endif
                    }
                }
            }
            if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (active) { a = stage[4 * lane]; b = stage[4 * lane + 1]; c = stage[4 * lane + 2]; }
            __builtin_amdgcn_wave_barrier();
        }
        if (active) { k = next_index(k + (unsigned)it * 0x9E3779B9u + threadIdx.x * 0x85EBCA6Bu + blockIdx.x, a, b, c, mask); acc += k; }
    }
    out[blockIdx.x * 768 + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
    const unsigned records = 1u << 20;  // 64 MB
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    f4* d_table; unsigned* d_out;
    hipMalloc((void**)&d_table, size_t(records) * 64);
    hipMalloc((void**)&d_out, 512 * 768 * 4);
    std::vector<float> h(size_t(records) * 16);
    unsigned s = 12345;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = float(s >> 8) * 1e-3f; }
    hipMemcpy(d_table, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (unsigned recs : {256u, 16384u, 1u << 20})
    for (int pct : {100, 60}) {
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0);
                const size_t lds = 12 * 4096;
                if (mode == 0) hipLaunchKernelGGL(walk<0>, dim3(512), dim3(768), lds, 0, d_table, recs - 1, iters, pct, d_out);
                if (mode == 1) hipLaunchKernelGGL(walk<1>, dim3(512), dim3(768), lds, 0, d_table, recs - 1, iters, pct, d_out);
                if (mode == 2) hipLaunchKernelGGL(walk<2>, dim3(512), dim3(768), lds, 0, d_table, recs - 1, iters, pct, d_out);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            std::vector<unsigned> o(512 * 768);
            hipMemcpy(o.data(), d_out, o.size() * 4, hipMemcpyDeviceToHost);
            unsigned long long chk = 0; for (unsigned v : o) chk += v;
            const double steps = 512.0 * 768.0 * iters * pct / 100.0;
            printf("records %7u active %3d%% mode %c: %8.3f ms  %7.2f G record reads/s  checksum %llu\n", recs, pct, "ABC"[mode], best, steps / best / 1e6, chk);
        }
    }
    return 0;
}
