// wavefront_queue.hip -- the dominant extra cost of a generate/extend/shade ("wavefront") path tracer on MI355X,
// measured in isolation: the ray-state round trip through device memory with stream compaction between bounces.
//
// A wavefront design keeps every live path as a record in a queue: per bounce one kernel reads each record, does its
// work (here: nothing -- the traversal and shading arithmetic of the megakernel would come on top), decides whether
// the path survives and appends the survivors' records, compacted, to the next queue.  Record = what the persistent
// megakernel keeps in registers per path (crt_kernels.hip: origin 12, direction 12, time 4, throughput 12, sky term 4,
// pixel 4, sample 4, event 4, depth 4 = 60 B; plus the hit the extend kernel hands to the shade kernel, t 4 + object 4,
// and, when extend and shade are separate kernels, 1/direction 12 and the closest-hit interval) -- rounded up to seven
// float4 = 112 B as in DESIGN.md's pricing.
//
// Workload: C2 -- 1200 x 675 x 500 = 405 M paths, 2.852 segments per path, i.e. 1.155e9 records read in all; survival
// per bounce s = 1 - 1/2.852 = 0.649 (geometric stand-in for C2's path-length distribution), depth limit 50.
// Paths are processed in batches of B (the queues of one batch: 2 x B x 112 B); per batch one launch per bounce until
// the queue is empty or the depth limit is hit.  Variants:
//   fused  one kernel per bounce: read record, survive?, compact (ballot + one atomic per wave), write record
//   split  two kernels per bounce: "extend" reads the ray part (32 B) and writes the hit (16 B) in place; "shade" reads
//          the whole record and writes the survivors
// Output: total ms for 405 M paths and the effective GB/s, per batch size.
//   hipcc --offload-arch=gfx950 -O3 -o wavefront_queue wavefront_queue.hip && ./wavefront_queue
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int REC_F4 = 7;  // 112 B, AoS-of-float4 planes: plane p of record i at q[p * cap + i] (coalesced SoA of 16-B lanes)

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// generate: B fresh records
__global__ __launch_bounds__(256) void generate_kernel(f4* q, uint32_t cap, uint32_t n, uint32_t first_path) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float id = __uint_as_float(first_path + i);
#pragma unroll
    for (int p = 0; p < REC_F4; ++p) q[(size_t)p * cap + i] = f4{id, (float)p, 1.0f, 0.5f};
}

// one bounce, fused: read, decide, compact, write.  Compaction: ballot per wave, the waves' counts meet in LDS and the
// workgroup takes its slice of the output queue with ONE atomic (one atomic per wave on the single queue tail was
// measured 4x slower end to end: same-address atomics serialise in L2)
constexpr int WG = 1024;
__global__ __launch_bounds__(WG) void bounce_fused(const f4* in, f4* out, uint32_t cap, uint32_t n, uint32_t* out_count,
                                                   uint32_t depth, uint32_t survive_u32) {
    __shared__ uint32_t wave_base[WG / 64 + 1];
    const uint32_t i = blockIdx.x * (uint32_t)WG + threadIdx.x;
    bool alive = false;
    f4 r[REC_F4];
    if (i < n) {
#pragma unroll
        for (int p = 0; p < REC_F4; ++p) r[p] = in[(size_t)p * cap + i];
        alive = hash32(__float_as_uint(r[0].x) * 64u + depth) < survive_u32;
        r[1].y += 1.0f;  // "the new ray"
    }
    const unsigned long long m = __ballot(alive);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_base[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t sum = 0;
        for (int w = 0; w < WG / 64; ++w) { const uint32_t c = wave_base[w]; wave_base[w] = sum; sum += c; }
        wave_base[WG / 64] = sum ? atomicAdd(out_count, sum) : 0u;
    }
    __syncthreads();
    const uint32_t base = wave_base[WG / 64] + wave_base[wave];
    if (alive) {
        const uint32_t o = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
#pragma unroll
        for (int p = 0; p < REC_F4; ++p) out[(size_t)p * cap + o] = r[p];
    }
}

// split: extend reads the ray (planes 0, 1 = 32 B), writes the hit (plane 6 = 16 B)
__global__ __launch_bounds__(256) void bounce_extend(f4* q, uint32_t cap, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const f4 a = q[i], b = q[(size_t)cap + i];
    q[(size_t)6 * cap + i] = f4{a.x, b.y, a.z + b.w, 1.0f};
}

struct result { double ms; double gb; long launches; };

static result run(bool fused, uint32_t B, uint64_t total_paths, double survive, int max_depth, f4* q[2], uint32_t* d_count,
                  uint32_t* h_count) {
    const uint32_t thr = (uint32_t)(survive * 4294967296.0);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    double bytes = 0;
    long launches = 0;
    CHECK(hipEventRecord(e0, nullptr));
    for (uint64_t first = 0; first < total_paths; first += B) {
        uint32_t n = (uint32_t)((total_paths - first) < B ? (total_paths - first) : B);
        hipLaunchKernelGGL(generate_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, q[0], B, n, (uint32_t)first);
        bytes += 112.0 * n; ++launches;
        int cur = 0;
        for (int depth = 0; depth < max_depth && n > 0; ++depth) {
            CHECK(hipMemsetAsync(d_count, 0, 4, nullptr));
            if (!fused) {
                hipLaunchKernelGGL(bounce_extend, dim3((n + 255) / 256), dim3(256), 0, nullptr, q[cur], B, n);
                bytes += 48.0 * n; ++launches;
            }
            hipLaunchKernelGGL(bounce_fused, dim3((n + WG - 1) / WG), dim3(WG), 0, nullptr, q[cur], q[cur ^ 1], B, n, d_count,
                               (uint32_t)depth, thr);
            ++launches;
            // the host needs the survivor count to size the next launch (a real implementation would do the same, or
            // launch indirectly / persistently; the read-back is part of the cost of the queue design)
            CHECK(hipMemcpyAsync(h_count, d_count, 4, hipMemcpyDeviceToHost, nullptr));
            CHECK(hipStreamSynchronize(nullptr));
            bytes += 112.0 * n + 112.0 * *h_count;
            n = *h_count;
            cur ^= 1;
        }
    }
    CHECK(hipEventRecord(e1, nullptr));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return {ms, bytes / 1e9, launches};
}

int main(int argc, char** argv) {
    const uint64_t total = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1200ull * 675 * 500;
    const double survive = 1.0 - 1.0 / 2.852;
    uint32_t* d_count; uint32_t* h_count;
    CHECK(hipMalloc((void**)&d_count, 4));
    CHECK(hipHostMalloc((void**)&h_count, 4));
    printf("# wavefront queue round trip: %llu paths, survival %.3f per bounce, 112-B records\n", (unsigned long long)total, survive);
    for (uint32_t B : {1u << 20, 1u << 22, 1u << 24, 1u << 26}) {
        f4* q[2];
        CHECK(hipMalloc((void**)&q[0], (size_t)B * 112));
        CHECK(hipMalloc((void**)&q[1], (size_t)B * 112));
        for (int fused = 1; fused >= 0; --fused) {
            run(fused, B, (uint64_t)B * 2 < total ? (uint64_t)B * 2 : total, survive, 50, q, d_count, h_count);  // warm-up
            const result r = run(fused, B, total, survive, 50, q, d_count, h_count);
            printf("batch %9u paths (%7.1f MB per queue)  %-5s  %9.2f ms  %8.1f GB moved  %7.0f GB/s  %6ld launches\n", B,
                   B * 112.0 / 1e6, fused ? "fused" : "split", r.ms, r.gb, r.gb / (r.ms * 1e-3), r.launches);
            fflush(stdout);
        }
        CHECK(hipFree(q[0])); CHECK(hipFree(q[1]));
    }
    return 0;
}
