// pin_third_party.cpp -- generates tests/golden/third_party_pins.json from the REAL
// third-party libraries present in the ROCm image (rocRAND, rocThrust), host side only.
// TEST INFRASTRUCTURE (oracle/): run once in the build container by `make -C oracle pins`;
// the JSON it prints is committed, this program never runs on the GPU box.
//
//  * rocRAND philox4x32_10 engine + rocrand_uniform4 (rocrand_philox4x32_10.h,
//    rocrand_uniform.h): pins the oracle's/kernel's Philox rounds, counter convention
//    (seed -> key, subsequence -> counter.zw, offset/4 -> counter.xy) and (0,1] mapping.
//  * rocThrust thrust::default_random_engine(17) + uniform_int_distribution<int>(0,2):
//    exactly the objects bvh.h:144-147 constructs and bvh.h:257 draws from.
//
// Build: hipcc --offload-arch=gfx950 -O1 pin_third_party.cpp -o pin_third_party  (no kernel is launched)
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_kernel.h>
#include <thrust/random.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main() {
    printf("{\n");
    // --- Philox blocks: (seed, pixel, sample, event) -> 4 raw words + 4 uniforms
    struct Q { unsigned long long seed; uint32_t pixel, sample, event; };
    const Q qs[] = {
        {0ull, 0, 0, 0}, {1ull, 0, 0, 0}, {1ull, 0, 0, 1}, {1ull, 1, 0, 0}, {1ull, 0, 1, 0},
        {1ull, 89999, 9, 3}, {0xdeadbeefcafef00dull, 809999, 499, 7}, {18ull, 8294399, 1023, 49},
        // sample < 2^30 so that (sample << 34) fits rocRAND's 64-bit offset
        {0xffffffffffffffffull, 0xffffffffu, 0x3fffffffu, 0xffffffffu},
    };
    printf("  \"philox_blocks\": [\n");
    const int nq = (int)(sizeof qs / sizeof qs[0]);
    for (int i = 0; i < nq; ++i) {
        const Q& q = qs[i];
        unsigned long long offset = ((unsigned long long)q.sample << 34) + 4ull * q.event;
        rocrand_state_philox4x32_10 a, b;
        rocrand_init(q.seed, (unsigned long long)q.pixel, offset, &a);
        rocrand_init(q.seed, (unsigned long long)q.pixel, offset, &b);
        uint4 raw = rocrand4(&a);
        float4 u = rocrand_uniform4(&b);
        printf("    {\"seed\": %llu, \"pixel\": %u, \"sample\": %u, \"event\": %u, \"raw\": [%u, %u, %u, %u], "
               "\"uniform_bits\": [%u, %u, %u, %u]}%s\n",
               q.seed, q.pixel, q.sample, q.event, raw.x, raw.y, raw.z, raw.w, bits(u.x), bits(u.y), bits(u.z),
               bits(u.w), i + 1 < nq ? "," : "");
    }
    printf("  ],\n");
    // --- sequential scene stream: seed 18, subsequence 1<<32 (counter.w = 1), first 64 uniforms
    {
        rocrand_state_philox4x32_10 s;
        rocrand_init(18ull, 1ull << 32, 0ull, &s);
        printf("  \"scene_stream_seed18_bits\": [");
        for (int i = 0; i < 64; ++i) printf("%u%s", bits(rocrand_uniform(&s)), i + 1 < 64 ? ", " : "");
        printf("],\n");
    }
    // --- uniform mapping edge cases through rocrand_device::detail::uniform_distribution
    {
        const uint32_t vs[] = {0u, 1u, 2u, 0x7fffffffu, 0x80000000u, 0xfffffeffu, 0xffffff7fu, 0xffffff80u, 0xffffffffu, 12345678u};
        printf("  \"uniform_edge\": [");
        const int nv = (int)(sizeof vs / sizeof vs[0]);
        for (int i = 0; i < nv; ++i)
            printf("[%u, %u]%s", vs[i], bits(rocrand_device::detail::uniform_distribution(vs[i])), i + 1 < nv ? ", " : "");
        printf("],\n");
    }
    // --- Thrust axis sequence (bvh.h:144-147, :257)
    {
        thrust::default_random_engine rng(17);
        thrust::uniform_int_distribution<int> dist(0, 2);
        printf("  \"thrust_axes_seed17\": \"");
        for (int i = 0; i < 4096; ++i) printf("%d", dist(rng));
        printf("\",\n");
        printf("  \"thrust_version\": %d\n", THRUST_VERSION);
    }
    printf("}\n");
    return 0;
}
