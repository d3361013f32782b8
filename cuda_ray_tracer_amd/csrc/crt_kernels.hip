// crt_kernels.hip -- the hot path, hand-written for gfx950 (CDNA4, wave64).
//
// Persistent waves pull (8x8 tile, sample chunk) units from a global counter; lanes take (pixel,
// sample) items of their wave's unit and are scheduled wave-wide by traversal phase (see
// trace_bvh_scheduled).  Radiance is accumulated in 64-bit fixed point, so the image does not
// depend on scheduling, launch geometry or GPU count -- it is deterministic by construction.
//
// The BVH (sign-sorted box planes + leaf tags; 50 KB for the reference's random-spheres scene)
// is staged into LDS once per workgroup, spheres and materials are read through L1/L2; scenes
// that do not fit are read from HBM / Infinity Cache through L1/L2 by the same kernel.
//
// Arithmetic contract: every fp32 operation is one IEEE rounding in the order the
// reference's C++ expressions give (no contraction, correctly rounded / and sqrt), so a
// path is bit-identical to oracle/rt_oracle.cpp for the same Philox stream.
// References: render_with_cuda.h:6-56,116-168; helper_experimental.h:1-78; sphere.h:44-73;
// material.h:22-98; vec3.h:130-160; aabb.h:48-62.
#include <hip/hip_runtime.h>

#include <cmath>

#include <cstddef>

#include "crt_device.h"
#include "crt_launch.h"

// CRT_FAST=1 (second translation unit, crt_kernels_fast.o; selected per context with crt_set_arith): the same
// kernels with fp contraction allowed (-ffp-contract=fast) and the slab test re-associated to one fma per plane,
// t = fma(plane, 1/d, -(o * 1/d)).  Not bit-identical to the oracle; held to the tolerance stated in DESIGN.md.
#ifndef CRT_FAST
#define CRT_FAST 0
#endif
// Where rays with a non-finite reciprocal take their literal walk: 1 = where a lane enters the walking state (start of a
// segment, after a sphere test), 0 = in front of the hot walking loop.  Measured per build (alternating A/B): the exact
// build gains 0.36 % with 1 (71.97 vs 72.23 ms), the tolerance build loses 0.6 % (62.8 vs 62.4 ms).
#ifndef CRT_LITERAL_AT_ENTRY
#define CRT_LITERAL_AT_ENTRY (!CRT_FAST)
#endif
#if CRT_FAST
#pragma clang fp contract(fast)
#else
#pragma clang fp contract(off)
#endif

// (packed fp32 -- v_pk_add/mul_f32 -- for the slab products was measured 3-5% slower and is not used; the
// build also turns the SLP vectorizer off for the same reason, see the Makefile)

namespace {
// One inline namespace per arithmetic mode: the two translation units then emit DIFFERENT kernel symbols
// ((anonymous namespace)::arith_exact::render_kernel<...> / ...::arith_fast::...), so rocprofv3 --stats keeps the
// modes apart (with identical symbols it blended their dispatches into one row).
#if CRT_FAST
inline namespace arith_fast {
#else
inline namespace arith_exact {
#endif

struct v3 { float x, y, z; };

__device__ __forceinline__ v3 mk(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ v3 ld3(const float* p) { return mk(p[0], p[1], p[2]); }
__device__ __forceinline__ v3 operator+(v3 a, v3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 operator-(v3 a, v3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 operator-(v3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ v3 operator*(float t, v3 a) { return mk(t * a.x, t * a.y, t * a.z); }
__device__ __forceinline__ v3 operator*(v3 a, v3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
#if CRT_FAST
// contraction spelled out for the two primitives that the same logical value can reach through different code paths
// (a rejection-sampling attempt is judged in the shade phase's block loop or in its whole-wave tail pass, depending on
// scheduling): left to the compiler, the two sites could fuse differently and the frame would depend on scheduling
__device__ __forceinline__ float sum3(float xx, float y, float z) { return __builtin_fmaf(z, z, __builtin_fmaf(y, y, xx)); }
__device__ __forceinline__ float dot(v3 a, v3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ float len2(v3 a) { return sum3(a.x * a.x, a.y, a.z); }
#else
__device__ __forceinline__ float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float len2(v3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
#endif
// reciprocal, square root, reciprocal square root: correctly rounded in the exact build (the reference's 1/x, sqrtf);
// the hardware's 1-ulp v_rcp_f32 / v_sqrt_f32 / v_rsq_f32 in the fast build (one instruction instead of the ~10 of the
// correctly rounded sequences)
#if CRT_FAST
__device__ __forceinline__ float rcp_(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float rsqrt_(float x) { return __builtin_amdgcn_rsqf(x); }
#else
// 1 / x as the reference's division rounds it.  For 2^-100 <= |x| <= 2^100 the hardware's 1-ulp reciprocal and one
// Newton step in two fmas -- r0 + r0 (1 - x r0) -- give the correctly rounded quotient: 3 instructions where the
// compiler's division sequence (two v_div_scale, v_rcp, five fmas, v_div_fmas, v_div_fixup: built for every exponent)
// takes 11.  The general sequence runs only when some lane of the wave holds an x outside that range (0, inf, NaN,
// denormals, the far ends).  PROVEN for every one of the 2^32 arguments by comparison with the compiler's division on
// the device: crt_selftest_reciprocal, tests/test_gpu_round3.py.
__device__ __forceinline__ bool rcp_plain(float x) {  // false for NaN
    return __builtin_amdgcn_fmed3f(__builtin_fabsf(x), 0x1p-100f, 0x1p+100f) == __builtin_fabsf(x);
}
__device__ __forceinline__ float rcp_newton(float x) {
    const float r0 = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(__builtin_fmaf(-x, r0, 1.0f), r0, r0);
}
__device__ __forceinline__ float rcp_(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r = rcp_newton(x);
    const bool plain = rcp_plain(x);
    if (__builtin_expect(__ballot(!plain) != 0, 0)) {
        if (!plain) r = 1.0f / x;
    }
    return r;
#else
    return 1.0f / x;  // host pass of the single-source compile: never runs
#endif
}
// sqrtf(x) as the reference's rounds it.  For 2^-96 <= x < inf this is the compiler's own correctly rounded sequence
// (v_sqrt_f32, then the neighbours one ulp down and up tried by their fma residuals) WITHOUT its scaling of small
// arguments and its fix-up of 0 / inf: 9 instructions instead of 16; anything else -- 0, denormals, inf, negative, NaN --
// takes the general sequence, when some lane of the wave holds one.  PROVEN for all 2^32 arguments like rcp_().
__device__ __forceinline__ bool sqrt_plain(float x) {  // false for NaN and for x < 0
    return __builtin_amdgcn_fmed3f(x, 0x1p-96f, 0x1.fffffep+127f) == x;
}
__device__ __forceinline__ float sqrt_short(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float down = __uint_as_float(__float_as_uint(s) - 1u), up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_down = __builtin_fmaf(-down, s, x), r_up = __builtin_fmaf(-up, s, x);
    float r = (r_down <= 0.0f) ? down : s;
    r = (r_up > 0.0f) ? up : r;
    return r;
}
__device__ __forceinline__ float sqrt_(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r = sqrt_short(x);
    const bool plain = sqrt_plain(x);
    if (__builtin_expect(__ballot(!plain) != 0, 0)) {
        if (!plain) r = sqrtf(x);
    }
    return r;
#else
    return sqrtf(x);  // host pass of the single-source compile: never runs
#endif
}
// 1 / sqrtf(x): a plain x has its root in [2^-48, 2^64], which is plain for the reciprocal -- one range test for both
__device__ __forceinline__ float rsqrt_(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r = rcp_newton(sqrt_short(x));
    const bool plain = sqrt_plain(x);
    if (__builtin_expect(__ballot(!plain) != 0, 0)) {
        if (!plain) r = 1.0f / sqrtf(x);
    }
    return r;
#else
    return 1.0f / sqrtf(x);  // host pass of the single-source compile: never runs
#endif
}
#endif
// the three reciprocals of a direction, with one range test for the three
__device__ __forceinline__ v3 rcp3(v3 d) {
#if defined(__HIP_DEVICE_COMPILE__) && !CRT_FAST
    v3 r = mk(rcp_newton(d.x), rcp_newton(d.y), rcp_newton(d.z));
    const float ax = __builtin_fabsf(d.x), ay = __builtin_fabsf(d.y), az = __builtin_fabsf(d.z);
    const bool plain = (fminf(fminf(ax, ay), az) >= 0x1p-100f) & (fmaxf(fmaxf(ax, ay), az) <= 0x1p+100f);
    const bool ordered = (d.x == d.x) & (d.y == d.y) & (d.z == d.z);  // min / max drop a NaN: test for it
    if (__builtin_expect(__ballot(!(plain & ordered)) != 0, 0)) {
        if (!(plain & ordered)) r = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    }
    return r;
#else
    return mk(rcp_(d.x), rcp_(d.y), rcp_(d.z));
#endif
}
__device__ __forceinline__ v3 unit(v3 a) { return rsqrt_(len2(a)) * a; }  // vec3.h:114-116,130-132: v / t == (1/t) * v
// squared length of a rejection-sampling attempt (vec3.h:147-149,156-158: x*x + y*y + z*z, z = 0 for the disk)
__device__ __forceinline__ float lensq3(float x, float y, float z) { return len2(mk(x, y, z)); }

// ---- Philox4x32-10, counter = (event, sample, pixel, 0), key = seed -------------------------
// a ^ b ^ c as ONE instruction: gfx950's v_bitop3_b32 with the truth table of the three-input xor (the compiler
// emits two v_xor_b32 for the plain expression: 20 of the ~60 VALU instructions of a block)
__device__ __forceinline__ uint32_t xor3(uint32_t x, uint32_t y, uint32_t z) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_bitop3_b32((int)x, (int)y, (int)z, 0x96);
#else
    return x ^ y ^ z;  // host pass of the single-source compile: never runs
#endif
}
__device__ __forceinline__ uint4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1) {
#if defined(__HIP_DEVICE_COMPILE__)
    // The key is uniform, so the round keys are scalar adds.  Opaque here, they are formed where the block is drawn (18
    // s_add_u32); visible, the compiler forms all twenty once per kernel and keeps them in scalar registers across the
    // whole state machine, runs out of those and parks values in the lanes of a vector register -- v_readlane_b32 in
    // the block loop, each as dear as two fp32 instructions (tools/ubench/valu_pair.hip).
    asm volatile("" : "+s"(k0), "+s"(k1));
#endif
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long m0 = (unsigned long long)0xD2511F53u * c0;
        const unsigned long long m1 = (unsigned long long)0xCD9E8D57u * c2;
        const uint32_t n0 = xor3((uint32_t)(m1 >> 32), c1, k0);
        const uint32_t n2 = xor3((uint32_t)(m0 >> 32), c3, k1);
        c0 = n0; c1 = (uint32_t)m1; c2 = n2; c3 = (uint32_t)m0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}

// (0,1], rocRAND's mapping; v * 2^-32 is exact, so the sum is one rounding
// (as one fma: the product by a power of two is exact, so the fma rounds what the separate add rounds)
__device__ __forceinline__ float u01(uint32_t v) { return __builtin_fmaf((float)v, 2.3283064e-10f, 2.3283064e-10f); }

struct stream {
    uint32_t k0, k1, pixel, sample, event;
    __device__ __forceinline__ float4 next_block() {
        const uint4 r = philox4x32_10(event, sample, pixel, 0u, k0, k1);
        ++event;
        return make_float4(u01(r.x), u01(r.y), u01(r.z), u01(r.w));
    }
};

// random_float(state, -1, 1) = min + (max - min) * (1 - u), rtweekend.h:25-28
// (2 * x is exact, so fma(2, x, -1) is the separately rounded sum bit for bit)
__device__ __forceinline__ float rf11(float u) { return __builtin_fmaf(2.0f, 1.0f - u, -1.0f); }

// vec3.h:145-152 -- one Philox block per attempt
__device__ __forceinline__ v3 random_unit_vector(stream& rs) {
    for (;;) {
        const float4 b = rs.next_block();
        const v3 p = mk(rf11(b.x), rf11(b.y), rf11(b.z));
        const float lensq = len2(p);
        if (0.0f < lensq && lensq <= 1.0f) return rsqrt_(lensq) * p;
    }
}

// vec3.h:154-160 -- one block carries two attempts
__device__ __forceinline__ void random_in_unit_disk(stream& rs, float& dx, float& dy) {
    for (;;) {
        const float4 b = rs.next_block();
        float x = rf11(b.x), y = rf11(b.y);
        if (lensq3(x, y, 0.0f) < 1.0f) { dx = x; dy = y; return; }
        x = rf11(b.z); y = rf11(b.w);
        if (lensq3(x, y, 0.0f) < 1.0f) { dx = x; dy = y; return; }
    }
}

struct ray_t { v3 o, d; float tm; };

// render_with_cuda.h:6-29
__device__ __forceinline__ ray_t get_ray(stream& rs, const crt_dev_camera& cam, int pixel_x, int pixel_y) {
    const float4 b = rs.next_block();
    const float x_offset = b.x - 0.5f;
    const float y_offset = b.y - 0.5f;
    const v3 pixel_sample = (ld3(cam.p00) + (((float)pixel_x + x_offset) * ld3(cam.du))) +
                            (((float)pixel_y + y_offset) * ld3(cam.dv));
    float dx, dy;
    random_in_unit_disk(rs, dx, dy);  // always sampled
    const v3 p_in_disk = (ld3(cam.center) + dx * ld3(cam.ddu)) + dy * ld3(cam.ddv);
    ray_t r;
    r.o = (cam.defocus_angle <= 0.0f) ? ld3(cam.center) : p_in_disk;
    r.d = pixel_sample - r.o;
    r.tm = 1.0f - b.z;
    return r;
}

struct sphere_rec { v3 c, mo; float radius; int mat; };

typedef float f4 __attribute__((ext_vector_type(4)));

// The address space is spelled out (LDS or global): a pointer that may be either makes the compiler
// emit flat loads, which wait on both memory counters and cost about twice as much.
template <bool IN_LDS>
__device__ __forceinline__ sphere_rec load_sphere(const float4* spheres, int i) {
    f4 a, b;
#if defined(__HIP_DEVICE_COMPILE__)
    if (IN_LDS) {
        typedef __attribute__((address_space(3))) const f4 lds_f4;
        lds_f4* p = (lds_f4*)(uint32_t)(uintptr_t)spheres;  // low half of a flat LDS address = LDS offset
        a = p[2 * i]; b = p[2 * i + 1];
    } else {
        // (an unsigned 32-bit byte offset from the uniform base -- sphere indices are >= 0 and 32 n bytes < 4 GB, crt_upload_scene --
        // is one shift and the scalar-base form of the load; the sign-extended 64-bit index cost three 64-bit instructions)
        typedef __attribute__((address_space(1))) const char glb_char;
        typedef __attribute__((address_space(1))) const f4 glb_f4;
        glb_char* base = (glb_char*)(uintptr_t)spheres;
        const uint32_t off = (uint32_t)i << 5;
        a = *(glb_f4*)(base + off); b = *(glb_f4*)(base + off + 16u);
    }
#else
    a = b = f4{0, 0, 0, 0};  // host pass of the single-source compile: never runs
#endif
    sphere_rec s;
    s.c = mk(a.x, a.y, a.z); s.radius = a.w;
    s.mo = mk(b.x, b.y, b.z); s.mat = __float_as_int(b.w);
    return s;
}

// sphere.h:44-66: returns the accepted root or a negative value (roots are > tmin = 0.001)
__device__ __forceinline__ float sphere_root(const sphere_rec& s, const ray_t& r, float a, float tmin, float tmax) {
    const v3 cc = s.c + r.tm * s.mo;  // center.at(r.time())
    const v3 oc = cc - r.o;
    const float h = dot(r.d, oc);
    const float c = len2(oc) - s.radius * s.radius;
    const float disc = h * h - a * c;
    if (disc < 0.0f) return -1.0f;
    const float sq = sqrt_(disc);
#if CRT_FAST
    const float ia = rcp_(a);
    float root = (h - sq) * ia;
    if (!(tmin < root && root < tmax)) {
        root = (h + sq) * ia;
        if (!(tmin < root && root < tmax)) return -1.0f;
    }
    return root;
#else
    float root = (h - sq) / a;
    if (!(tmin < root && root < tmax)) {
        root = (h + sq) / a;
        if (!(tmin < root && root < tmax)) return -1.0f;
    }
    return root;
#endif
}

// aabb.h:48-62 for one box (the :41-46 guard is not reproduced)
__device__ __forceinline__ bool slab(float bx0, float bx1, float by0, float by1, float bz0, float bz1, v3 o, v3 inv,
                                     float lo, float hi) {
    float t0 = (bx0 - o.x) * inv.x, t1 = (bx1 - o.x) * inv.x;
    bool f = t0 < t1;
    float nr = f ? t0 : t1, fr = f ? t1 : t0;
    lo = nr > lo ? nr : lo; hi = fr < hi ? fr : hi;
    t0 = (by0 - o.y) * inv.y; t1 = (by1 - o.y) * inv.y;
    f = t0 < t1; nr = f ? t0 : t1; fr = f ? t1 : t0;
    lo = nr > lo ? nr : lo; hi = fr < hi ? fr : hi;
    t0 = (bz0 - o.z) * inv.z; t1 = (bz1 - o.z) * inv.z;
    f = t0 < t1; nr = f ? t0 : t1; fr = f ? t1 : t0;
    lo = nr > lo ? nr : lo; hi = fr < hi ? fr : hi;
    // lo only grows and hi only shrinks, so "max <= min after some axis" == "at the end"
    return !(hi <= lo);
}

// The same test with v_min/v_max: identical decisions whenever no product is NaN, i.e. whenever the
// three reciprocals are finite (then (b - o) * inv is finite or +-inf).  Rays with a non-finite
// reciprocal (a zero or denormal direction component) take slab() above.
__device__ __forceinline__ bool slab_fast(float bx0, float bx1, float by0, float by1, float bz0, float bz1, v3 o,
                                          v3 inv, v3 noi, float lo, float hi) {
#if CRT_FAST
    const float tx0 = __builtin_fmaf(bx0, inv.x, noi.x), tx1 = __builtin_fmaf(bx1, inv.x, noi.x);
    const float ty0 = __builtin_fmaf(by0, inv.y, noi.y), ty1 = __builtin_fmaf(by1, inv.y, noi.y);
    const float tz0 = __builtin_fmaf(bz0, inv.z, noi.z), tz1 = __builtin_fmaf(bz1, inv.z, noi.z);
#else
    const float tx0 = (bx0 - o.x) * inv.x, tx1 = (bx1 - o.x) * inv.x;
    const float ty0 = (by0 - o.y) * inv.y, ty1 = (by1 - o.y) * inv.y;
    const float tz0 = (bz0 - o.z) * inv.z, tz1 = (bz1 - o.z) * inv.z;
#endif
    lo = fmaxf(lo, fminf(tx0, tx1)); hi = fminf(hi, fmaxf(tx0, tx1));
    lo = fmaxf(lo, fminf(ty0, ty1)); hi = fminf(hi, fmaxf(ty0, ty1));
    lo = fmaxf(lo, fminf(tz0, tz1)); hi = fminf(hi, fmaxf(tz0, tz1));
    return !(hi <= lo);
}


// slab_fast's two interval ends (the caller compares them: successor_lohi)
__device__ __forceinline__ void slab_lohi(float bx0, float bx1, float by0, float by1, float bz0, float bz1, v3 o,
                                          v3 inv, v3 noi, float lo, float hi, float& lo_out, float& hi_out) {
#if CRT_FAST
    const float tx0 = __builtin_fmaf(bx0, inv.x, noi.x), tx1 = __builtin_fmaf(bx1, inv.x, noi.x);
    const float ty0 = __builtin_fmaf(by0, inv.y, noi.y), ty1 = __builtin_fmaf(by1, inv.y, noi.y);
    const float tz0 = __builtin_fmaf(bz0, inv.z, noi.z), tz1 = __builtin_fmaf(bz1, inv.z, noi.z);
#else
    const float tx0 = (bx0 - o.x) * inv.x, tx1 = (bx1 - o.x) * inv.x;
    const float ty0 = (by0 - o.y) * inv.y, ty1 = (by1 - o.y) * inv.y;
    const float tz0 = (bz0 - o.z) * inv.z, tz1 = (bz1 - o.z) * inv.z;
#endif
    lo = fmaxf(lo, fminf(tx0, tx1)); hi = fminf(hi, fmaxf(tx0, tx1));
    lo = fmaxf(lo, fminf(ty0, ty1)); hi = fminf(hi, fmaxf(ty0, ty1));
    lo = fmaxf(lo, fminf(tz0, tz1)); hi = fminf(hi, fmaxf(tz0, tz1));
    lo_out = lo; hi_out = hi;
}

// Successor of a traversal step (helper_experimental.h:28-45 over the implicit heap): descend -- left child first, the
// right one remembered in `pending` when both were hit -- or pop the nearest pending right sibling.
__device__ __forceinline__ void successor(uint32_t& k, uint32_t& pending, bool hitL, bool hitR) {
    const bool both = hitL && hitR, any = hitL || hitR;
    const uint32_t kd = (k << 1) | (hitL ? 0u : 1u), pd = (pending << 1) | (both ? 1u : 0u);
    const uint32_t up = (uint32_t)__builtin_ctz(pending);  // the sentinel keeps pending != 0
    const uint32_t kp = (k >> up) ^ 1u, pp = (pending >> up) ^ 1u;
    k = any ? kd : kp;
    pending = any ? pd : pp;
}

// The same successor from the interval ends of the two children (child hit <=> !(hi <= lo)), hand-scheduled: the
// compiler's form spends 13 vector instructions on it (two compares, two 0/1 materialisations, two shift-ors, ctz,
// two shifts, two xors, two selects) and 3 scalar ones; this one 9 and 5 -- the descent is k + k + carry (v_addc_co_u32 with the lane masks
// "left missed" / "both hit" as carry-in) and the select is done by the EXEC mask, which is restored before the block
// ends.  k and pending are updated in place.  Identical results to successor() for every input (tests).
__device__ __forceinline__ void successor_lohi(uint32_t& k, uint32_t& pending, float loL, float hiL, float loR, float hiR) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CRT_SUCCESSOR_PLAIN)
    uint32_t t0, t1;
    unsigned long long mL, m_any, saved;
    // mL / VCC: lane masks "left child MISSED" / "right child missed" (hi <= lo; false for NaN, like the C expression);
    // inactive lanes read 0 there, so `any` and `both` carry ones for them -- harmless, every use is under EXEC
    asm("v_ffbl_b32 %[t0], %[p]\n\t"
        "v_cmp_le_f32 %[mL], %[hiL], %[loL]\n\t"
        "v_cmp_le_f32 vcc, %[hiR], %[loR]\n\t"
        "v_lshrrev_b32 %[t1], %[t0], %[k]\n\t"
        "v_lshrrev_b32 %[t0], %[t0], %[p]\n\t"
        "s_nand_b64 %[any], %[mL], vcc\n\t"            // any child hit
        "s_nor_b64 vcc, %[mL], vcc\n\t"                // both hit
        "s_and_saveexec_b64 %[sv], %[any]\n\t"          // lanes that descend
        // (the carry comes in through VCC: with any other scalar pair as carry-in a v_addc_co_u32 occupies the SIMD
        // twice as long -- tools/ubench/valu_pair.hip -- which is worth the one s_mov_b64)
        "v_addc_co_u32 %[p], vcc, %[p], %[p], vcc\n\t"  // 2 pending + (both hit)
        "s_mov_b64 vcc, %[mL]\n\t"
        "v_addc_co_u32 %[k], vcc, %[k], %[k], vcc\n\t"  // 2k + (left missed)
        "s_andn2_b64 exec, %[sv], %[any]\n\t"           // lanes that pop
        "v_xor_b32 %[k], 1, %[t1]\n\t"
        "v_xor_b32 %[p], 1, %[t0]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [k] "+v"(k), [p] "+v"(pending), [t0] "=&v"(t0), [t1] "=&v"(t1), [mL] "=&s"(mL),
          [any] "=&s"(m_any), [sv] "=&s"(saved)
        : [loL] "v"(loL), [hiL] "v"(hiL), [loR] "v"(loR), [hiR] "v"(hiR)
        : "vcc");
#else
    successor(k, pending, !(hiL <= loL), !(hiR <= loR));
#endif
}

__device__ __forceinline__ bool finite3(v3 a) {
    return (fabsf(a.x) < __builtin_huge_valf()) & (fabsf(a.y) < __builtin_huge_valf()) & (fabsf(a.z) < __builtin_huge_valf());
}

struct hit_t { float t; int obj; };

// hittable_list::hit, hittable_list.h:43-58
template <bool IN_LDS, bool COUNT>
__device__ __forceinline__ hit_t closest_hit_list(const float4* spheres, int n, const ray_t& r, uint32_t& n_tests) {
    hit_t best; best.t = __builtin_huge_valf(); best.obj = -1;
    const float a = len2(r.d);
    for (int i = 0; i < n; ++i) {
        const sphere_rec s = load_sphere<IN_LDS>(spheres, i);
        const float root = sphere_root(s, r, a, 0.001f, best.t);
        if (root > 0.0f) { best.t = root; best.obj = i; }
    }
    if (COUNT) n_tests += (uint32_t)n;
    return best;
}

// external_hit, helper_experimental.h:1-50, over the implicit heap: left child first,
// right child remembered in a bit mask, both children tested against the CURRENT
// interval before descending.
template <bool IN_LDS, bool COUNT>
__device__ __forceinline__ hit_t closest_hit_bvh(const float4* pairs, const float4* spheres, int num_leaves,
                                                 const ray_t& r, uint32_t& n_pairs, uint32_t& n_tests) {
    hit_t best; best.t = __builtin_huge_valf(); best.obj = -1;
    const float a = len2(r.d);
    const float tmin = 0.001f;
    if (num_leaves == 1) {  // the root is the only leaf: no box is tested (helper_experimental.h:16)
        const sphere_rec s = load_sphere<IN_LDS>(spheres, __float_as_int(pairs[3].x));
        const float root = sphere_root(s, r, a, tmin, best.t);
        if (COUNT) n_tests += 1;
        if (root > 0.0f) { best.t = root; best.obj = __float_as_int(pairs[3].x); }
        return best;
    }
    const v3 inv = mk(rcp_(r.d.x), rcp_(r.d.y), rcp_(r.d.z));  // aabb.h:48
    const uint32_t leaf_parents = (uint32_t)num_leaves >> 1;
    uint32_t k = 1, pending = 0;
    for (;;) {
        const float4 q0 = pairs[4 * k], q1 = pairs[4 * k + 1], q2 = pairs[4 * k + 2], q3 = pairs[4 * k + 3];
        const int tagL = __float_as_int(q3.x), tagR = __float_as_int(q3.y);
        if (COUNT) n_pairs += 1;
        const bool hitL = (tagL != CRT_TAG_EMPTY) && slab(q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, r.o, inv, tmin, best.t);
        const bool hitR = (tagR != CRT_TAG_EMPTY) && slab(q0.z, q0.w, q1.z, q1.w, q2.z, q2.w, r.o, inv, tmin, best.t);
        bool pop;
        if (k >= leaf_parents) {  // children are leaves: left sphere, then right sphere
            if (hitL) {
                const float root = sphere_root(load_sphere<IN_LDS>(spheres, tagL), r, a, tmin, best.t);
                if (COUNT) n_tests += 1;
                if (root > 0.0f) { best.t = root; best.obj = tagL; }
            }
            if (hitR) {
                const float root = sphere_root(load_sphere<IN_LDS>(spheres, tagR), r, a, tmin, best.t);
                if (COUNT) n_tests += 1;
                if (root > 0.0f) { best.t = root; best.obj = tagR; }
            }
            pop = true;
        } else if (hitL) {
            pending = (pending << 1) | (hitR ? 1u : 0u);
            k = 2 * k;
            pop = false;
        } else if (hitR) {
            pending <<= 1;
            k = 2 * k + 1;
            pop = false;
        } else {
            pop = true;
        }
        if (pop) {
            if (pending == 0) break;
            const int up = __builtin_ctz(pending);
            k = (k >> up) ^ 1u;  // the pending right sibling, `up` levels above
            pending = (pending >> up) ^ 1u;
        }
    }
    return best;
}

// material.h:92-97, powf(x,5) as multiplies.  r0 = ((1 - ri) / (1 + ri))^2 depends on the material and the side only: the
// two values are in the material record (crt_upload_scene forms them with the same fp32 operations, correctly rounded on
// the host as on the device), which saves the division here.
__device__ __forceinline__ float reflectance(float cosine, float r0) {
    const float x = 1.0f - cosine;
    const float x2 = x * x;
    const float x4 = x2 * x2;
    const float x5 = x4 * x;
    return r0 + (1.0f - r0) * x5;
}

// One (pixel, sample) path state
struct path_t {
    ray_t r;
    v3 thr;        // attenuation_mult
    float sky_a;   // 0.5*(unit(primary dir).y + 1): the sky term uses the PRIMARY ray (D2)
    int depth;
};

// hit record + material::scatter.  Returns false when the path ends; `rad` is the
// radiance of the ended path (black on absorption).
__device__ __forceinline__ bool shade(path_t& p, stream& rs, const sphere_rec& s, float4 mrec, float t) {
    // sphere.h:66-70, hittable.h:16-22
    const v3 pos = p.r.o + t * p.r.d;
    const v3 cc = s.c + p.r.tm * s.mo;
    const v3 outward = rcp_(s.radius) * (pos - cc);
    const bool front = dot(p.r.d, outward) < 0.0f;
    const v3 nrm = front ? outward : -outward;
    const int type = (uint32_t)s.mat >> CRT_MAT_TYPE_SHIFT;
    if (type == 3) return false;  // the base class never scatters (material.h:11-19): no draw, the path ends black
    if (type == 0) {  // lambertian, material.h:26-38
        v3 dir = nrm + random_unit_vector(rs);
        if (((double)fabsf(dir.x) < 1e-8) && ((double)fabsf(dir.y) < 1e-8) && ((double)fabsf(dir.z) < 1e-8)) dir = nrm;
        p.r.o = pos; p.r.d = dir;
        p.thr = p.thr * mk(mrec.x, mrec.y, mrec.z);
        return true;
    } else if (type == 1) {  // metal, material.h:48-55
        const v3 refl = p.r.d - (2.0f * dot(p.r.d, nrm)) * nrm;
        const v3 dir = unit(refl) + (mrec.w * random_unit_vector(rs));
        p.r.o = pos; p.r.d = dir;
        p.thr = p.thr * mk(mrec.x, mrec.y, mrec.z);
        return dot(dir, nrm) > 0.0f;
    } else {  // dielectric, material.h:66-85
        const float ri = front ? mrec.y : mrec.x;
        const v3 ud = unit(p.r.d);
        const float cos_theta = fminf(dot(-ud, nrm), 1.0f);
        const float sin_theta = sqrt_(1.0f - cos_theta * cos_theta);
        bool do_reflect = ri * sin_theta > 1.0f;
        if (!do_reflect) do_reflect = reflectance(cos_theta, front ? mrec.w : mrec.z) > (1.0f - rs.next_block().x);
        v3 dir;
        if (do_reflect) {
            dir = ud - (2.0f * dot(ud, nrm)) * nrm;
        } else {  // refract, vec3.h:138-143
            const v3 perp = ri * (ud + cos_theta * nrm);
            const v3 par = (-sqrt_(fabsf(1.0f - len2(perp)))) * nrm;
            dir = perp + par;
        }
        p.r.o = pos; p.r.d = dir;  // attenuation (1,1,1): thr * 1 == thr
        return true;
    }
}

__device__ __forceinline__ v3 sky(const path_t& p) {  // render_with_cuda.h:49-52
    const float a = p.sky_a;
    const v3 c = mk((1.0f - a) * 1.0f + a * 0.5f, (1.0f - a) * 1.0f + a * 0.7f, (1.0f - a) * 1.0f + a * 1.0f);
    return p.thr * c;
}

__device__ __forceinline__ void start_path(path_t& p, stream& rs, const crt_render_args& a, int px, int py,
                                           uint32_t pixel_index, uint32_t sample) {
    rs.pixel = pixel_index; rs.sample = sample; rs.event = 0;
    p.r = get_ray(rs, a.cam, px, py);
    p.thr = mk(1.0f, 1.0f, 1.0f);
    p.sky_a = 0.5f * (unit(p.r.d).y + 1.0f);
    p.depth = 0;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---------------------------------------------------------------------------------------
// Accumulation: a sample's radiance is added to its pixel as a 64-bit fixed-point number with 40
// fractional bits (values >= 2^-16 are represented exactly, smaller ones are truncated to 2^-40).
// Integer addition is associative, so the per-pixel sum does not depend on the order in which
// samples finish, on the launch geometry, or on how samples are split over launches and GPUs.
__device__ __forceinline__ unsigned long long to_fixed(float r) {
    return (unsigned long long)(long long)(fminf(r, 4096.0f) * 1099511627776.0f);  // 2^40; NaN -> 4096
}
// The same conversion for a radiance that is not negative (every radiance of a scene with non-negative albedos): the
// two halves of the integer by one floor and one exact fma -- v = hi 2^32 + lo with hi = floor(v 2^-32), and
// v - hi 2^32 < 2^32 keeps v's own low bits, which the unsigned conversion truncates -- 5 instructions where the
// compiler's signed float -> int64 sequence takes 11 (absolute value, and the sign put back by two xors and a 64-bit
// subtraction).  PROVEN equal to to_fixed() for every one of the 2^31 + 1 arguments it accepts (crt_selftest_exact_math).
__device__ __forceinline__ float fixed_scaled(float r) { return fminf(r, 4096.0f) * 1099511627776.0f; }
__device__ __forceinline__ bool fixed_plain(float v) { return v >= 0.0f; }  // v = fixed_scaled(r): never NaN
__device__ __forceinline__ unsigned long long fixed_short(float v) {
    const float hf = floorf(v * 2.3283064365386963e-10f);  // 2^-32: exact scaling
    const uint32_t lo = (uint32_t)__builtin_fmaf(hf, -4294967296.0f, v);
    return ((unsigned long long)(uint32_t)hf << 32) | lo;
}
// the three channels of one sample: the short conversion, the general one when some lane of the wave holds a negative one
__device__ __forceinline__ void to_fixed3(v3 rad, unsigned long long& fx, unsigned long long& fy, unsigned long long& fz) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float vx = fixed_scaled(rad.x), vy = fixed_scaled(rad.y), vz = fixed_scaled(rad.z);
    fx = fixed_short(vx); fy = fixed_short(vy); fz = fixed_short(vz);
    const bool plain = fixed_plain(fminf(fminf(vx, vy), vz));
    if (__builtin_expect(__ballot(!plain) != 0, 0)) {
        if (!plain) { fx = to_fixed(rad.x); fy = to_fixed(rad.y); fz = to_fixed(rad.z); }
    }
#else
    fx = to_fixed(rad.x); fy = to_fixed(rad.y); fz = to_fixed(rad.z);  // host pass of the single-source compile: never runs
#endif
}

// The camera vectors of the kernel argument, read by scalar loads AT THE POINT OF USE (a new path's primary ray).  Read
// through `a.cam`, the compiler loads the 18 floats once per kernel, holds them in scalar registers across the whole
// state machine, runs out of those and parks them in the lanes of a vector register: 18 v_readlane_b32 per shade
// phase, each as dear as two fp32 instructions (tools/ubench/valu_pair.hip), and one VGPR less for everything else.
struct cam_ref {
    enum { center = 0, p00 = 3, du = 6, dv = 9, ddu = 12, ddv = 15 };  // float offsets in crt_dev_camera
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(4))) const float kfloat;
    kfloat* f;
    __device__ __forceinline__ explicit cam_ref(const crt_render_args&) {
        // the kernel's only argument starts the kernel-argument segment, the camera starts the argument
        static_assert(offsetof(crt_render_args, cam) == 0 && offsetof(crt_dev_camera, center) == 0 &&
                      offsetof(crt_dev_camera, ddv) == 15 * sizeof(float), "camera layout");
        f = (kfloat*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(f));  // not a known invariant load any more: stays where it is written
    }
#else
    const float* f;
    __device__ __forceinline__ explicit cam_ref(const crt_render_args& a) : f(a.cam.center) {}  // host pass of the single-source compile: never runs
#endif
    __device__ __forceinline__ v3 v(int at) const { return mk(f[at], f[at + 1], f[at + 2]); }
};

// Wave-scheduled BVH path tracer.  Every lane is a small state machine:
//   RUN    walking internal nodes (one child-pair record per step),
//   LEAF   holding one or two candidate spheres of a leaf-parent, waiting for the sphere phase,
//   SHADE  traversal finished (or no path yet), waiting for hit record + scatter + next ray,
//   DONE   the workgroup's item pool is empty.
// Each iteration the WAVE picks one phase to run (wave-uniform branch on ballot counts): the
// sphere and shade phases are postponed until enough lanes wait for them, so their code is not
// re-executed for one or two lanes on every traversal step.  A lane's own sequence of operations
// is exactly external_hit / ray_color (helper_experimental.h:1-78): a lane never steps past a
// pending sphere test, so visit order, intervals and results are those of the simple loop.
//
// Work items.  A workgroup owns T tiles of 8x8 pixels and all `ns` samples of them: 64*T*ns items,
// item i = sample (i / 64T), tile-pixel (i % 64T).  A lane that needs a path takes the next item
// from a counter in LDS (one ds_add per wave and shade phase), so no lane idles while items are
// left, whatever the path lengths of "its" pixels.  Radiance goes to fixed-point accumulators in
// LDS (ds_add_u64) and from there to the frame with one global atomic per component.
enum { ST_RUN = 0, ST_LEAF = 1, ST_SHADE = 2, ST_DONE = 3 };

struct wg_pool {
    unsigned long long* acc;  // LDS: [64 * T][3]; PERSIST: this wave's [64][3]
    uint32_t* counter;        // LDS: next unassigned item
    const int2* origin;       // LDS: pixel origin (x, y) of each of the T tiles; x < 0: tile outside the frame
    uint32_t total;           // 64 * T * ns
    int shift;                // log2(64 * T)
};

// Persistent waves (PERSIST).  The frame is cut into units = (8x8 tile, chunk of `unit_samples`
// samples); a wave takes the next unit from a global counter whenever its lanes have used up the
// items (pixel, sample) of the current one, and exits when the counter runs past the last unit.
// Nothing is shared between waves, so there is no barrier and no workgroup-wide tail.  The wave
// keeps ONE tile of accumulators in LDS: on a unit switch it flushes them to the frame (global
// 64-bit atomics) and the paths of the old unit that are still in flight (at most 64) add their
// radiance to the frame directly when they end.  Integer sums: the order does not matter.
// The wave's unit state lives in LDS next to its accumulators (8 dwords, wave-private, read and
// written by all active lanes with identical values), not in registers: it is touched once per
// shade phase, and the walking loop keeps its register budget.
enum { U_NITEMS = 0, U_NEXT = 1, U_EPOCH = 2, U_FLAGS = 3, U_OX = 4, U_OY = 5, U_SBASE = 6 };  // U_FLAGS: 1 have, 2 drained

__device__ __forceinline__ uint32_t uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// The unit state is exchanged between the lanes of one wave through LDS: whichever lanes are active write it, all
// lanes read it later.  The accesses are plain ds_read / ds_write; wave_sync() (a fence at wavefront scope: no
// instruction, the lanes of a wave run in lock step) tells the compiler that other lanes may have written LDS, so it
// never carries a value loaded before a block this lane sat out across it.  (Relaxed atomic accesses for every word
// were measured 1.1 % slower: 79 instead of 77 VGPRs and unmerged LDS reads.)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ uint32_t ust_ld(const uint32_t* ust, int i) { return ust[i]; }
__device__ __forceinline__ void ust_st(uint32_t* ust, int i, uint32_t v) { ust[i] = v; }

__device__ __forceinline__ void unit_flush(const crt_render_args& a, unsigned long long* acc, const uint32_t* ust, int lane) {
    const int ox = (int)uniform(ust_ld(ust, U_OX)), oy = (int)uniform(ust_ld(ust, U_OY));
    // the 192 accumulators are dealt round-robin to whichever lanes are active here
    const unsigned long long m_act = __ballot(true);
    const int n_act = __popcll(m_act);
    const int first = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m_act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_act, 0u));
    for (int e = first; e < 192; e += n_act) {
        const unsigned long long q = acc[e];
        acc[e] = 0ull;
        const int pix = e / 3, c = e - 3 * pix;
        const int px = ox + (pix & 7), py = oy + (pix >> 3);
        if (q != 0ull && px < a.cam.width && py < a.cam.height)
            atomicAdd(&a.sums[3 * ((size_t)py * a.cam.width + px) + c], q);
    }
}

// flush the current unit and take the next one from the global counter (any set of active lanes)
__device__ __forceinline__ void unit_next(const crt_render_args& a, unsigned long long* acc, uint32_t* ust, int lane) {
    if (uniform(ust_ld(ust, U_FLAGS)) & 1u) unit_flush(a, acc, ust, lane);
    uint32_t u = 0;
    const unsigned long long m_act = __ballot(true);
    if (lane == __builtin_ctzll(m_act)) u = (uint32_t)atomicAdd(&a.counters[15], 1ull);
    u = uniform(u);  // readfirstlane: the first active lane is the one that asked
    ust_st(ust, U_NEXT, 0u);
    if (u >= (uint32_t)a.n_units) {
        // no unit left: the paths still in flight belong to a flushed tile, like after any switch
        ust_st(ust, U_FLAGS, 2u); ust_st(ust, U_NITEMS, 0u);
        if ((a.census & 2) && lane == __builtin_ctzll(m_act)) {  // timeline census: when this wave found the queue empty
            const unsigned long long t = wall_clock64();
            atomicAdd(&a.counters[22], t); atomicMax(&a.counters[23], t);
            const uint32_t wid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
            if (wid < 8192u) a.counters[CRT_TIMELINE_BASE + 3 * wid + 1] = t;
        }
        ust_st(ust, U_EPOCH, uniform(ust_ld(ust, U_EPOCH)) + 1u);
    } else {
        // coarse units first, then the fine ones (crt_device.h)
        const bool fine = u >= (uint32_t)a.n_coarse;
        const uint32_t v = fine ? u - (uint32_t)a.n_coarse : u;
        const uint32_t chunks = (uint32_t)(fine ? a.fine_chunks : a.unit_chunks);
        const uint32_t size = (uint32_t)(fine ? a.fine_samples : a.unit_samples);
        const uint32_t share_tile = v / chunks, chunk = v - share_tile * chunks;
        const uint32_t tile = share_tile * (uint32_t)a.tile_stride + (uint32_t)a.tile_offset;  // this launch's tile share
        const uint32_t ty = tile / (uint32_t)a.tiles_x, tx = tile - ty * (uint32_t)a.tiles_x;
        const uint32_t first = (fine ? (uint32_t)a.ns_coarse : 0u) + chunk * size;
        const uint32_t left = (uint32_t)a.ns - first;  // coarse chunks are all full (ns_coarse <= ns)
        ust_st(ust, U_OX, tx * 8u); ust_st(ust, U_OY, ty * 8u);
        ust_st(ust, U_SBASE, (uint32_t)a.s0 + first);
        ust_st(ust, U_NITEMS, 64u * (left < size ? left : size));
        ust_st(ust, U_EPOCH, uniform(ust_ld(ust, U_EPOCH)) + 1u);
        ust_st(ust, U_FLAGS, 1u);
    }
    wave_sync();
}

// Scene access of the walking loop.  In LDS (IN_LDS) the child boxes are stored once per ray
// direction sign: array 2*axis + (dir[axis] < 0) holds, for pair k, (near_L, far_L, near_R, far_R) =
// the box planes this ray meets first / last on that axis.  A lane reads three float4 (one per axis,
// picked by ITS signs), and max/min over known near/far planes replace the per-axis min/max pair:
// t_near/t_far are the same products (b - o) * inv the reference forms, only looked up instead of
// compared (plane_lo <= plane_hi and rounding is monotone, so near == min and far == max exactly).
struct walk_scene {
    const float4* pairs;      // HBM: 64-byte child-pair records (crt_device.h), always valid
    const float4* axes;       // LDS: [6][L] sign-sorted planes, or null
    const float4* top;        // scene in HBM: LDS copy of q0, q1, q2 of the first `top_n` pair records ([3][top_n]), or null
    uint32_t top_n;
    const int2* leaf_tags;    // LDS: sphere tags of the leaf-parents' children, indexed by pair index (pre-offset)
    uint32_t L;               // num_leaves
};

// ---------------------------------------------------------------------------------------
// SCHED=3: the extend stage split out into WALKER waves (north_star's generate / extend / shade alternative, with the
// queues in LDS instead of HBM).  The first `n_walkers` waves of a workgroup only walk boxes: each of their lanes holds
// one compact walk request (origin, 1/d, interval end, node index, pending mask) and is refilled from a request ring
// the step after it reaches a leaf or finishes, so the walking step runs with (nearly) all 64 lanes.  The other waves
// are OWNERS: the scheduled tracer below without its walking loop -- they post a request whenever one of their rays
// has boxes to walk and run the sphere and shade phases on what comes back.  Every ray's own operation sequence is
// unchanged (the walker executes exactly the steps the owner lane would have), so the frame is bit-identical.
//
// Exchange, all in LDS, per OWNER wave (single producer, single consumer: walker j serves the owners o = j mod n_walkers):
//   ring   CRT_RING_CAP entries of two float4 (SoA): { o.xyz, best_t }, { inv.xyz, k | pending << 12 | owner lane << 26 }
//   ctl    { tail (owner writes), head (walker writes), done (owner writes once), - }
//   mbox   64 words, one per owner lane: the walker's answer  VALID | k | pending << 12  (k == 0: finished; k >= L:
//          leaf held for the sphere phase, bit 0 of pending = its sibling was hit too)
// DS operations of one wave execute in issue order, so "entry, then tail" / "entry read, then head" need no more
// than the compiler keeping that order (workgroup-scope fences).
#define CRT_MBOX_VALID 0x80000000u  // (CRT_RING_CAP, CRT_XCHG_FLOAT4_PER_OWNER: crt_device.h, the launcher sizes the LDS with them)

struct owner_link {
    uint32_t q0, q1;  // LDS byte offsets of the two ring halves of this owner wave
    uint32_t ctl;     // LDS byte offset of { tail, head, done, - }
    uint32_t mbox;    // LDS byte offset of its 64 mailbox words
};

__device__ __forceinline__ uint32_t lds_off(const void* p) { return (uint32_t)(uintptr_t)p; }  // low half of a flat LDS address
__device__ __forceinline__ uint32_t xl_ld(uint32_t off) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    return __hip_atomic_load((lds_u32*)off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    return off;
#endif
}
__device__ __forceinline__ void xl_st(uint32_t off, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    __hip_atomic_store((lds_u32*)off, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
}
__device__ __forceinline__ f4 xl_ld4(uint32_t off) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) const volatile f4 lds_f4;
    return *(lds_f4*)off;
#else
    return f4{0, 0, 0, 0};
#endif
}
__device__ __forceinline__ void xl_st4(uint32_t off, f4 v) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) volatile f4 lds_f4;
    *(lds_f4*)off = v;
#endif
}
__device__ __forceinline__ void wg_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
__device__ __forceinline__ void wg_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }

// A walker wave (SCHED=3).  xchg: LDS byte offset of the owners' exchange areas (CRT_XCHG_FLOAT4_PER_OWNER float4 each);
// this walker serves the owners first, first + stride, ... < n_owners.
template <bool COUNT>
__device__ __forceinline__ void walker_loop(const crt_render_args& a, const walk_scene& ws, uint32_t xchg, uint32_t first,
                                            uint32_t stride, uint32_t n_owners, uint32_t& n_pairs) {
    const float tmin = 0.001f;
    const float4* pairs = ws.pairs;
    const uint32_t NL = ws.L;
    const int lane = threadIdx.x & 63;
    v3 o = mk(0, 0, 0), inv = mk(1, 1, 1);
    float best_t = 0.0f;
    uint32_t k = 0, pending = 0, dst = 0;
    uint32_t offx = 0, offy = 0, offz = 0;
    bool exact = false, busy = false;
    const int refill_min = a.refill_min;
    unsigned long long c_refills = 0, c_taken = 0, c_idle = 0, c_steps = 0, c_busy = 0;  // census
    uint32_t rr = 0;           // which of its owners the walker visits first (rotates)
    uint32_t idle_polls = 0;   // watchdog: consecutive polls that found nothing
    int backoff = 0, patience = a.walker_patience;
    if (a.walker_prio) __builtin_amdgcn_s_setprio(2);
    for (;;) {
        const unsigned long long m_idle = __ballot(!busy);
        const int n_idle = __popcll(m_idle);
        if (backoff > 0 && n_idle < 64) --backoff;  // the rings were empty a moment ago: walk a few steps before looking again
        else if (n_idle >= refill_min) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_idle, 0u));
            uint32_t taken = 0;
            bool all_done = true;
            uint32_t ow = first + rr * stride;
            if (ow >= n_owners) { ow = first; rr = 0; }
            for (uint32_t i = first; i < n_owners; i += stride) {
                const uint32_t base = xchg + ow * (16u * CRT_XCHG_FLOAT4_PER_OWNER);
                const uint32_t ctl = base + 32u * CRT_RING_CAP;
                const f4 cw = xl_ld4(ctl);  // { tail, head, done, - } in one read
                const uint32_t tail = uniform(__float_as_uint(cw.x)), head = uniform(__float_as_uint(cw.y));
                all_done = all_done && (uniform(__float_as_uint(cw.z)) != 0u);
                const uint32_t avail = tail - head, room = (uint32_t)n_idle - taken;
                const uint32_t take = avail < room ? avail : room;
                if (take) {
                    wg_acquire();
                    if (!busy && rank >= taken && rank < taken + take) {
                        const uint32_t e = (head + (rank - taken)) & (CRT_RING_CAP - 1u);
                        const f4 q0 = xl_ld4(base + 16u * e), q1 = xl_ld4(base + 16u * CRT_RING_CAP + 16u * e);
                        o = mk(q0.x, q0.y, q0.z); best_t = q0.w;
                        inv = mk(q1.x, q1.y, q1.z);
                        const uint32_t w = __float_as_uint(q1.w);
                        k = w & 0xfffu; pending = (w >> 12) & 0x3fffu;
                        dst = base + 16u * (2u * CRT_RING_CAP + 1u) + 4u * (w >> 26);  // the owner lane's mailbox word
                        exact = !finite3(inv);
                        // finite reciprocals: sign(1/d) == sign(d), and d == -0 gives -inf (exact path, offsets unused)
                        offx = (inv.x < 0.0f ? NL : 0u) << 4;
                        offy = (inv.y < 0.0f ? 3u * NL : 2u * NL) << 4;
                        offz = (inv.z < 0.0f ? 5u * NL : 4u * NL) << 4;
                        busy = true;
                    }
                    taken += take;
                    wg_release();  // the entries have been read: hand them back to the owner
                    if (lane == 0) xl_st(ctl + 4u, head + take);
                }
                ow += stride;
                if (ow >= n_owners) ow = first;
            }
            ++rr;
            if (COUNT && (a.census & 1)) {  // walker census: refill attempts, requests taken, polls of a wholly idle wave
                c_refills += 1; c_taken += taken;
                if (taken == 0u && n_idle == 64) c_idle += 1;
            }
            if (taken == 0u && n_idle == 64) {
                if (all_done) break;                    // every owner has finished: nothing can arrive any more
                if (++idle_polls > (1u << 24)) {        // watchdog (seconds): never spin for ever on a protocol error
                    if (lane == 0) atomicAdd(&a.counters[29], 1ull);
                    break;
                }
                for (int i = 0; i < a.owner_sleep; ++i) __builtin_amdgcn_s_sleep(1);  // 64 clocks each
                continue;
            }
            idle_polls = 0;
            backoff = taken == 0u ? a.walker_backoff : 0;
        }
        // a thin wave (fewer than min_busy rays on board) waits a little for more requests instead of spending a whole
        // wave-step on a few lanes -- but only `patience` polls in a row, the rays on board must get on
        if (a.walker_min_busy > 0) {
            const int n_busy = __popcll(__ballot(busy));
            if (n_busy < a.walker_min_busy && patience > 0) {
                --patience;
                backoff = 0;
                __builtin_amdgcn_s_sleep(1);
                continue;
            }
            patience = a.walker_patience;
        }
        if (COUNT && (a.census & 1)) {  // walking steps and the lanes they serve
            c_steps += 1; c_busy += (unsigned long long)__popcll(__ballot(busy));
        }
        if (busy) {
            if (COUNT) n_pairs += 1;
            bool hitL, hitR;
            if (__builtin_expect(exact, 0)) {
                const float4 q0 = pairs[4 * k], q1 = pairs[4 * k + 1], q2 = pairs[4 * k + 2], q3 = pairs[4 * k + 3];
                hitL = slab(q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, o, inv, tmin, best_t);
                hitR = slab(q0.z, q0.w, q1.z, q1.w, q2.z, q2.w, o, inv, tmin, best_t);
                hitL = hitL & (__float_as_int(q3.x) != CRT_TAG_EMPTY);
                hitR = hitR & (__float_as_int(q3.y) != CRT_TAG_EMPTY);
            } else {
                const char* planes = (const char*)ws.axes;
                const float4 ax = *(const float4*)(planes + ((k << 4) + offx));
                const float4 ay = *(const float4*)(planes + ((k << 4) + offy));
                const float4 az = *(const float4*)(planes + ((k << 4) + offz));
#if CRT_FAST
                const v3 noi = mk(-(o.x * inv.x), -(o.y * inv.y), -(o.z * inv.z));
                const float nLx = __builtin_fmaf(ax.x, inv.x, noi.x), fLx = __builtin_fmaf(ax.y, inv.x, noi.x);
                const float nRx = __builtin_fmaf(ax.z, inv.x, noi.x), fRx = __builtin_fmaf(ax.w, inv.x, noi.x);
                const float nLy = __builtin_fmaf(ay.x, inv.y, noi.y), fLy = __builtin_fmaf(ay.y, inv.y, noi.y);
                const float nRy = __builtin_fmaf(ay.z, inv.y, noi.y), fRy = __builtin_fmaf(ay.w, inv.y, noi.y);
                const float nLz = __builtin_fmaf(az.x, inv.z, noi.z), fLz = __builtin_fmaf(az.y, inv.z, noi.z);
                const float nRz = __builtin_fmaf(az.z, inv.z, noi.z), fRz = __builtin_fmaf(az.w, inv.z, noi.z);
#else
                const float nLx = (ax.x - o.x) * inv.x, fLx = (ax.y - o.x) * inv.x;
                const float nRx = (ax.z - o.x) * inv.x, fRx = (ax.w - o.x) * inv.x;
                const float nLy = (ay.x - o.y) * inv.y, fLy = (ay.y - o.y) * inv.y;
                const float nRy = (ay.z - o.y) * inv.y, fRy = (ay.w - o.y) * inv.y;
                const float nLz = (az.x - o.z) * inv.z, fLz = (az.y - o.z) * inv.z;
                const float nRz = (az.z - o.z) * inv.z, fRz = (az.w - o.z) * inv.z;
#endif
                const float loL = fmaxf(fmaxf(fmaxf(nLx, nLy), nLz), tmin), hiL = fminf(fminf(fminf(fLx, fLy), fLz), best_t);
                const float loR = fmaxf(fmaxf(fmaxf(nRx, nRy), nRz), tmin), hiR = fminf(fminf(fminf(fRx, fRy), fRz), best_t);
                hitL = !(hiL <= loL);
                hitR = !(hiR <= loR);
            }
            const bool both = hitL && hitR, any = hitL || hitR;
            const uint32_t kd = (k << 1) | (hitL ? 0u : 1u), pd = (pending << 1) | (both ? 1u : 0u);
            const uint32_t up = (uint32_t)__builtin_ctz(pending);  // the sentinel keeps pending != 0
            const uint32_t kp = (k >> up) ^ 1u, pp = (pending >> up) ^ 1u;
            k = any ? kd : kp;
            pending = any ? pd : pp;
            if (!((k - 1u) < (NL - 1u))) {  // a leaf to test, or the traversal is over: back to the owner lane
                xl_st(dst, CRT_MBOX_VALID | k | (pending << 12));
                busy = false;
            }
        }
    }
    __builtin_amdgcn_s_setprio(0);
    if (COUNT && (a.census & 1) && lane == 0) {  // (summed in registers: atomics inside the polling loops would serialise the launch)
        atomicAdd(&a.counters[0], c_refills); atomicAdd(&a.counters[1], c_taken); atomicAdd(&a.counters[2], c_idle);
        atomicAdd(&a.counters[11], c_steps); atomicAdd(&a.counters[12], c_busy);
    }
}

template <bool IN_LDS, bool COUNT, bool PERSIST, int ROLE = 0>
__device__ __forceinline__ void trace_bvh_scheduled(const crt_render_args& a, const walk_scene& ws, const float4* spheres,
                                                    const float4* mats, const wg_pool& pool, uint32_t& n_rays,
                                                    uint32_t& n_samples, uint32_t& n_pairs, uint32_t& n_tests,
                                                    const owner_link link = owner_link{0, 0, 0, 0}) {
    const float tmin = 0.001f;
    const float4* pairs = ws.pairs;
    const uint32_t NL = ws.L;
    const int lane = threadIdx.x & 63;
    stream rs; rs.k0 = a.key0; rs.k1 = a.key1; rs.pixel = 0; rs.sample = 0; rs.event = 0;
    path_t p;
    p.r.o = mk(0, 0, 0); p.r.d = mk(0, 0, 1); p.r.tm = 0; p.thr = mk(1, 1, 1); p.sky_a = 0; p.depth = 0;
    v3 inv = mk(1, 1, 1);
    float ray_a = 1.0f, best_t = __builtin_huge_valf();
    int best_obj = -1;
    // The traversal state IS the node index k (1-based heap order, NL = number of leaves):
    //   0 < k < NL   RUN: k is the pair (parent) whose two child boxes are tested next
    //   k >= NL      LEAF: k is the first hit leaf, its sibling k^1 was hit too if bit 0 of `pending` is set
    //   k == 0       SHADE (traversal finished or no path yet), or DONE once `finished`
    // pending: bit i = the right sibling i levels above the current level is still to be visited;
    // above the root's level sits a sentinel bit, so a pop is always defined and popping the
    // sentinel yields k == 0.
    uint32_t k = 0, pending = 0, slot = 0;
    uint32_t offx = 0, offy = 0, offz = 0;  // IN_LDS: byte offset of this ray's plane arrays (+ base)
    bool fresh = true, exact = false, finished = false;
    bool out = false;          // ROLE 1 (owner): this lane's ray is with a walker (posted, answer not yet collected)
    uint32_t ring_tail = 0;    // ROLE 1: the owner's own copy of its ring's tail (it is the only writer)
    uint32_t wait_polls = 0;   // ROLE 1: watchdog, consecutive polls without anything to do
    unsigned long long c_waits = 0;  // ROLE 1 census: scheduler iterations that found too little to do
    uint32_t* ust = (uint32_t*)(pool.acc + 192);  // PERSIST: the wave's unit state, behind its accumulators
    // census (COUNT && a.census): shader cycles of this wave per phase, counters[16..21]
    const bool timing = COUNT && (a.census & 2);
    unsigned long long tc[6] = {0, 0, 0, 0, 0, 0};  // scheduler, walking, spheres, shade part 1, block loop, shade part 3
    long long t_mark = timing ? (long long)__builtin_readcyclecounter() : 0;
    if (timing && lane == 0) {  // timeline census (100 MHz wall clock): start, queue-empty and exit time of each wave
        const unsigned long long t = wall_clock64();
        atomicMax(&a.counters[24], ~t); atomicAdd(&a.counters[27], t); atomicAdd(&a.counters[28], 1ull);
        const uint32_t wid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // per-wave records: start, queue empty, exit
        if (PERSIST && wid < 8192u) a.counters[CRT_TIMELINE_BASE + 3 * wid] = t;
    }
#if CRT_LITERAL_AT_ENTRY
    // Rays with a non-finite reciprocal (a zero or denormal direction component: rare) take the literal comparisons of
    // aabb.h:48-62 on the HBM records until they reach a leaf or finish.
    auto walk_literal = [&]() {
        while (exact && ((k - 1u) < (NL - 1u))) {
            if (COUNT) n_pairs += 1;
            const float4 q0 = pairs[4 * k], q1 = pairs[4 * k + 1], q2 = pairs[4 * k + 2], q3 = pairs[4 * k + 3];
            bool hitL = slab(q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, p.r.o, inv, tmin, best_t);
            bool hitR = slab(q0.z, q0.w, q1.z, q1.w, q2.z, q2.w, p.r.o, inv, tmin, best_t);
            hitL = hitL & (__float_as_int(q3.x) != CRT_TAG_EMPTY);
            hitR = hitR & (__float_as_int(q3.y) != CRT_TAG_EMPTY);
            successor(k, pending, hitL, hitR);
        }
    };
#endif
#define CRT_LAP(i) do { if (timing) { const long long t_ = (long long)__builtin_readcyclecounter(); tc[i] += (unsigned long long)(t_ - t_mark); t_mark = t_; } } while (0)

    for (;;) {
      // The scheduler, the walking phase and the sphere phase form an inner loop of their own: it
      // changes only k, `pending` and the best hit, so the rest of the lane state stays in place in
      // its registers across it (no copies at the loop's merge points).
      int phase = -1;
      // (`finished` only changes in the shade phase: its lane mask is formed once per outer iteration, and the scheduler's
      // ballot below is a plain compare and a scalar and-not instead of a materialised boolean)
      const unsigned long long m_fin = (ROLE == 1) ? 0ull : __ballot(finished);
#if CRT_FAST
      // (likewise the lanes with a non-finite reciprocal, set where a segment begins -- in the tolerance build only: 63.2 ->
      // 62.5 ms there, 72.4 -> 72.8 ms in the exact build, whose registers fall differently; alternating A/B)
      const unsigned long long m_exact = __ballot(exact);
#endif
      for (;;) {
        if (ROLE == 1) {
            // collect: a lane whose ray is out reads its mailbox word
            if (out) {
                const uint32_t v = xl_ld(link.mbox + 4u * (uint32_t)lane);
                if (v & CRT_MBOX_VALID) {
                    k = v & 0xfffu; pending = (v >> 12) & 0x3ffffu; out = false;
                    xl_st(link.mbox + 4u * (uint32_t)lane, 0u);
                }
            }
            // post: rays with boxes to walk go into the ring (as many as it has room for; the rest try again)
            const bool want = !out && ((k - 1u) < (NL - 1u));
            const unsigned long long m_post = __ballot(want);
            if (m_post != 0) {
                const uint32_t head = uniform(xl_ld(link.ctl + 4u));
                const uint32_t room = CRT_RING_CAP - (uniform(ring_tail) - head);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_post >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_post, 0u));
                if (want && rank < room) {
                    const uint32_t e = (uniform(ring_tail) + rank) & (CRT_RING_CAP - 1u);
                    xl_st4(link.q0 + 16u * e, f4{p.r.o.x, p.r.o.y, p.r.o.z, best_t});
                    xl_st4(link.q1 + 16u * e, f4{inv.x, inv.y, inv.z, __uint_as_float(k | (pending << 12) | ((uint32_t)lane << 26))});
                    out = true;
                }
                const uint32_t n_post = (uint32_t)__popcll(m_post);
                ring_tail = uniform(ring_tail) + (n_post < room ? n_post : room);
                wg_release();  // entries before the tail
                if (lane == __builtin_ctzll(m_post)) xl_st(link.ctl, ring_tail);
            }
        }
        const unsigned long long m_run = __ballot((k - 1u) < (NL - 1u));
        const unsigned long long m_leaf = __ballot(k >= NL);
        const unsigned long long m_shade = (ROLE == 1) ? __ballot((k == 0u) & !finished) : (__builtin_amdgcn_uicmp(k, 0u, 32 /* eq */) & ~m_fin);  // (an owner's watchdog may set `finished` inside this loop)
        const int n_run = __popcll(m_run), n_leaf = __popcll(m_leaf), n_shade = __popcll(m_shade);
        const int alive = n_run + n_leaf + n_shade;
        if (alive == 0) break;
        int ph;
        if (n_leaf * 16 >= alive * a.leaf_num && n_leaf > 0) ph = ST_LEAF;
        else if (n_shade * 16 >= alive * a.shade_num && n_shade > 0) ph = ST_SHADE;
        else if (n_run > 0) ph = ST_RUN;
        else ph = (n_leaf >= n_shade) ? ST_LEAF : ST_SHADE;
        if (COUNT && (a.census & 1) && ROLE == 1 && ph == ST_RUN) c_waits += 1;  // (owner waiting: counted in a register)
        else if (COUNT && (a.census & 1) && lane == 0) {  // scheduler census: iterations and served lanes per phase
            const int served = ph == ST_RUN ? n_run : ph == ST_LEAF ? n_leaf : n_shade;
            atomicAdd(&a.counters[4 + 2 * ph], 1ull);
            atomicAdd(&a.counters[5 + 2 * ph], (unsigned long long)served);
            atomicAdd(&a.counters[10], (unsigned long long)alive);
        }
        CRT_LAP(0);
        if (ph == ST_SHADE) { phase = ph; break; }
        if (ROLE == 1 && ph == ST_RUN) {
            // owner: the rays are with the walkers and too few lanes wait for the sphere or shade phase -- wait
            if (++wait_polls > (1u << 24)) {  // watchdog (seconds): never spin for ever on a protocol error
                if (lane == 0) atomicAdd(&a.counters[29], 1ull);
                k = 0u; finished = true; out = false;
            }
            for (int i = 0; i < a.owner_sleep; ++i) __builtin_amdgcn_s_sleep(1);  // 64 clocks each
            CRT_LAP(1);
        } else if (ROLE != 1 && ph == ST_RUN) {
            // stay in the box-walking loop until enough lanes have left it to be worth re-scheduling
            const int leave_at = n_run - max(1, n_run >> a.run_shift);
            // Rays with a non-finite reciprocal (a zero or denormal direction component: rare) take the literal
            // comparisons of aabb.h:48-62 on the HBM records, HERE, until they reach a leaf or finish -- so the hot
            // loop below carries no branch for them (6 scalar instructions per step less).
#if CRT_LITERAL_AT_ENTRY
            // (they walk where they enter the walking state: at the start of a segment and after a sphere test)
#else
            // in front of the hot loop, which then carries no branch for them
#if CRT_FAST
            if ((m_exact & m_run) != 0) {
#else
            if (__ballot(exact && ((k - 1u) < (NL - 1u))) != 0) {
#endif
                while (exact && ((k - 1u) < (NL - 1u))) {
                    if (COUNT) n_pairs += 1;
                    const float4 q0 = pairs[4 * k], q1 = pairs[4 * k + 1], q2 = pairs[4 * k + 2], q3 = pairs[4 * k + 3];
                    bool hitL = slab(q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, p.r.o, inv, tmin, best_t);
                    bool hitR = slab(q0.z, q0.w, q1.z, q1.w, q2.z, q2.w, p.r.o, inv, tmin, best_t);
                    hitL = hitL & (__float_as_int(q3.x) != CRT_TAG_EMPTY);
                    hitR = hitR & (__float_as_int(q3.y) != CRT_TAG_EMPTY);
                    successor(k, pending, hitL, hitR);
                }
            }
#endif
            // (the set of walking lanes is carried round the loop as a lane MASK in scalar registers and turned back
            // into the branch condition for free; carried as a boolean it cost three scalar mask merges per step)
#if CRT_LITERAL_AT_ENTRY
            unsigned long long m_walk = m_run;
#else
            unsigned long long m_walk = __ballot((k - 1u) < (NL - 1u));
#endif
            for (;;) {
                if (COUNT && (a.census & 1) && lane == 0) { atomicAdd(&a.counters[11], 1ull); atomicAdd(&a.counters[12], (unsigned long long)__popcll(m_walk)); }
                if (__builtin_amdgcn_inverse_ballot_w64(m_walk)) {
                    if (COUNT) n_pairs += 1;
                    float loL, hiL, loR, hiR;
                    if (IN_LDS) {
                        // empty children carry a far-away degenerate box (crt_api.cpp): near == far on
                        // every axis, so max(near) >= min(far) and the test rejects them by itself
                        const char* planes = (const char*)ws.axes;
                        const float4 ax = *(const float4*)(planes + ((k << 4) + offx));
                        const float4 ay = *(const float4*)(planes + ((k << 4) + offy));
                        const float4 az = *(const float4*)(planes + ((k << 4) + offz));
#if CRT_FAST
                        // -(o * inv) is formed here, per step (3 multiplies against the 12 subtractions saved), rather
                        // than kept in three more registers across the whole state machine: 81 VGPRs would cost the
                        // second workgroup per CU
                        const v3 noi = mk(-(p.r.o.x * inv.x), -(p.r.o.y * inv.y), -(p.r.o.z * inv.z));
                        const float nLx = __builtin_fmaf(ax.x, inv.x, noi.x), fLx = __builtin_fmaf(ax.y, inv.x, noi.x);
                        const float nRx = __builtin_fmaf(ax.z, inv.x, noi.x), fRx = __builtin_fmaf(ax.w, inv.x, noi.x);
                        const float nLy = __builtin_fmaf(ay.x, inv.y, noi.y), fLy = __builtin_fmaf(ay.y, inv.y, noi.y);
                        const float nRy = __builtin_fmaf(ay.z, inv.y, noi.y), fRy = __builtin_fmaf(ay.w, inv.y, noi.y);
                        const float nLz = __builtin_fmaf(az.x, inv.z, noi.z), fLz = __builtin_fmaf(az.y, inv.z, noi.z);
                        const float nRz = __builtin_fmaf(az.z, inv.z, noi.z), fRz = __builtin_fmaf(az.w, inv.z, noi.z);
#else
                        const float nLx = (ax.x - p.r.o.x) * inv.x, fLx = (ax.y - p.r.o.x) * inv.x;
                        const float nRx = (ax.z - p.r.o.x) * inv.x, fRx = (ax.w - p.r.o.x) * inv.x;
                        const float nLy = (ay.x - p.r.o.y) * inv.y, fLy = (ay.y - p.r.o.y) * inv.y;
                        const float nRy = (ay.z - p.r.o.y) * inv.y, fRy = (ay.w - p.r.o.y) * inv.y;
                        const float nLz = (az.x - p.r.o.z) * inv.z, fLz = (az.y - p.r.o.z) * inv.z;
                        const float nRz = (az.z - p.r.o.z) * inv.z, fRz = (az.w - p.r.o.z) * inv.z;
#endif
                        loL = fmaxf(fmaxf(fmaxf(nLx, nLy), nLz), tmin); hiL = fminf(fminf(fminf(fLx, fLy), fLz), best_t);
                        loR = fmaxf(fmaxf(fmaxf(nRx, nRy), nRz), tmin); hiR = fminf(fminf(fminf(fRx, fRy), fRz), best_t);
                    } else {
                        // scene in HBM: the upper levels of the tree (every ray walks them) come from LDS,
                        // which halves the L1 traffic that bounds this instance
                        // (explicit address spaces: left to itself the compiler merges the two branches into
                        // flat loads through a selected pointer, which cost twice the time per step)
                        f4 q0, q1, q2;
#if defined(__HIP_DEVICE_COMPILE__)
                        if (k < ws.top_n) {
                            typedef __attribute__((address_space(3))) const f4 lds_f4;
                            lds_f4* t = (lds_f4*)(uint32_t)(uintptr_t)ws.top;  // low half of a flat LDS address = LDS offset
                            q0 = t[k]; q1 = t[ws.top_n + k]; q2 = t[2u * ws.top_n + k];
                        } else {
                            typedef __attribute__((address_space(1))) const f4 glb_f4;
                            glb_f4* g = (glb_f4*)(uintptr_t)pairs;
                            q0 = g[4 * (size_t)k]; q1 = g[4 * (size_t)k + 1]; q2 = g[4 * (size_t)k + 2];
                        }
#else
                        q0 = q1 = q2 = f4{0, 0, 0, 0};  // host pass of the single-source compile: never runs
#endif
                        const v3 noi = mk(-(p.r.o.x * inv.x), -(p.r.o.y * inv.y), -(p.r.o.z * inv.z));  // CRT_FAST only (dead otherwise)
                        slab_lohi(q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, p.r.o, inv, noi, tmin, best_t, loL, hiL);
                        slab_lohi(q0.z, q0.w, q1.z, q1.w, q2.z, q2.w, p.r.o, inv, noi, tmin, best_t, loR, hiR);
                    }
                    // branch-free successor: descend (left first, right remembered) -- onto a leaf:
                    // hold it for the sphere phase -- or pop the nearest pending right sibling
                    successor_lohi(k, pending, loL, hiL, loR, hiR);
                }
                m_walk = __ballot((k - 1u) < (NL - 1u));
                if (__popcll(m_walk) <= leave_at) break;
            }
            CRT_LAP(1);
        }
        if (ph == ST_LEAF) {
            if (ROLE == 1) wait_polls = 0;
            if (k >= NL) {
                // sphere tags of the leaf pair that holds node k (helper_experimental.h:16-31: left, then right)
                // (in LDS the tag pairs of the leaf parents form one int array indexed by the leaf's own node index)
                int cand;
                if (IN_LDS) cand = ((const int*)ws.leaf_tags)[k];
                else { const float4 q3 = pairs[4 * (k >> 1) + 3]; cand = (k & 1u) ? __float_as_int(q3.y) : __float_as_int(q3.x); }
                const float root = sphere_root(load_sphere<IN_LDS && CRT_SCHED_SPHERES_IN_LDS>(spheres, cand), p.r, ray_a, tmin, best_t);
                if (COUNT) n_tests += 1;
                if (root > 0.0f) { best_t = root; best_obj = cand; }
                // ONE sphere per lane and phase: a hit sibling (2% of the leaf visits) is the lane's next
                // leaf, tested in the next sphere phase, instead of a second test that the whole wave
                // would step through for the one lane that needs it
                // -- which is the ordinary pop: bit 0 of `pending` set means "my sibling, zero levels up", and the pop
                // formula with up = 0 gives k ^ 1 and clears the bit; a root that is the only leaf (k = 1 under the
                // sentinel bit 0) pops to k = 0 the same way.  One formula, no case distinction (6 vector instructions less).
                const uint32_t up = (uint32_t)__builtin_ctz(pending);  // the sentinel keeps pending != 0
                k = (k >> up) ^ 1u;
                pending = (pending >> up) ^ 1u;
#if CRT_LITERAL_AT_ENTRY
                if (ROLE != 1) walk_literal();
#endif
            }
            CRT_LAP(2);
        }
      }
      if (phase < 0) break;
      if (ROLE == 1) wait_polls = 0;
      {
        {
            // The shade phase has three parts: (1) hit record / next work item, lanes in the shade
            // state only; (2) the Philox block loop, whole wave (lanes that need no block idle or, in
            // a tail pass, draw blocks for the lanes that do); (3) scatter direction or primary ray.
            const bool shading = (k == 0u) & !finished;
            enum { K_NONE = 0, K_VEC = 1, K_U = 2, K_CAM = 3, K_DISK = 4 };
            int kind = K_NONE, type = -1;
            // The values below are written in part 1 or in the block loop by exactly the lanes that read them in part 3
            // (by material type / kind of block), so they need no initial value.  Left uninitialised they save 17 v_mov
            // at the phase's merge points (1447 -> 1409 vector instructions in the code): the tolerance build gains
            // 1.4 % (64.5 -> 63.6 ms), the exact build -- 80 instead of 78 VGPRs, another allocation -- loses 0.15 %
            // (alternating A/B), so only the tolerance build does it.
#if CRT_FAST
#define CRT_INIT(v)
#else
#define CRT_INIT(v) = v
#endif
            v3 nrm CRT_INIT(mk(0, 0, 0)), base CRT_INIT(mk(0, 0, 0));
            float scale CRT_INIT(0.0f), aux CRT_INIT(0.0f);  // VEC: fuzz / -, dielectric: cos_theta / ri
            float u0 CRT_INIT(0.0f), u1 CRT_INIT(0.0f), u2 CRT_INIT(0.0f), u3 CRT_INIT(0.0f);  // VEC: p.xyz, lensq; K_U: u (u1: the Schlick reflectance); CAM: x_off, y_off, time
            bool cannot CRT_INIT(false);
            int px CRT_INIT(0), py CRT_INIT(0);
            float dx CRT_INIT(0.0f), dy CRT_INIT(0.0f);
#undef CRT_INIT
            if (shading) {
                // One Philox loop serves every lane of the phase, whatever it needs the block for:
                //   K_VEC  an attempt of random_unit_vector (lambertian, metal)     vec3.h:145-152
                //   K_U    the dielectric's reflect/refract draw                    material.h:78
                //   K_CAM  the camera block of a new path, then K_DISK              render_with_cuda.h:11-26
                // Each lane still draws ITS blocks in its own order (event counter per path); only
                // the wave executes one Philox per iteration instead of one per material loop.
                if (!fresh) {
                    ++n_rays;
                    if (best_obj >= 0) {
                        // hit record: sphere.h:66-70, hittable.h:16-22
                        const sphere_rec sp = load_sphere<IN_LDS && CRT_SCHED_SPHERES_IN_LDS>(spheres, best_obj);
                        const float4 mrec = mats[sp.mat & CRT_MAT_ID_MASK];
                        const v3 pos = p.r.o + best_t * p.r.d;
                        const v3 cc = sp.c + p.r.tm * sp.mo;
                        const v3 outward = rcp_(sp.radius) * (pos - cc);
                        const bool front = dot(p.r.d, outward) < 0.0f;
                        nrm = front ? outward : -outward;
                        type = (int)((uint32_t)sp.mat >> CRT_MAT_TYPE_SHIFT);
                        p.r.o = pos;
                        if (type == 0) {         // lambertian: nrm + ruv == nrm + 1.0f * ruv exactly
                            base = nrm; scale = 1.0f; kind = K_VEC;
                            p.thr = p.thr * mk(mrec.x, mrec.y, mrec.z);
                        } else if (type == 1) {  // metal, material.h:50-51
                            base = unit(p.r.d - (2.0f * dot(p.r.d, nrm)) * nrm); scale = mrec.w; kind = K_VEC;
                            p.thr = p.thr * mk(mrec.x, mrec.y, mrec.z);
                        } else if (type == 3) {  // the base class never scatters (material.h:11-19): no block is drawn
                            kind = K_NONE;
                        } else {                 // dielectric, material.h:69-75
                            aux = front ? mrec.y : mrec.x;
                            base = unit(p.r.d);
                            scale = fminf(dot(-base, nrm), 1.0f);
                            const float sin_theta = sqrt_(1.0f - scale * scale);
                            cannot = aux * sin_theta > 1.0f;
                            kind = cannot ? K_NONE : K_U;  // the draw is short-circuited away under TIR
                            u1 = reflectance(scale, front ? mrec.w : mrec.z);  // (a K_U lane's block leaves u1 alone)
                        }
                    } else {
                        const v3 rad = sky(p);
                        // (explicit address spaces: otherwise the two branches merge into flat atomics on a
                        // selected pointer)
                        unsigned long long fx, fy, fz;
                        to_fixed3(rad, fx, fy, fz);
#if defined(__HIP_DEVICE_COMPILE__)
                        if (PERSIST) wave_sync();  // the epoch may have been advanced by lanes that took a new unit while this one traced
                        if (PERSIST && (slot >> 6) != ust_ld(ust, U_EPOCH)) {
                            // the path outlived its unit, whose tile has been flushed: straight to the frame
                            typedef __attribute__((address_space(1))) unsigned long long glb_u64;
                            glb_u64* out = (glb_u64*)(uintptr_t)(a.sums + 3 * (size_t)rs.pixel);
                            __hip_atomic_fetch_add(out + 0, fx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_fetch_add(out + 1, fy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_fetch_add(out + 2, fz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        } else {
                            typedef __attribute__((address_space(3))) unsigned long long lds_u64;
                            const uint32_t sl = PERSIST ? (slot & 63u) : slot;
                            lds_u64* out = (lds_u64*)(uint32_t)(uintptr_t)(pool.acc + 3 * sl);  // low half of a flat LDS address
                            __hip_atomic_fetch_add(out + 0, fx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_add(out + 1, fy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_add(out + 2, fz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
#else
                        (void)fx; (void)fy; (void)fz;  // host pass of the single-source compile: never runs
#endif
                        ++n_samples;
                        fresh = true;
                    }
                }
                // lanes without a path take the next items of the workgroup's pool
                if (PERSIST) {
                    // at most two passes: the rest of the current unit, then the head of the next one
                    for (int pass = 0; pass < 2; ++pass) {
                        const unsigned long long m_need = __ballot(fresh);
                        if (m_need == 0) break;
                        wave_sync();
                        if (uniform(ust_ld(ust, U_NEXT)) >= uniform(ust_ld(ust, U_NITEMS)) && !(uniform(ust_ld(ust, U_FLAGS)) & 2u)) {
                            unit_next(a, pool.acc, ust, lane);
                        }
                        if (uniform(ust_ld(ust, U_FLAGS)) & 2u) { if (fresh) finished = true; break; }
                        const uint32_t next_item = uniform(ust_ld(ust, U_NEXT)), n_items = uniform(ust_ld(ust, U_NITEMS));
                        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_need >> 32),
                                                                        __builtin_amdgcn_mbcnt_lo((uint32_t)m_need, 0u));
                        const uint32_t avail = n_items - next_item, need = (uint32_t)__popcll(m_need);
                        if (fresh && rank < avail) {
                            const uint32_t item = next_item + rank;
                            slot = (item & 63u) | (uniform(ust_ld(ust, U_EPOCH)) << 6);
                            px = (int)uniform(ust_ld(ust, U_OX)) + (int)(item & 7u); py = (int)uniform(ust_ld(ust, U_OY)) + (int)((item >> 3) & 7u);
                            if (px < a.cam.width && py < a.cam.height) {
                                rs.pixel = (uint32_t)(py * a.cam.width + px);
                                rs.sample = uniform(ust_ld(ust, U_SBASE)) + (item >> 6);
                                rs.event = 0;
                                kind = K_CAM;
                                fresh = false;
                            }  // else: an item outside the frame; ask again
                        }
                        ust_st(ust, U_NEXT, next_item + (need < avail ? need : avail));
                        wave_sync();
                    }
                }
                const unsigned long long m_need = PERSIST ? 0ull : __ballot(fresh);
                if (m_need != 0) {
                    const int leader = __builtin_ctzll(m_need);
                    uint32_t first = 0;
                    if (lane == leader) first = atomicAdd(pool.counter, (uint32_t)__popcll(m_need));
                    first = (uint32_t)__builtin_amdgcn_readlane((int)first, leader);
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_need >> 32),
                                                                    __builtin_amdgcn_mbcnt_lo((uint32_t)m_need, 0u));
                    if (fresh) {
                        const uint32_t item = first + rank;
                        if (item >= pool.total) {
                            finished = true;
                        } else {
                            slot = item & ((1u << pool.shift) - 1u);
                            const int2 org = pool.origin[slot >> 6];
                            px = org.x + (int)(slot & 7u); py = org.y + (int)((slot >> 3) & 7u);
                            if (org.x >= 0 && px < a.cam.width && py < a.cam.height) {
                                rs.pixel = (uint32_t)(py * a.cam.width + px);
                                rs.sample = (uint32_t)a.s0 + (item >> pool.shift);
                                rs.event = 0;
                                kind = K_CAM;
                                fresh = false;
                            }  // else: an item outside the frame; ask again in the next shade phase
                        }
                    }
                }
            }
            CRT_LAP(3);
            // (2) the shared block loop
            for (;;) {
                const unsigned long long m_blk = __ballot(kind != K_NONE);
                if (m_blk == 0) break;
                const int n_blk = __popcll(m_blk);
                if (COUNT && (a.census & 1) && lane == 0) {  // block-loop census: iterations (per wave) and lanes served
                    atomicAdd(&a.counters[13], 1ull); atomicAdd(&a.counters[14], (unsigned long long)n_blk);
                }
                const unsigned long long m_rej = __ballot(kind == K_VEC || kind == K_DISK);
                if (PERSIST && m_rej == m_blk && n_blk <= 32 && a.tail_pass) {
                    // Tail pass.  Only rejection-sampling lanes are left (52% / 79% acceptance per attempt)
                    // and they are few: the WHOLE wave draws for them, G = 2, 4 or 8 consecutive blocks of
                    // each lane's stream at once (lane h: the block of requester h / G at its event
                    // counter + h % G).  A requester takes the first accepted attempt in stream order
                    // and advances its counter past it, exactly what one block per iteration would
                    // have done; the blocks drawn beyond it are dropped.
                    const int g_shift = n_blk <= 8 ? 3 : (n_blk <= 16 ? 2 : 1);
                    const uint32_t rank = 31u & __builtin_amdgcn_mbcnt_hi((uint32_t)(m_blk >> 32),  // requesters: < n_blk <= 32
                                                                          __builtin_amdgcn_mbcnt_lo((uint32_t)m_blk, 0u));
                    uint4* req = (uint4*)(ust + 8);  // 32 requests of 16 bytes behind the unit state
                    if (kind != K_NONE) req[rank] = make_uint4(rs.pixel, rs.sample, rs.event, (uint32_t)kind);
                    wave_sync();  // the requests are read by other lanes
                    const uint32_t hr = (uint32_t)lane >> g_shift, hj = (uint32_t)lane & ((1u << g_shift) - 1u);
                    bool ok = false;
                    float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f;
                    if (hr < (uint32_t)n_blk) {
                        const uint4 q = req[hr];
                        const uint4 rr = philox4x32_10(q.z + hj, q.y, q.x, 0u, rs.k0, rs.k1);
                        const float4 b = make_float4(u01(rr.x), u01(rr.y), u01(rr.z), u01(rr.w));
                        if (q.w == (uint32_t)K_VEC) {
                            v0 = rf11(b.x); v1 = rf11(b.y); v2 = rf11(b.z);
                            const float l = lensq3(v0, v1, v2);
                            ok = 0.0f < l && l <= 1.0f;
                        } else {
                            v0 = rf11(b.x); v1 = rf11(b.y);
                            ok = lensq3(v0, v1, 0.0f) < 1.0f;
                            if (!ok) { v0 = rf11(b.z); v1 = rf11(b.w); ok = lensq3(v0, v1, 0.0f) < 1.0f; }
                        }
                    }
                    const unsigned long long m_ok = __ballot(ok);
                    const uint32_t G = 1u << g_shift;
                    const uint32_t mine = (uint32_t)(m_ok >> ((rank << g_shift) & 63u)) & ((1u << G) - 1u);  // my G attempts (only requesters, rank < 64 >> g_shift, use it)
                    const uint32_t first = (uint32_t)__builtin_ctz(mine | 0x100u);
                    const int src = (int)((((rank << g_shift) + (first & (G - 1u))) & 63u) << 2);
                    const float w0 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(v0)));
                    const float w1 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(v1)));
                    const float w2 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(v2)));
                    if (kind != K_NONE) {
                        if (mine != 0u) {
                            rs.event += first + 1u;
                            if (kind == K_VEC) { u0 = w0; u1 = w1; u2 = w2; u3 = lensq3(u0, u1, u2); }
                            else { dx = w0; dy = w1; }
                            kind = K_NONE;
                        } else {
                            rs.event += G;
                        }
                    }
                } else if (kind != K_NONE) {
                    const float4 b = rs.next_block();
                    if (kind == K_VEC) {
                        u0 = rf11(b.x); u1 = rf11(b.y); u2 = rf11(b.z);
                        u3 = lensq3(u0, u1, u2);
                        if (0.0f < u3 && u3 <= 1.0f) kind = K_NONE;
                    } else if (kind == K_U) {
                        u0 = b.x; kind = K_NONE;
                    } else if (kind == K_CAM) {
                        u0 = b.x - 0.5f; u1 = b.y - 0.5f; u2 = 1.0f - b.z; kind = K_DISK;
                    } else {  // K_DISK: two attempts per block
                        dx = rf11(b.x); dy = rf11(b.y);
                        if (lensq3(dx, dy, 0.0f) < 1.0f) kind = K_NONE;
                        else {
                            dx = rf11(b.z); dy = rf11(b.w);
                            if (lensq3(dx, dy, 0.0f) < 1.0f) kind = K_NONE;
                        }
                    }
                }
            }
            CRT_LAP(4);
            if (shading) {
                // finish: scatter direction, or the new primary ray
                if (type >= 0) {
                    v3 dir;
                    bool absorbed = false;
                    if (type == 3) {
                        dir = p.r.d; absorbed = true;
                    } else if (type <= 1) {
                        const v3 ruv = rsqrt_(u3) * mk(u0, u1, u2);
                        dir = base + scale * ruv;
                        if (type == 0) {  // vec3.h:62-66 near_zero (double threshold)
                            // vec3.h:62-66 near_zero compares in double: a float is below the double 1e-8 exactly when it is below the
                            // smallest float above it -- one fp32 compare per component instead of a conversion and an fp64 compare
                            const float nz = __uint_as_float(0x322BCC78u);  // the float next above 1e-8 (which lies between 0x322BCC77 and it)
                            if ((fabsf(dir.x) < nz) && (fabsf(dir.y) < nz) && (fabsf(dir.z) < nz)) dir = nrm;
                        } else {
                            absorbed = !(dot(dir, nrm) > 0.0f);  // material.h:54
                        }
                    } else {
                        bool do_reflect = cannot;
                        if (!do_reflect) do_reflect = u1 > (1.0f - u0);
                        if (do_reflect) {
                            dir = base - (2.0f * dot(base, nrm)) * nrm;
                        } else {  // refract, vec3.h:138-143
                            const v3 perp = aux * (base + scale * nrm);
                            const v3 par = (-sqrt_(fabsf(1.0f - len2(perp)))) * nrm;
                            dir = perp + par;
                        }
                    }
                    p.r.d = dir;
                    ++p.depth;
                    if (absorbed || p.depth >= a.max_depth) {  // black: nothing to add to the pixel
                        ++n_samples;
                        fresh = true;                          // takes its next item in the next shade phase
                    }
                } else if (!fresh && !finished) {  // new path: render_with_cuda.h:14-28
                    const cam_ref cam(a);  // the camera's 18 floats by scalar loads, here
                    const v3 pixel_sample = (cam.v(cam.p00) + (((float)px + u0) * cam.v(cam.du))) +
                                            (((float)py + u1) * cam.v(cam.dv));
                    const v3 p_in_disk = (cam.v(cam.center) + dx * cam.v(cam.ddu)) + dy * cam.v(cam.ddv);
                    p.r.o = (a.cam.defocus_angle <= 0.0f) ? cam.v(cam.center) : p_in_disk;
                    p.r.d = pixel_sample - p.r.o;
                    p.r.tm = u2;
                    p.thr = mk(1.0f, 1.0f, 1.0f);
                    p.sky_a = 0.5f * (unit(p.r.d).y + 1.0f);
                    p.depth = 0;
                }
                if (!finished && !fresh) {  // begin external_hit for the new segment
                    best_t = __builtin_huge_valf(); best_obj = -1;
                    ray_a = len2(p.r.d);
                    k = 1u; pending = 1u;  // the root, below the sentinel
                    if (NL == 1u) {
                        // the root is the only leaf (k >= NL: sphere phase next): no box is tested
                    } else {
                        inv = rcp3(p.r.d);  // aabb.h:48
                        exact = !finite3(inv);
                        if (IN_LDS) {
                            // the sign BIT picks the plane array (NL is a power of two: the array index is an or).  -0 and a
                            // negative NaN differ from "d < 0", but their reciprocal is not finite: such a ray takes the
                            // literal walk and never uses the offsets.
                            const uint32_t n16 = NL << 4;
                            offx = (uint32_t)(__float_as_int(p.r.d.x) >> 31) & n16;
                            offy = ((uint32_t)(__float_as_int(p.r.d.y) >> 31) & n16) | (2u * n16);
                            offz = ((uint32_t)(__float_as_int(p.r.d.z) >> 31) & n16) | (4u * n16);
                        }
#if CRT_LITERAL_AT_ENTRY
                        if (ROLE != 1) walk_literal();
#endif
                    }
                }
            }
            CRT_LAP(5);
        }
      }
    }
    if (PERSIST && (uniform(ust_ld(ust, U_FLAGS)) & 1u)) unit_flush(a, pool.acc, ust, lane);
    if (ROLE == 1 && lane == 0) xl_st(link.ctl + 8u, 1u);  // tells this owner's walker that nothing more will be posted
    if (ROLE == 1 && COUNT && (a.census & 1) && lane == 0) atomicAdd(&a.counters[3], c_waits);
    if (timing && lane == 0) {
        for (int i = 0; i < 6; ++i) atomicAdd(&a.counters[16 + i], tc[i]);
        const unsigned long long t = wall_clock64();
        atomicAdd(&a.counters[25], t); atomicMax(&a.counters[26], t);
        const uint32_t wid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        if (PERSIST && wid < 8192u) a.counters[CRT_TIMELINE_BASE + 3 * wid + 2] = t;
    }
#undef CRT_LAP
}

// ---------------------------------------------------------------------------------------
// SCHED: 0 = simple per-lane loop, 1 = wave-scheduled with a workgroup item pool, 2 = wave-scheduled
// persistent waves, 3 = persistent owner waves + walker waves exchanging walk requests through LDS rings
template <int WORLD, bool IN_LDS, bool COUNT, int BLOCK, int SCHED>
__global__ __launch_bounds__(BLOCK) void render_kernel(const crt_render_args a) {
    extern __shared__ float4 smem[];
    constexpr int NW = BLOCK / 64;
    const float4* pairs = (const float4*)a.pairs;
    const float4* spheres = (const float4*)a.spheres;
    const float4* mats = (const float4*)a.mats;
    // LDS image of the scene, in float4 units: the scheduled kernel keeps the sign-sorted plane
    // arrays (6 * L) and the leaf pairs' sphere tags, the simple loop the pair records (4 * L)
    const int L = a.num_leaves;
    const int leaf_parents = L >> 1, n_tagrec = L - leaf_parents;  // pairs [leaf_parents, L) hold sphere tags
    const int top_n = (!IN_LDS && WORLD == 1 && SCHED) ? a.top_pairs : 0;
    const int np = !(IN_LDS && WORLD == 1) ? 3 * top_n : (SCHED ? 6 * L + (n_tagrec + 1) / 2 : 4 * L);
    const int nsph = (IN_LDS && !(WORLD == 1 && SCHED && !CRT_SCHED_SPHERES_IN_LDS)) ? 2 * a.n : 0;
    walk_scene ws;
    ws.pairs = pairs; ws.axes = nullptr; ws.leaf_tags = nullptr; ws.L = (uint32_t)L; ws.top = nullptr; ws.top_n = 0;
    if (top_n > 0) {
        for (int k = threadIdx.x; k < top_n; k += BLOCK) {
            smem[k] = pairs[4 * k]; smem[top_n + k] = pairs[4 * k + 1]; smem[2 * top_n + k] = pairs[4 * k + 2];
        }
        ws.top = smem; ws.top_n = (uint32_t)top_n;
    }
    if (IN_LDS) {
        // material records are read once per ray and stay in HBM/L2
        float4* s_pairs = smem;
        float4* s_spheres = smem + np;
        if (WORLD == 1 && SCHED) {
            int2* s_tags = (int2*)(smem + 6 * L);
            for (int k = threadIdx.x; k < L; k += BLOCK) {
                const float4 q0 = pairs[4 * k], q1 = pairs[4 * k + 1], q2 = pairs[4 * k + 2], q3 = pairs[4 * k + 3];
                s_pairs[k] = q0;         s_pairs[L + k] = make_float4(q0.y, q0.x, q0.w, q0.z);
                s_pairs[2 * L + k] = q1; s_pairs[3 * L + k] = make_float4(q1.y, q1.x, q1.w, q1.z);
                s_pairs[4 * L + k] = q2; s_pairs[5 * L + k] = make_float4(q2.y, q2.x, q2.w, q2.z);
                if (k >= leaf_parents) s_tags[k - leaf_parents] = make_int2(__float_as_int(q3.x), __float_as_int(q3.y));
            }
            ws.axes = s_pairs;
            ws.leaf_tags = s_tags - leaf_parents;
        } else {
            for (int i = threadIdx.x; i < np; i += BLOCK) s_pairs[i] = pairs[i];
            pairs = s_pairs;
        }
        if (nsph) {
            for (int i = threadIdx.x; i < nsph; i += BLOCK) s_spheres[i] = spheres[i];
            spheres = s_spheres;
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t n_rays = 0, n_samples = 0, n_pairs = 0, n_tests = 0;

    if (WORLD == 1 && SCHED == 2) {
        // per wave: 192 accumulators + 8 dwords of unit state (all zero at the start) + 32 block requests
        unsigned long long* acc = (unsigned long long*)(smem + np + nsph) + 260 * wave;
        for (int i = lane; i < 260; i += 64) acc[i] = 0ull;
        __syncthreads();  // the scene image is complete; from here on the waves run independently
        wg_pool pool;
        pool.acc = acc; pool.counter = nullptr; pool.origin = nullptr; pool.shift = 6; pool.total = 0;
        trace_bvh_scheduled<IN_LDS, COUNT, true>(a, ws, spheres, mats, pool, n_rays, n_samples, n_pairs, n_tests);
    } else if (WORLD == 1 && SCHED == 3 && IN_LDS) {
        // walker / owner waves (see walker_loop): the first n_walkers waves walk, the others own the paths
        const int n_walk = a.n_walkers, n_own = NW - n_walk;
        float4* xb = smem + np + nsph + 130 * n_own;  // behind the owners' accumulator areas (260 u64 = 130 float4 each)
        for (int i = threadIdx.x; i < n_own * (int)CRT_XCHG_FLOAT4_PER_OWNER; i += BLOCK) xb[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        unsigned long long* acc = (unsigned long long*)(smem + np + nsph) + 260 * (wave >= n_walk ? wave - n_walk : 0);
        if (wave >= n_walk)
            for (int i = lane; i < 260; i += 64) acc[i] = 0ull;
        __syncthreads();  // scene image, rings and mailboxes are set up; from here on only the exchange protocol orders the waves
        if (wave < n_walk) {
            walker_loop<COUNT>(a, ws, lds_off(xb), (uint32_t)wave, (uint32_t)n_walk, (uint32_t)n_own, n_pairs);
        } else {
            owner_link link;
            link.q0 = lds_off(xb) + (uint32_t)(wave - n_walk) * (16u * CRT_XCHG_FLOAT4_PER_OWNER);
            link.q1 = link.q0 + 16u * CRT_RING_CAP;
            link.ctl = link.q0 + 32u * CRT_RING_CAP;
            link.mbox = link.ctl + 16u;
            wg_pool pool;
            pool.acc = acc; pool.counter = nullptr; pool.origin = nullptr; pool.shift = 6; pool.total = 0;
            trace_bvh_scheduled<IN_LDS, COUNT, true, 1>(a, ws, spheres, mats, pool, n_rays, n_samples, n_pairs, n_tests, link);
        }
    } else if (WORLD == 1 && SCHED == 1) {
        const int T = a.tiles_per_wg;  // power of two
        unsigned long long* acc = (unsigned long long*)(smem + np + nsph);
        int2* origin = (int2*)(acc + 192 * T);
        uint32_t* counter = (uint32_t*)(origin + T);
        for (int i = threadIdx.x; i < 192 * T; i += BLOCK) acc[i] = 0ull;
        if ((int)threadIdx.x < T) {
            const int share_tile = blockIdx.x * T + threadIdx.x;
            const int tile = share_tile * a.tile_stride + a.tile_offset;
            const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
            origin[threadIdx.x] = make_int2(share_tile < a.n_tiles ? tx * 8 : -1, ty * 8);
        }
        if (threadIdx.x == 0) *counter = 0u;
        __syncthreads();
        wg_pool pool;
        pool.acc = acc; pool.counter = counter; pool.origin = origin;
        pool.shift = 6 + (31 - __builtin_clz((unsigned)T));
        pool.total = (uint32_t)a.ns << pool.shift;
        trace_bvh_scheduled<IN_LDS, COUNT, false>(a, ws, spheres, mats, pool, n_rays, n_samples, n_pairs, n_tests);
        __syncthreads();
        for (int i = threadIdx.x; i < 192 * T; i += BLOCK) {
            const unsigned long long q = acc[i];
            const int s = i / 3, c = i - 3 * s;
            const int2 org = origin[s >> 6];
            const int px = org.x + (s & 7), py = org.y + ((s >> 3) & 7);
            if (q != 0ull && org.x >= 0 && px < a.cam.width && py < a.cam.height)
                atomicAdd(&a.sums[3 * ((size_t)py * a.cam.width + px) + c], q);
        }
    } else {
        // simple loop: a wave owns one tile and one chunk of its samples, a lane one pixel (brute-force list world, and
        // the A/B reference of the scheduler).  Small frames get several chunks per tile so that the launch still fills
        // the chip (C1: 1 407 tiles x 10 samples -> 14 070 waves instead of 1 407 serial ones); the sums are integers,
        // so the split does not show in the frame.
        if (IN_LDS) __syncthreads();
        const int unit = blockIdx.x * NW + wave;
        const int share_tile = unit / a.simple_chunks, chunk = unit - share_tile * a.simple_chunks;
        const int s_first = (int)((long long)chunk * a.ns / a.simple_chunks);
        const int s_end = (int)((long long)(chunk + 1) * a.ns / a.simple_chunks);
        const int tile = share_tile * a.tile_stride + a.tile_offset;
        const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
        const int px = tx * 8 + (lane & 7), py = ty * 8 + (lane >> 3);
        const bool valid = (share_tile < a.n_tiles) && (px < a.cam.width) && (py < a.cam.height) && (s_end > s_first);
        const uint32_t pixel_index = (uint32_t)(py * a.cam.width + px);
        if (valid) {
            unsigned long long sx = 0, sy = 0, sz = 0;
            stream rs; rs.k0 = a.key0; rs.k1 = a.key1;
            path_t p;
            int s = s_first;
            start_path(p, rs, a, px, py, pixel_index, (uint32_t)(a.s0 + s));
            for (;;) {
                hit_t h;
                if (WORLD == 1) h = closest_hit_bvh<IN_LDS, COUNT>(pairs, spheres, a.num_leaves, p.r, n_pairs, n_tests);
                else h = closest_hit_list<IN_LDS, COUNT>(spheres, a.n, p.r, n_tests);
                ++n_rays;
                bool ended;
                v3 rad = mk(0.0f, 0.0f, 0.0f);
                if (h.obj >= 0) {
                    const sphere_rec sp = load_sphere<IN_LDS>(spheres, h.obj);
                    const float4 mrec = mats[sp.mat & CRT_MAT_ID_MASK];
                    const bool cont = shade(p, rs, sp, mrec, h.t);
                    ++p.depth;
                    ended = !cont || (p.depth >= a.max_depth);  // absorbed or loop exhausted: black
                } else {
                    rad = sky(p);
                    ended = true;
                }
                if (ended) {
                    sx += to_fixed(rad.x); sy += to_fixed(rad.y); sz += to_fixed(rad.z);
                    ++n_samples;
                    ++s;
                    if (s >= s_end) break;
                    start_path(p, rs, a, px, py, pixel_index, (uint32_t)(a.s0 + s));
                }
            }
            unsigned long long* out = a.sums + 3 * (size_t)pixel_index;
            atomicAdd(&out[0], sx); atomicAdd(&out[1], sy); atomicAdd(&out[2], sz);
        }
    }

    // statistics: one atomic per wave and counter
    {
        const uint32_t r = wave_sum(n_rays), sm = wave_sum(n_samples);
        // (the persistent kernel's 6144 waves finish spread over the launch's drain: one stripe, and its code stays as tuned)
        unsigned long long* stats = a.counters + CRT_STATS_BASE + (SCHED >= 2 ? 0 : 4 * (blockIdx.x & (CRT_STAT_STRIPES - 1)));
        if (lane == 0 && r) { atomicAdd(&stats[0], (unsigned long long)r); atomicAdd(&stats[1], (unsigned long long)sm); }
        if (COUNT) {
            const uint32_t pv = wave_sum(n_pairs), stt = wave_sum(n_tests);
            if (lane == 0) { atomicAdd(&stats[2], (unsigned long long)pv); atomicAdd(&stats[3], (unsigned long long)stt); }
        }
    }
}

// frame resolve: fixed-point sums -> fp32 radiance means, fb = float(double(sum) * (2^-40 / spp))
__global__ __launch_bounds__(256) void resolve_kernel(const unsigned long long* sums, float* fb, size_t count, double scale) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) fb[i] = (float)((double)(long long)sums[i] * scale);
}

__global__ __launch_bounds__(256) void sum_frames_kernel(unsigned long long* dst, const crt_frame_list src, size_t count) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    unsigned long long v = dst[i];
    for (int j = 0; j < src.n; ++j) v += src.p[j][i];
    dst[i] = v;
}

// single paths, for fine-grained parity checks (same device functions as the render kernel)
template <int WORLD>
__global__ __launch_bounds__(64) void trace_samples_kernel(const crt_render_args a, int count, const int* pxs,
                                                           const int* pys, const int* samples, float* ray7,
                                                           float* rgb, int* segments) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= count) return;
    const float4* pairs = (const float4*)a.pairs;
    const float4* spheres = (const float4*)a.spheres;
    const float4* mats = (const float4*)a.mats;
    const int px = pxs[i], py = pys[i];
    stream rs; rs.k0 = a.key0; rs.k1 = a.key1;
    path_t p;
    start_path(p, rs, a, px, py, (uint32_t)(py * a.cam.width + px), (uint32_t)samples[i]);
    ray7[7 * i + 0] = p.r.o.x; ray7[7 * i + 1] = p.r.o.y; ray7[7 * i + 2] = p.r.o.z;
    ray7[7 * i + 3] = p.r.d.x; ray7[7 * i + 4] = p.r.d.y; ray7[7 * i + 5] = p.r.d.z;
    ray7[7 * i + 6] = p.r.tm;
    v3 rad = mk(0.0f, 0.0f, 0.0f);
    int segs = 0;
    uint32_t d0 = 0, d1 = 0;
    for (;;) {
        hit_t h;
        if (WORLD == 1) h = closest_hit_bvh<false, false>(pairs, spheres, a.num_leaves, p.r, d0, d1);
        else h = closest_hit_list<false, false>(spheres, a.n, p.r, d1);
        ++segs;
        if (h.obj >= 0) {
            const sphere_rec sp = load_sphere<false>(spheres, h.obj);
            const bool cont = shade(p, rs, sp, mats[sp.mat & CRT_MAT_ID_MASK], h.t);
            ++p.depth;
            if (!cont || p.depth >= a.max_depth) break;
        } else {
            rad = sky(p);
            break;
        }
    }
    rgb[3 * i + 0] = rad.x; rgb[3 * i + 1] = rad.y; rgb[3 * i + 2] = rad.z;
    segments[i] = segs;
}

template <int WORLD, bool IN_LDS, bool COUNT, int BLOCK, int SCHED>
hipError_t launch_sched(const crt_render_args& a, int grid, size_t lds_bytes, hipStream_t stream) {
    auto k = render_kernel<WORLD, IN_LDS, COUNT, BLOCK, SCHED>;
    if (lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(BLOCK), lds_bytes, stream, a);
    return hipGetLastError();
}

template <int WORLD, bool IN_LDS, bool COUNT, int BLOCK>
hipError_t launch_one(const crt_render_args& a, int grid, size_t lds_bytes, hipStream_t stream) {
    if (WORLD == 1 && IN_LDS && a.sched == 3 && a.persist && (BLOCK == 768 || BLOCK == 1024))
        return launch_sched<WORLD, IN_LDS, COUNT, BLOCK, (WORLD == 1 && IN_LDS && (BLOCK == 768 || BLOCK == 1024) ? 3 : 0)>(a, grid, lds_bytes, stream);
    if (WORLD == 1 && a.sched && a.persist) return launch_sched<WORLD, IN_LDS, COUNT, BLOCK, (WORLD == 1 ? 2 : 0)>(a, grid, lds_bytes, stream);
    if (WORLD == 1 && a.sched) return launch_sched<WORLD, IN_LDS, COUNT, BLOCK, (WORLD == 1 ? 1 : 0)>(a, grid, lds_bytes, stream);
    return launch_sched<WORLD, IN_LDS, COUNT, BLOCK, 0>(a, grid, lds_bytes, stream);
}

template <int WORLD, bool IN_LDS, bool COUNT>
hipError_t launch_block(const crt_render_args& a, int block, int grid, size_t lds_bytes, hipStream_t stream) {
    switch (block) {
        case 256: return launch_one<WORLD, IN_LDS, COUNT, 256>(a, grid, lds_bytes, stream);
        case 512: return launch_one<WORLD, IN_LDS, COUNT, 512>(a, grid, lds_bytes, stream);
        case 768: return launch_one<WORLD, IN_LDS, COUNT, 768>(a, grid, lds_bytes, stream);
        case 1024: return launch_one<WORLD, IN_LDS, COUNT, 1024>(a, grid, lds_bytes, stream);
    }
    return hipErrorInvalidValue;
}

#if !CRT_FAST
// Proof by exhaustion for rcp_(), sqrt_() and rsqrt_(): every one of the 2^32 arguments against the compiler's
// correctly rounded division and square root.  Per function f = 0, 1, 2: out[3f] = arguments whose results differ (as
// bit patterns; two NaNs count as equal), out[3f + 1] = the smallest such argument's bits (else ~0), out[3f + 2] =
// arguments that took the short sequence.
__global__ __launch_bounds__(256) void math_selftest_kernel(unsigned long long* out) {
    const unsigned long long n_threads = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long bad[3] = {0, 0, 0}, plain[3] = {0, 0, 0}, first[3] = {~0ull, ~0ull, ~0ull};
    unsigned long long fbad = 0, fplain = 0, ffirst = ~0ull;  // function 3: the fixed-point conversion of a radiance
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += n_threads) {
        const float x = __uint_as_float((uint32_t)i);
        const float want[3] = {1.0f / x, sqrtf(x), 1.0f / sqrtf(x)};
        const float got[3] = {rcp_(x), sqrt_(x), rsqrt_(x)};
        const bool is_plain[3] = {rcp_plain(x), sqrt_plain(x), sqrt_plain(x)};
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            const bool same = (__float_as_uint(want[f]) == __float_as_uint(got[f])) | ((want[f] != want[f]) & (got[f] != got[f]));
            if (!same) { ++bad[f]; if (i < first[f]) first[f] = i; }
            if (is_plain[f]) ++plain[f];
        }
        // (a radiance of -inf or below -2^23 overflows the reference conversion's int64: undefined there, not compared)
        const float v = fixed_scaled(x);
        if (fixed_plain(v)) {
            ++fplain;
            if (fixed_short(v) != to_fixed(x)) { ++fbad; if (i < ffirst) ffirst = i; }
        }
    }
    if (fbad) { atomicAdd(&out[9], fbad); atomicMin(&out[10], ffirst); }
    atomicAdd(&out[11], fplain);
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        if (bad[f]) { atomicAdd(&out[3 * f], bad[f]); atomicMin(&out[3 * f + 1], first[f]); }
        atomicAdd(&out[3 * f + 2], plain[f]);
    }
}
#endif

}  // inline namespace arith_exact / arith_fast
}  // namespace

#if !CRT_FAST
hipError_t crt_launch_resolve(const void* sums, float* fb, size_t count, int spp, hipStream_t stream) {
    const double scale = ldexp(1.0, -40) / (double)spp;
    hipLaunchKernelGGL(resolve_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream,
                       (const unsigned long long*)sums, fb, count, scale);
    return hipGetLastError();
}

// dst[i] += src[0][i] + ... + src[n-1][i]: the frames of a group's other members (crt_group_render), integer sums
hipError_t crt_launch_sum_frames(unsigned long long* dst, const crt_frame_list& src, size_t count, hipStream_t stream) {
    hipLaunchKernelGGL(sum_frames_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, dst, src, count);
    return hipGetLastError();
}
hipError_t crt_launch_math_selftest(unsigned long long* out12, hipStream_t stream) {
    hipLaunchKernelGGL(math_selftest_kernel, dim3(256 * 16), dim3(256), 0, stream, out12);
    return hipGetLastError();
}
#define crt_launch_render_tu crt_launch_render
#define crt_launch_trace_samples_tu crt_launch_trace_samples
#else
#define crt_launch_render_tu crt_launch_render_fast
#define crt_launch_trace_samples_tu crt_launch_trace_samples_fast
#endif

// Host-visible launchers (called from crt_api.cpp)
hipError_t crt_launch_render_tu(const crt_render_args& a, int world, bool in_lds, bool count, int block, int grid,
                             size_t lds_bytes, hipStream_t stream) {
    if (world == 1) {
        if (in_lds) return count ? launch_block<1, true, true>(a, block, grid, lds_bytes, stream)
                                 : launch_block<1, true, false>(a, block, grid, lds_bytes, stream);
        return count ? launch_block<1, false, true>(a, block, grid, lds_bytes, stream)
                     : launch_block<1, false, false>(a, block, grid, lds_bytes, stream);
    }
    if (in_lds) return count ? launch_block<0, true, true>(a, block, grid, lds_bytes, stream)
                             : launch_block<0, true, false>(a, block, grid, lds_bytes, stream);
    return count ? launch_block<0, false, true>(a, block, grid, lds_bytes, stream)
                 : launch_block<0, false, false>(a, block, grid, lds_bytes, stream);
}

hipError_t crt_launch_trace_samples_tu(const crt_render_args& a, int world, int count, const int* px, const int* py,
                                    const int* samples, float* ray7, float* rgb, int* segments, hipStream_t stream) {
    const int grid = (count + 63) / 64;
    if (world == 1) hipLaunchKernelGGL(trace_samples_kernel<1>, dim3(grid), dim3(64), 0, stream, a, count, px, py, samples, ray7, rgb, segments);
    else hipLaunchKernelGGL(trace_samples_kernel<0>, dim3(grid), dim3(64), 0, stream, a, count, px, py, samples, ray7, rgb, segments);
    return hipGetLastError();
}
