// crt_api.cpp -- the C ABI of include/crt_hip.h: context, scene upload, BVH, camera, launches.
// Product code; never touches oracle/.  Fails with an error code when no HIP device or no
// gfx950 code object is usable -- there is no CPU fallback.
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/crt_hip.h"
#include "crt_host.h"
#include "crt_bvh.h"
#include "crt_launch.h"

struct crt_context {
    int device = 0;
    uint64_t seed = 1;
    std::string err;
    // host copies
    int n = 0, m = 0;
    std::vector<float> center, motion, radius;
    int num_leaves = 0;
    crt_camera cam{};
    bool have_cam = false;
    // device
    void* d_pairs = nullptr;
    crt_bvh_node* d_nodes = nullptr;  // the node array (bvh_world::d_nodes, bvh.h:131), read back by crt_get_bvh_nodes
    double bvh_seconds = 0;
    int bvh_host = 0;  // CRT_BVH_HOST=1: build on the host (crt_host.cpp) instead of the GPU (crt_bvh.hip)
    void* d_spheres = nullptr;
    void* d_mats = nullptr;
    float* d_fb = nullptr;
    unsigned long long* d_sums = nullptr;
    size_t fb_floats = 0;
    unsigned long long* d_counters = nullptr;
    unsigned long long* d_static = nullptr;  // this GPU's static scratch (crt_static.hip), resolved by the first crt_render_static
    int *d_px = nullptr, *d_py = nullptr, *d_smp = nullptr, *d_seg = nullptr;
    float *d_ray7 = nullptr, *d_rgb = nullptr;
    int dbg_cap = 0;
    static constexpr int EV_RING = 64;  // event pairs of the most recent launches
    hipEvent_t ev_ring[2 * EV_RING] = {};
    bool ev_timed[EV_RING] = {};  // false: the launch was captured into a graph, its events were not recorded
    long launches = 0;
    bool launched = false;
    double wall_seconds = 0;
    unsigned long long* last_counters = nullptr;  // counters of the last launch (crt_get_stats)
    int arith = CRT_ARITH_EXACT;
    char last_kernel[160] = "none";
    // tuning
    int block = 0;  // 0: chosen per launch from the LDS footprint (CRT_BLOCK overrides)
    int tiles_per_wg = 0;  // 0 = auto
    bool count_stats = false;
    int census = 0;  // 1: scheduler counts, 2: per-phase cycles (internal, tools/tune.py)
    size_t lds_limit = 126 * 1024;  // scene image that still leaves room for one 16-wave workgroup per CU (55 KB: two 12-wave ones)
    int sched = 1, leaf_num = 2, shade_num = 12, run_shift = 1;
    int tail_pass = 1;
    int walkers = 4, refill_min = 16, owner_sleep = 4, walker_backoff = 2, walker_prio = 1, walker_min_busy = 0, walker_patience = 8;  // SCHED=3 (CRT_SCHED=3): walker waves per workgroup, refill threshold, poll interval
    int top_pairs_max = 1024;  // CRT_TOP_PAIRS
    int persist = 1, unit_samples = 0;  // persistent waves; samples per coarse unit (0: chosen per launch)
    int fine_samples = 0, fine_pct = 30;  // samples per fine unit (0: a quarter of the coarse ones), share of the samples cut into fine units
    int num_cus = 256;
};

namespace {

std::string g_create_error;

int fail(crt_context* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg; else g_create_error = msg;
    return code;
}

#define CRT_HIP_TRY(ctx, expr)                                                                       \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return fail(ctx, CRT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));        \
    } while (0)

float as_float(int32_t bits) { float f; memcpy(&f, &bits, 4); return f; }

// a complete node array (heap order, L leaves) from the host: upload it and derive the child-pair records on the device
int install_nodes(crt_context* ctx, const crt_bvh_node* nodes, int L) {
    if (ctx->d_pairs) { (void)hipFree(ctx->d_pairs); ctx->d_pairs = nullptr; }
    if (ctx->d_nodes) { (void)hipFree(ctx->d_nodes); ctx->d_nodes = nullptr; }
    ctx->num_leaves = 0;
    const size_t count = 2 * size_t(L) - 1;
    hipError_t e = hipMalloc((void**)&ctx->d_nodes, count * sizeof(crt_bvh_node));
    if (e == hipSuccess) e = hipMalloc(&ctx->d_pairs, 16 * size_t(L) * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(ctx->d_nodes, nodes, count * sizeof(crt_bvh_node), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = crt_gpu_pairs_from_nodes(ctx->d_nodes, (float*)ctx->d_pairs, L, nullptr);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) {
        if (ctx->d_pairs) { (void)hipFree(ctx->d_pairs); ctx->d_pairs = nullptr; }
        if (ctx->d_nodes) { (void)hipFree(ctx->d_nodes); ctx->d_nodes = nullptr; }
        return fail(ctx, CRT_ERR_HIP, std::string("BVH upload: ") + hipGetErrorString(e));
    }
    ctx->num_leaves = L;
    return CRT_OK;
}

}  // namespace

extern "C" {

const char* crt_last_error(const crt_context* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int crt_create(int32_t device, uint64_t seed, crt_context** out) {
    if (!out) return fail(nullptr, CRT_ERR_INVALID_ARGUMENT, "crt_create: out is NULL");
    *out = nullptr;
    // CRT_TRACE_INIT=1: wall-clock laps of the set-up calls on stderr (where does context creation go?)
    const bool trace = getenv("CRT_TRACE_INIT") && atoi(getenv("CRT_TRACE_INIT")) != 0;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!trace) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "crt_create: %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
        t_last = t;
    };
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    lap("hipGetDeviceCount");
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, CRT_ERR_NO_DEVICE, std::string("crt_create: no HIP device (") + hipGetErrorString(e) + ")");
    if (device < 0 || device >= count) return fail(nullptr, CRT_ERR_INVALID_ARGUMENT, "crt_create: device index out of range");
    CRT_HIP_TRY(nullptr, hipSetDevice(device));
    lap("hipSetDevice");
    hipDeviceProp_t prop;
    CRT_HIP_TRY(nullptr, hipGetDeviceProperties(&prop, device));
    lap("hipGetDeviceProperties");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, CRT_ERR_NO_DEVICE, std::string("crt_create: device is ") + prop.gcnArchName + ", this library carries gfx950 code only");
    crt_context* ctx = new crt_context;
    ctx->device = device;
    ctx->seed = seed;
    if (const char* b = getenv("CRT_BLOCK")) { int v = atoi(b); if (v == 256 || v == 512 || v == 768 || v == 1024) ctx->block = v; }
    if (const char* b = getenv("CRT_TILES")) { int v = atoi(b); if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) ctx->tiles_per_wg = v; }
    if (const char* b = getenv("CRT_SCHED")) { const int v = atoi(b); ctx->sched = v == 3 ? 3 : (v != 0); }
    if (const char* b = getenv("CRT_WALKERS")) { int v = atoi(b); if (v >= 1 && v <= 8) ctx->walkers = v; }
    if (const char* b = getenv("CRT_REFILL_MIN")) { int v = atoi(b); if (v >= 1 && v <= 64) ctx->refill_min = v; }
    if (const char* b = getenv("CRT_OWNER_SLEEP")) { int v = atoi(b); if (v >= 0 && v <= 64) ctx->owner_sleep = v; }
    if (const char* b = getenv("CRT_WALKER_BACKOFF")) { int v = atoi(b); if (v >= 0 && v <= 64) ctx->walker_backoff = v; }
    if (const char* b = getenv("CRT_WALKER_PRIO")) ctx->walker_prio = atoi(b) != 0;
    if (const char* b = getenv("CRT_WALKER_MIN_BUSY")) { int v = atoi(b); if (v >= 0 && v <= 64) ctx->walker_min_busy = v; }
    if (const char* b = getenv("CRT_WALKER_PATIENCE")) { int v = atoi(b); if (v >= 0 && v <= 4096) ctx->walker_patience = v; }
    if (const char* b = getenv("CRT_RUN_SHIFT")) { int v = atoi(b); if (v >= 0 && v <= 6) ctx->run_shift = v; }
    if (const char* b = getenv("CRT_LDS_LIMIT")) ctx->lds_limit = size_t(atol(b));
    if (const char* b = getenv("CRT_LEAF_NUM")) { int v = atoi(b); if (v >= 0 && v <= 16) ctx->leaf_num = v; }
    if (const char* b = getenv("CRT_SHADE_NUM")) { int v = atoi(b); if (v >= 0 && v <= 16) ctx->shade_num = v; }
    if (const char* b = getenv("CRT_PERSIST")) ctx->persist = atoi(b) != 0;
    if (const char* b = getenv("CRT_BVH_HOST")) ctx->bvh_host = atoi(b) != 0;
    if (const char* b = getenv("CRT_TOP_PAIRS")) { int v = atoi(b); if (v >= 0 && v <= 2048) ctx->top_pairs_max = v; }
    if (const char* b = getenv("CRT_TAIL_PASS")) ctx->tail_pass = atoi(b) != 0;
    if (const char* b = getenv("CRT_UNIT_SAMPLES")) { int v = atoi(b); if (v >= 0 && v <= 64) ctx->unit_samples = v; }
    if (const char* b = getenv("CRT_UNIT_FINE")) { int v = atoi(b); if (v >= 0 && v <= 64) ctx->fine_samples = v; }
    if (const char* b = getenv("CRT_UNIT_FINE_PCT")) { int v = atoi(b); if (v >= 0 && v <= 100) ctx->fine_pct = v; }
    if (const char* b = getenv("CRT_ARITH")) ctx->arith = strcmp(b, "fast") == 0 ? CRT_ARITH_FAST : CRT_ARITH_EXACT;
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hipMalloc((void**)&ctx->d_counters, CRT_COUNTER_WORDS * sizeof(unsigned long long)) != hipSuccess) {
        delete ctx;
        return fail(nullptr, CRT_ERR_HIP, "crt_create: device allocation failed");
    }
    lap("hipMalloc(counters)");
    if (hipMemset(ctx->d_counters, 0, CRT_COUNTER_WORDS * sizeof(unsigned long long)) != hipSuccess) {
        (void)hipFree(ctx->d_counters);
        delete ctx;
        return fail(nullptr, CRT_ERR_HIP, "crt_create: device memset failed");
    }
    lap("hipMemset(counters)");
    for (hipEvent_t& e : ctx->ev_ring)
        if (hipEventCreate(&e) != hipSuccess) {
            crt_destroy(ctx);
            return fail(nullptr, CRT_ERR_HIP, "crt_create: event creation failed");
        }
    lap("hipEventCreate x128");
    *out = ctx;
    return CRT_OK;
}

int crt_destroy(crt_context* ctx) {
    if (!ctx) return CRT_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    void* bufs[] = {ctx->d_pairs, ctx->d_nodes, ctx->d_spheres, ctx->d_mats, ctx->d_fb, ctx->d_sums, ctx->d_counters, ctx->d_px, ctx->d_py,
                    ctx->d_smp, ctx->d_seg, ctx->d_ray7, ctx->d_rgb};
    for (void* b : bufs) if (b) (void)hipFree(b);
    for (hipEvent_t e : ctx->ev_ring) if (e) (void)hipEventDestroy(e);
    delete ctx;
    return CRT_OK;
}

int crt_set_seed(crt_context* ctx, uint64_t seed) {
    if (!ctx) return CRT_ERR_INVALID_ARGUMENT;
    ctx->seed = seed;
    return CRT_OK;
}

int crt_set_count_stats(crt_context* ctx, int32_t enable) {
    if (!ctx) return CRT_ERR_INVALID_ARGUMENT;
    ctx->count_stats = enable != 0;
    ctx->census = enable == 2 ? 1 : (enable == 3 ? 2 : 0);  // internal: scheduler census / phase cycles for tools/tune.py
    return CRT_OK;
}

int crt_set_arith(crt_context* ctx, int32_t arith) {
    if (!ctx) return CRT_ERR_INVALID_ARGUMENT;
    if (arith != CRT_ARITH_EXACT && arith != CRT_ARITH_FAST) return fail(ctx, CRT_ERR_INVALID_ARGUMENT, "crt_set_arith: unknown mode");
    ctx->arith = arith;
    return CRT_OK;
}

int crt_selftest_exact_math(crt_context* ctx, uint64_t* out12) {
    if (!ctx) return CRT_ERR_INVALID_ARGUMENT;
    if (!out12) return fail(ctx, CRT_ERR_INVALID_ARGUMENT, "crt_selftest_exact_math: out12 is NULL");
    CRT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    unsigned long long* d = nullptr;
    const unsigned long long init[12] = {0ull, ~0ull, 0ull, 0ull, ~0ull, 0ull, 0ull, ~0ull, 0ull, 0ull, ~0ull, 0ull};
    unsigned long long res[12];
    CRT_HIP_TRY(ctx, hipMalloc((void**)&d, sizeof init));
    hipError_t e = hipMemcpy(d, init, sizeof init, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = crt_launch_math_selftest(d, nullptr);
    if (e == hipSuccess) e = hipMemcpy(res, d, sizeof res, hipMemcpyDeviceToHost);  // waits for the kernel
    (void)hipFree(d);
    if (e != hipSuccess) return fail(ctx, CRT_ERR_HIP, std::string("crt_selftest_exact_math: ") + hipGetErrorString(e));
    for (int i = 0; i < 12; ++i) out12[i] = res[i];
    return CRT_OK;
}

const char* crt_last_kernel_name(const crt_context* ctx) { return ctx ? ctx->last_kernel : "none"; }

#ifndef CRT_BUILD_ID
#define CRT_BUILD_ID "unknown"
#endif
const char* crt_build_id(void) { return CRT_BUILD_ID; }

int crt_upload_scene(crt_context* ctx, int32_t n, const float* center, const float* motion, const float* radius,
                     const int32_t* mat_id, int32_t m, const int32_t* mat_type, const float* mat_albedo,
                     const float* mat_param) {
    if (!ctx) return CRT_ERR_INVALID_ARGUMENT;
    if (n < 1 || m < 1 || !center || !motion || !radius || !mat_id || !mat_type || !mat_albedo || !mat_param)
        return fail(ctx, CRT_ERR_INVALID_ARGUMENT, "crt_upload_scene: empty scene or NULL array");
    if (m > CRT_MAT_ID_MASK) return fail(ctx, CRT_ERR_TOO_LARGE, "crt_upload_scene: too many materials");
    for (int i = 0; i < m; ++i)
        if (mat_type[i] < 0 || mat_type[i] > CRT_MATERIAL_BASE) return fail(ctx, CRT_ERR_INVALID_ARGUMENT, "crt_upload_scene: unknown material type");
    for (int i = 0; i < n; ++i)
        if (mat_id[i] < 0 || mat_id[i] >= m) return fail(ctx, CRT_ERR_INVALID_ARGUMENT, "crt_upload_scene: mat_id out of range");
    CRT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    CRT_HIP_TRY(ctx, hipDeviceSynchronize());
    std::vector<float> sph(8 * size_t(n)), mats(4 * size_t(m));
    for (int i = 0; i < n; ++i) {
        float* s = &sph[8 * size_t(i)];
        s[0] = center[3 * i]; s[1] = center[3 * i + 1]; s[2] = center[3 * i + 2];
        s[3] = fmaxf(0.0f, radius[i]);  // sphere.h:11 radius(fmaxf(0, radius))
        s[4] = motion[3 * i]; s[5] = motion[3 * i + 1]; s[6] = motion[3 * i + 2];
        s[7] = as_float(int32_t(uint32_t(mat_id[i]) | (uint32_t(mat_type[mat_id[i]]) << CRT_MAT_TYPE_SHIFT)));
    }
    for (int i = 0; i < m; ++i) {
        float* q = &mats[4 * size_t(i)];
        if (mat_type[i] == CRT_DIELECTRIC) {
            q[0] = mat_param[i];
            q[1] = float(1.0 / double(mat_param[i]));  // material.h:69 (1.0/refraction_index)
            // material.h:94-95: r0 = ((1 - ri) / (1 + ri))^2 for ri = ior (ray inside) and ri = 1/ior (ray outside), fp32
            for (int side = 0; side < 2; ++side) {
                const float ri = q[side];
                float r0 = (1.0f - ri) / (1.0f + ri);
                r0 = r0 * r0;
                q[2 + side] = r0;
            }
        } else {
            q[0] = mat_albedo[3 * i]; q[1] = mat_albedo[3 * i + 1]; q[2] = mat_albedo[3 * i + 2];
            q[3] = mat_type[i] == CRT_METAL ? (mat_param[i] < 1.0f ? mat_param[i] : 1.0f) : 0.0f;  // material.h:46
        }
    }
    if (ctx->d_spheres) { (void)hipFree(ctx->d_spheres); ctx->d_spheres = nullptr; }
    if (ctx->d_mats) { (void)hipFree(ctx->d_mats); ctx->d_mats = nullptr; }
    if (ctx->d_pairs) { (void)hipFree(ctx->d_pairs); ctx->d_pairs = nullptr; }
    if (ctx->d_nodes) { (void)hipFree(ctx->d_nodes); ctx->d_nodes = nullptr; }
    ctx->num_leaves = 0;
    ctx->n = ctx->m = 0;  // no scene until the new one is completely on the device
    CRT_HIP_TRY(ctx, hipMalloc(&ctx->d_spheres, sph.size() * sizeof(float)));
    CRT_HIP_TRY(ctx, hipMalloc(&ctx->d_mats, mats.size() * sizeof(float)));
    CRT_HIP_TRY(ctx, hipMemcpy(ctx->d_spheres, sph.data(), sph.size() * sizeof(float), hipMemcpyHostToDevice));
    CRT_HIP_TRY(ctx, hipMemcpy(ctx->d_mats, mats.data(), mats.size() * sizeof(float), hipMemcpyHostToDevice));
    ctx->n = n; ctx->m = m;
    ctx->center.assign(center, center + 3 * size_t(n));
    ctx->motion.assign(motion, motion + 3 * size_t(n));
    ctx->radius.assign(radius, radius + size_t(n));
    return CRT_OK;
}

int crt_build_bvh(crt_context* ctx, uint32_t axis_seed, int32_t* num_nodes_out) {
    if (!ctx) return CRT_ERR_INVALID_ARGUMENT;
    if (ctx->n < 1) return fail(ctx, CRT_ERR_STATE, "crt_build_bvh: no scene uploaded");
    CRT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    CRT_HIP_TRY(ctx, hipDeviceSynchronize());
    if (crt::bvh_depth(ctx->n) > 28) return fail(ctx, CRT_ERR_TOO_LARGE, "crt_build_bvh: tree deeper than 28 levels (the traversal keeps one bit per level in 32)");
    const int depth = crt::bvh_depth(ctx->n);
    const int L = 1 << depth;
    if (ctx->d_pairs) { (void)hipFree(ctx->d_pairs); ctx->d_pairs = nullptr; }
    if (ctx->d_nodes) { (void)hipFree(ctx->d_nodes); ctx->d_nodes = nullptr; }
    ctx->num_leaves = 0;
    if (!ctx->bvh_host) {
        // GPU builder (crt_bvh.hip): boxes and the axis table come from the host code both builders share
        const size_t n = size_t(ctx->n);
        std::vector<float> boxes(6 * n);
        std::vector<uint8_t> axis(2 * n - 1);
        crt::sphere_boxes(ctx->n, ctx->center.data(), ctx->motion.data(), ctx->radius.data(), boxes.data());
        crt::axis_table(ctx->n, axis_seed, axis.data());
        float* d_boxes = nullptr; uint8_t* d_axis = nullptr;
        hipError_t e = hipMalloc((void**)&d_boxes, boxes.size() * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void**)&d_axis, axis.size());
        if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_nodes, (2 * size_t(L) - 1) * sizeof(crt_bvh_node));
        if (e == hipSuccess) e = hipMalloc(&ctx->d_pairs, 16 * size_t(L) * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(d_boxes, boxes.data(), boxes.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_axis, axis.data(), axis.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = crt_gpu_bvh_build(ctx->n, depth, d_boxes, d_axis, ctx->d_nodes, (float*)ctx->d_pairs, nullptr);
        (void)hipFree(d_boxes); (void)hipFree(d_axis);
        if (e != hipSuccess) {
            if (ctx->d_pairs) { (void)hipFree(ctx->d_pairs); ctx->d_pairs = nullptr; }
            if (ctx->d_nodes) { (void)hipFree(ctx->d_nodes); ctx->d_nodes = nullptr; }
            return fail(ctx, CRT_ERR_HIP, std::string("crt_build_bvh (GPU builder): ") + hipGetErrorString(e));
        }
        ctx->num_leaves = L;
        if (num_nodes_out) *num_nodes_out = int32_t(2 * size_t(L) - 1);
        return CRT_OK;
    }
    // host builder (A/B of the GPU builder): the node array goes up as it is, the pair records are derived on the device
    std::vector<crt_bvh_node> nodes(2 * size_t(L) - 1);
    crt::bvh_build(ctx->n, ctx->center.data(), ctx->motion.data(), ctx->radius.data(), axis_seed, nodes.data());
    if (int rc = install_nodes(ctx, nodes.data(), L)) return rc;
    if (num_nodes_out) *num_nodes_out = int32_t(nodes.size());
    return CRT_OK;
}

int crt_upload_bvh_nodes(crt_context* ctx, const crt_bvh_node* nodes, int32_t count) {
    if (!ctx) return CRT_ERR_INVALID_ARGUMENT;
    if (!nodes) return fail(ctx, CRT_ERR_INVALID_ARGUMENT, "crt_upload_bvh_nodes: nodes is NULL");
    if (ctx->n < 1) return fail(ctx, CRT_ERR_STATE, "crt_upload_bvh_nodes: no scene uploaded");
    if (crt::bvh_depth(ctx->n) > 28) return fail(ctx, CRT_ERR_TOO_LARGE, "crt_upload_bvh_nodes: tree deeper than 28 levels");
    const int L = 1 << crt::bvh_depth(ctx->n);
    if (int64_t(count) != 2 * int64_t(L) - 1)
        return fail(ctx, CRT_ERR_INVALID_ARGUMENT, "crt_upload_bvh_nodes: count is not crt_bvh_num_nodes(n) of the uploaded scene");
    // the traversal trusts leaf tags as sphere indices: check them here, once
    for (int j = 0; j < L; ++j) {
        const crt_bvh_node& nd = nodes[size_t(L) - 1 + j];
        const bool empty = nd.xmin > nd.xmax;
        if (!empty && (nd.loc_for_leaf < 0 || nd.loc_for_leaf >= ctx->n))
            return fail(ctx, CRT_ERR_INVALID_ARGUMENT, "crt_upload_bvh_nodes: leaf with loc_for_leaf outside [0, n)");
    }
    CRT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    CRT_HIP_TRY(ctx, hipDeviceSynchronize());
    return install_nodes(ctx, nodes, L);
}

int crt_get_bvh_nodes(const crt_context* ctx, crt_bvh_node* out, int32_t capacity) {
    if (!ctx || !out) return CRT_ERR_INVALID_ARGUMENT;
    if (ctx->num_leaves < 1) return CRT_ERR_STATE;
    const size_t count = 2 * size_t(ctx->num_leaves) - 1;
    if (size_t(capacity) < count) return CRT_ERR_TOO_LARGE;
    if (!ctx->d_nodes) return CRT_ERR_STATE;
    if (hipSetDevice(ctx->device) != hipSuccess ||
        hipMemcpy(out, ctx->d_nodes, count * sizeof(crt_bvh_node), hipMemcpyDeviceToHost) != hipSuccess)
        return CRT_ERR_HIP;
    return CRT_OK;
}

int crt_set_camera(crt_context* ctx, const crt_camera* cam) {
    if (!ctx) return CRT_ERR_INVALID_ARGUMENT;
    if (!cam || cam->image_width < 1 || cam->image_height < 1)
        return fail(ctx, CRT_ERR_INVALID_ARGUMENT, "crt_set_camera: bad camera");
    if (int64_t(cam->image_width) * cam->image_height > (1ll << 31) - 1)
        return fail(ctx, CRT_ERR_TOO_LARGE, "crt_set_camera: more than 2^31-1 pixels");
    ctx->cam = *cam;
    ctx->have_cam = true;
    return CRT_OK;
}

// state and argument checks of every render entry point; nothing is enqueued before they pass
static int validate_render(crt_context* ctx, int world, int s0, int ns, int max_depth) {
    if (!ctx->have_cam) return fail(ctx, CRT_ERR_STATE, "render: no camera set");
    if (ctx->n < 1) return fail(ctx, CRT_ERR_STATE, "render: no scene uploaded");
    if (world != CRT_WORLD_LIST && world != CRT_WORLD_BVH) return fail(ctx, CRT_ERR_INVALID_ARGUMENT, "render: unknown world kind");
    if (world == CRT_WORLD_BVH && !ctx->d_pairs) return fail(ctx, CRT_ERR_STATE, "render: CRT_WORLD_BVH needs crt_build_bvh first");
    if (ns < 1 || s0 < 0 || max_depth < 1) return fail(ctx, CRT_ERR_INVALID_ARGUMENT, "render: nsamples, sample_offset or max_depth out of range");
    if (ns > (1 << 20)) return fail(ctx, CRT_ERR_TOO_LARGE, "render: more than 2^20 samples in one call (split the range)");
    return CRT_OK;
}

static int fill_args(crt_context* ctx, int world, int block, int s0, int ns, int max_depth, int tile_stride, int tile_offset,
                     unsigned long long* d_sums, unsigned long long* counters, crt_render_args& a) {
    if (int rc = validate_render(ctx, world, s0, ns, max_depth)) return rc;
    const crt_camera& c = ctx->cam;
    memcpy(a.cam.center, c.center, 12); memcpy(a.cam.p00, c.pixel00_loc, 12);
    memcpy(a.cam.du, c.pixel_delta_u, 12); memcpy(a.cam.dv, c.pixel_delta_v, 12);
    memcpy(a.cam.ddu, c.defocus_disk_u, 12); memcpy(a.cam.ddv, c.defocus_disk_v, 12);
    a.cam.defocus_angle = c.defocus_angle;
    a.cam.width = c.image_width; a.cam.height = c.image_height;
    a.pairs = ctx->d_pairs; a.spheres = ctx->d_spheres; a.mats = ctx->d_mats;
    a.num_leaves = ctx->num_leaves; a.n = ctx->n; a.m = ctx->m;
    a.key0 = uint32_t(ctx->seed); a.key1 = uint32_t(ctx->seed >> 32);
    a.s0 = s0; a.ns = ns;
    // tiles per workgroup: enough (pixel, sample) items that every thread gets >= 16 of them
    int t = ctx->tiles_per_wg;
    if (t == 0) { t = 1; while (t < 16 && int64_t(64) * t * ns < int64_t(16) * block) t <<= 1; }
    a.tiles_per_wg = t;
    a.tiles_x = (c.image_width + 7) / 8; a.tiles_y = (c.image_height + 7) / 8;
    if (tile_stride < 1 || tile_offset < 0 || tile_offset >= tile_stride)
        return fail(ctx, CRT_ERR_INVALID_ARGUMENT, "render: tile share needs tile_stride >= 1 and 0 <= tile_offset < tile_stride");
    const int64_t all_tiles = int64_t(a.tiles_x) * a.tiles_y;
    a.tile_stride = tile_stride; a.tile_offset = tile_offset;
    a.n_tiles = int(tile_offset < all_tiles ? (all_tiles - tile_offset + tile_stride - 1) / tile_stride : 0);
    a.max_depth = max_depth;
    a.sched = ctx->sched; a.census = ctx->census; a.run_shift = ctx->run_shift; a.leaf_num = ctx->leaf_num; a.shade_num = ctx->shade_num;
    a.sums = d_sums;
    a.counters = counters;
    return CRT_OK;
}

// One render launch: the samples [sample_offset, sample_offset + nsamples) of a tile share, added to d_sums on
// `stream`; `counters` is the launch's scratch (work-queue counter, statistics, census).
static int launch_accumulate(crt_context* ctx, int32_t world, int32_t sample_offset, int32_t nsamples, int32_t max_depth,
                             int32_t tile_stride, int32_t tile_offset, int64_t* d_sums, void* stream_,
                             unsigned long long* counters) {
    if (!ctx) return CRT_ERR_INVALID_ARGUMENT;
    const int ns_ = nsamples;
    if (!d_sums) return fail(ctx, CRT_ERR_INVALID_ARGUMENT, "render: d_sums is NULL");
    // LDS image of the scene (render_kernel): spheres, and for a BVH world the pair records (simple
    // loop, 64 B per pair) or the sign-sorted plane arrays + leaf tags (scheduled, 96 B + 4 B per pair)
    const bool scheduled = world == CRT_WORLD_BVH && ctx->sched;
    const size_t L = size_t(ctx->num_leaves);
    size_t scene_bytes = (scheduled && !CRT_SCHED_SPHERES_IN_LDS) ? 0 : 32 * size_t(ctx->n);
    if (world == CRT_WORLD_BVH) scene_bytes += scheduled ? 96 * L + 16 * ((L - L / 2 + 1) / 2) : 64 * L;
    const bool in_lds = scene_bytes <= ctx->lds_limit;
    // a scene that stays in HBM still gets the first pair records (the upper tree levels) in LDS
    int top_pairs = 0;
    if (scheduled && ctx->persist && !in_lds && ctx->top_pairs_max > 0) { top_pairs = 1; while (2 * top_pairs <= ctx->top_pairs_max && size_t(2 * top_pairs) <= L) top_pairs *= 2; }
    // workgroup size: 24 waves per CU (6 per SIMD: the kernel stays within 80 VGPRs, 77 in the round-2 build) from as many workgroups as
    // the 160 KB of LDS hold
    int block = ctx->block;
    if (block == 0) {
        if (scheduled && ctx->persist)  // + 2080 B of LDS per wave: two 12-wave workgroups per CU up to a 55 KB scene image
            block = !in_lds ? (top_pairs > 0 ? 768 : 512) : (scene_bytes <= 55 * 1024 ? 768 : 1024);
        else
            block = !in_lds || scene_bytes <= 50 * 1024 ? 512 : (scene_bytes <= 78 * 1024 ? 768 : 1024);
    }
    // SCHED=3 (walker + owner waves): LDS-resident trees under persistent waves only, 12- or 16-wave workgroups
    const bool split_waves = scheduled && ctx->sched == 3 && ctx->persist && in_lds;
    if (split_waves && block != 768 && block != 1024) block = scene_bytes <= 52 * 1024 ? 768 : 1024;
    crt_render_args a;
    memset(&a, 0, sizeof a);
    if (int rc = fill_args(ctx, world, block, sample_offset, nsamples, max_depth, tile_stride, tile_offset,
                           (unsigned long long*)d_sums, counters, a)) return rc;
    if (!split_waves && a.sched == 3) a.sched = 1;
    CRT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = (hipStream_t)stream_;
    const int nw = block / 64;
    const int n_walk = split_waves ? (ctx->walkers < nw - 1 ? ctx->walkers : nw - 1) : 0;
    const int n_own = nw - n_walk;  // waves that own paths (all of them without walkers)
    const int64_t tiles = a.n_tiles;  // of this launch's share
    size_t lds = in_lds ? scene_bytes : 48 * size_t(top_pairs);
    int grid;
    if (scheduled && ctx->persist) {
        // persistent waves: as many workgroups as the chip holds at once (LDS- or wave-limited), each
        // wave pulling (tile, sample chunk) units; coarse units sized for about 16 per wave, 8..32 samples each
        lds += size_t(n_own) * 260 * 8;  // per path-owning wave: one tile of accumulators, its unit state, 32 block requests
        if (split_waves) lds += size_t(n_own) * 16 * CRT_XCHG_FLOAT4_PER_OWNER;  // + its request ring, control words and mailboxes
        a.n_walkers = n_walk; a.refill_min = ctx->refill_min; a.owner_sleep = ctx->owner_sleep;
        a.walker_backoff = ctx->walker_backoff; a.walker_prio = ctx->walker_prio;
        a.walker_min_busy = ctx->walker_min_busy; a.walker_patience = ctx->walker_patience;
        int per_cu = int((160 * 1024) / (lds > 0 ? lds : 1));
        const int by_waves = 24 / nw;  // 6 waves per SIMD: the kernel stays within 80 VGPRs
        if (per_cu > by_waves) per_cu = by_waves;
        if (per_cu < 1) per_cu = 1;
        const int64_t resident_waves = int64_t(ctx->num_cus) * per_cu * n_own;
        int us = ctx->unit_samples;
        if (us == 0) { us = int(int64_t(ns_) * tiles / (16 * resident_waves)); us = us < 8 ? 8 : (us > 32 ? 32 : us); }
        a.persist = 1;
        a.top_pairs = top_pairs;
        a.tail_pass = ctx->tail_pass;
        // the last fine_pct % of the samples go out in smaller units, after all the coarse ones
        int fs = ctx->fine_samples;
        if (fs == 0) fs = us / 4 < 2 ? 2 : us / 4;
        if (ctx->fine_pct == 0 || fs >= us) fs = us;  // one unit size: the fine round only takes the remainder
        const int ns_fine = fs < us ? int(int64_t(ns_) * ctx->fine_pct / 100) : 0;
        a.unit_samples = us;
        a.unit_chunks = (ns_ - ns_fine) / us;
        a.ns_coarse = a.unit_chunks * us;
        a.n_coarse = int(tiles * a.unit_chunks);
        a.fine_samples = fs;
        a.fine_chunks = (ns_ - a.ns_coarse + fs - 1) / fs;
        const int64_t units = tiles * (int64_t(a.unit_chunks) + a.fine_chunks);
        if (units > (1ll << 31) - 1) return fail(ctx, CRT_ERR_TOO_LARGE, "render: more than 2^31-1 work units (split the sample range)");
        a.n_units = int(units);
        const int64_t wgs = (units + n_own - 1) / n_own;
        grid = int(wgs < int64_t(ctx->num_cus) * per_cu ? wgs : int64_t(ctx->num_cus) * per_cu);
    } else {
        // simple loop: one wave per (tile, sample chunk); enough chunks that a small frame still fills the chip
        if (!scheduled) {
            const int64_t want_waves = int64_t(ctx->num_cus) * 64;
            int64_t chunks = tiles > 0 ? (want_waves + tiles - 1) / tiles : 1;
            chunks = chunks < 1 ? 1 : (chunks > ns_ ? ns_ : chunks);
            if (const char* b = getenv("CRT_SIMPLE_CHUNKS")) { const int v = atoi(b); if (v >= 1) chunks = v < ns_ ? v : ns_; }
            a.simple_chunks = int(chunks);
        }
        const int64_t groups = scheduled ? (tiles + a.tiles_per_wg - 1) / a.tiles_per_wg : (tiles * a.simple_chunks + nw - 1) / nw;
        if (groups > (1ll << 31) - 1) return fail(ctx, CRT_ERR_TOO_LARGE, "render: more than 2^31-1 workgroups (split the sample range)");
        grid = int(groups);
        if (scheduled) lds += size_t(a.tiles_per_wg) * (192 * 8 + 8) + 16;  // accumulators, tile origins, item counter
    }
    CRT_HIP_TRY(ctx, hipMemsetAsync(counters, 0, CRT_TIMELINE_BASE * sizeof(unsigned long long), stream));  // census + striped statistics
    const int slot = int(ctx->launches % crt_context::EV_RING);
    const bool fast = ctx->arith == CRT_ARITH_FAST;
    // Under stream capture (crt_render_static exists for it) an event record would become a graph node and the
    // event would never be recorded on a real stream: the slot is marked untimed instead (kernel_seconds = 0).
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    const bool timed = cap == hipStreamCaptureStatusNone;
    if (timed) CRT_HIP_TRY(ctx, hipEventRecord(ctx->ev_ring[2 * slot], stream));
    if (tiles > 0)  // (a share beyond the last tile is empty: nothing to launch, the statistics read zero)
        CRT_HIP_TRY(ctx, (fast ? crt_launch_render_fast : crt_launch_render)(a, world, in_lds, ctx->count_stats, block, grid, lds, stream));
    if (timed) CRT_HIP_TRY(ctx, hipEventRecord(ctx->ev_ring[2 * slot + 1], stream));
    ctx->ev_timed[slot] = timed;
    ctx->launches += 1;
    ctx->launched = true;
    ctx->last_counters = counters;
    const int sched_id = world == CRT_WORLD_BVH ? (ctx->sched ? (ctx->persist ? (split_waves ? 3 : 2) : 1) : 0) : 0;
    snprintf(ctx->last_kernel, sizeof ctx->last_kernel, "render_kernel<WORLD=%s, IN_LDS=%d, COUNT=%d, BLOCK=%d, SCHED=%d>%s",
             world == CRT_WORLD_BVH ? "BVH" : "LIST", int(in_lds), int(ctx->count_stats), block, sched_id, fast ? " [arith=fast]" : "");
    return CRT_OK;
}

int crt_render_accumulate(crt_context* ctx, int32_t world, int32_t sample_offset, int32_t nsamples,
                          int32_t max_depth, int64_t* d_sums, void* stream) {
    if (!ctx) return CRT_ERR_INVALID_ARGUMENT;
    return launch_accumulate(ctx, world, sample_offset, nsamples, max_depth, 1, 0, d_sums, stream, ctx->d_counters);
}

int crt_render_accumulate_tiles(crt_context* ctx, int32_t world, int32_t sample_offset, int32_t nsamples,
                                int32_t max_depth, int32_t tile_stride, int32_t tile_offset, int64_t* d_sums,
                                void* stream) {
    if (!ctx) return CRT_ERR_INVALID_ARGUMENT;
    return launch_accumulate(ctx, world, sample_offset, nsamples, max_depth, tile_stride, tile_offset, d_sums, stream,
                             ctx->d_counters);
}

int crt_render_static(crt_context* ctx, int32_t world, int32_t spp, int32_t sample_offset, int32_t max_depth,
                      float* d_fb, void* stream_) {
    if (!ctx) return CRT_ERR_INVALID_ARGUMENT;
    if (!d_fb) return fail(ctx, CRT_ERR_INVALID_ARGUMENT, "crt_render_static: d_fb is NULL");
    if (!ctx->have_cam) return fail(ctx, CRT_ERR_STATE, "crt_render_static: no camera set");
    if (spp < 1) return fail(ctx, CRT_ERR_INVALID_ARGUMENT, "crt_render_static: spp < 1");
    const size_t pixels = size_t(ctx->cam.image_width) * size_t(ctx->cam.image_height);
    if (pixels > size_t(CRT_STATIC_MAX_PIXELS))
        return fail(ctx, CRT_ERR_TOO_LARGE, "crt_render_static: frame of " + std::to_string(pixels) + " pixels exceeds the static scratch (" +
                                                std::to_string(CRT_STATIC_MAX_PIXELS) + " pixels); use crt_render / crt_render_accumulate");
    static_assert(CRT_STATIC_MAX_PIXELS == CRT_STATIC_MAX_PIXELS_, "header and scratch size disagree");
    // every check first: a rejected call enqueues nothing (under capture: adds no node to the graph)
    if (int rc = validate_render(ctx, world, sample_offset, spp, max_depth)) return rc;
    CRT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    // The address of THIS GPU's copy of the array: resolved on the first call on a device (crt_static.hip caches it
    // per device).  Resolving it at crt_create would make every context pay for loading the array (17 ms measured for
    // 48 MiB) whether or not it ever renders this way; like the first launch of any kernel, the first call on a device
    // loads a code object -- a caller that must not allocate makes one warm-up call before it starts.
    if (!ctx->d_static) CRT_HIP_TRY(ctx, crt_static_scratch(ctx->device, &ctx->d_static));
    hipStream_t stream = (hipStream_t)stream_;
    unsigned long long* counters = ctx->d_static;                       // this GPU's array, shared by every call on it
    unsigned long long* sums = ctx->d_static + CRT_STATIC_COUNTER_WORDS;
    // the array keeps the last call's contents: the sums of THIS frame must start at zero (the launch clears the counters)
    CRT_HIP_TRY(ctx, hipMemsetAsync(sums, 0, 3 * pixels * sizeof(unsigned long long), stream));
    if (int rc = launch_accumulate(ctx, world, sample_offset, spp, max_depth, 1, 0, (int64_t*)sums, stream_, counters)) return rc;
    CRT_HIP_TRY(ctx, crt_launch_resolve(sums, d_fb, 3 * pixels, spp, stream));  // camera.h:76 scale
    return CRT_OK;
}

int crt_resolve(crt_context* ctx, const int64_t* d_sums, int32_t spp, float* d_fb, void* stream) {
    if (!ctx) return CRT_ERR_INVALID_ARGUMENT;
    if (!d_sums || !d_fb || spp < 1 || !ctx->have_cam) return fail(ctx, CRT_ERR_INVALID_ARGUMENT, "crt_resolve: bad arguments or no camera");
    CRT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t count = 3 * size_t(ctx->cam.image_width) * size_t(ctx->cam.image_height);
    CRT_HIP_TRY(ctx, crt_launch_resolve(d_sums, d_fb, count, spp, (hipStream_t)stream));
    return CRT_OK;
}

int crt_get_stats(crt_context* ctx, crt_render_stats* stats) {
    if (!ctx || !stats) return CRT_ERR_INVALID_ARGUMENT;
    memset(stats, 0, sizeof *stats);
    if (!ctx->launched) return fail(ctx, CRT_ERR_STATE, "crt_get_stats: nothing launched yet");
    CRT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int slot = int((ctx->launches - 1) % crt_context::EV_RING);
    float ms = 0;
    if (ctx->ev_timed[slot]) {
        CRT_HIP_TRY(ctx, hipEventSynchronize(ctx->ev_ring[2 * slot + 1]));
        CRT_HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_ring[2 * slot], ctx->ev_ring[2 * slot + 1]));
    } else {
        CRT_HIP_TRY(ctx, hipDeviceSynchronize());  // a captured launch: no events; the counters are read after the device is idle
    }
    unsigned long long stripes[4 * CRT_STAT_STRIPES], c[4] = {0, 0, 0, 0}, watchdog = 0;
    CRT_HIP_TRY(ctx, hipMemcpy(stripes, ctx->last_counters + CRT_STATS_BASE, sizeof stripes, hipMemcpyDeviceToHost));
    CRT_HIP_TRY(ctx, hipMemcpy(&watchdog, ctx->last_counters + 29, sizeof watchdog, hipMemcpyDeviceToHost));
    if (watchdog != 0) return fail(ctx, CRT_ERR_HIP, "render: the walker/owner exchange watchdog fired (" + std::to_string(watchdog) + " waves gave up waiting): the frame is incomplete");
    for (int i = 0; i < CRT_STAT_STRIPES; ++i)
        for (int j = 0; j < 4; ++j) c[j] += stripes[4 * i + j];
    stats->kernel_seconds = ms * 1e-3;
    stats->wall_seconds = ctx->wall_seconds;
    stats->rays = c[0]; stats->samples = c[1]; stats->pair_visits = c[2]; stats->sphere_tests = c[3];
    return CRT_OK;
}

int crt_get_kernel_times(crt_context* ctx, int32_t count, double* seconds_out) {
    if (!ctx || !seconds_out || count < 1) return CRT_ERR_INVALID_ARGUMENT;
    if (count > crt_context::EV_RING || count > ctx->launches)
        return fail(ctx, CRT_ERR_INVALID_ARGUMENT, "crt_get_kernel_times: more launches asked for than recorded (ring of 64)");
    CRT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (int i = 0; i < count; ++i) {  // oldest of the `count` most recent launches first
        const int slot = int((ctx->launches - count + i) % crt_context::EV_RING);
        float ms = 0;
        if (ctx->ev_timed[slot]) {  // (a launch captured into a graph is untimed: 0)
            CRT_HIP_TRY(ctx, hipEventSynchronize(ctx->ev_ring[2 * slot + 1]));
            CRT_HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_ring[2 * slot], ctx->ev_ring[2 * slot + 1]));
        }
        seconds_out[i] = ms * 1e-3;
    }
    return CRT_OK;
}

// not part of the public header: scheduler census of the last counting launch (tools/tune.py)
int crt_debug_counters(crt_context* ctx, uint64_t* out16) {
    if (!ctx || !out16) return CRT_ERR_INVALID_ARGUMENT;
    CRT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    CRT_HIP_TRY(ctx, hipDeviceSynchronize());
    CRT_HIP_TRY(ctx, hipMemcpy(out16, ctx->last_counters ? ctx->last_counters : ctx->d_counters, 32 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return CRT_OK;
}

// not part of the public header: (start, queue-empty, exit) wall-clock ticks (100 MHz) of the first 8192 waves of the
// last phase-cycle census launch (crt_set_count_stats(ctx, 3)); tools/tune.py
int crt_debug_timeline(crt_context* ctx, uint64_t* out, int32_t waves) {
    if (!ctx || !out || waves < 1 || waves > 8192) return CRT_ERR_INVALID_ARGUMENT;
    CRT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    CRT_HIP_TRY(ctx, hipDeviceSynchronize());
    CRT_HIP_TRY(ctx, hipMemcpy(out, (ctx->last_counters ? ctx->last_counters : ctx->d_counters) + CRT_TIMELINE_BASE, size_t(waves) * 3 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return CRT_OK;
}

int crt_render(crt_context* ctx, int32_t world, int32_t spp, int32_t sample_offset, int32_t max_depth, float* fb_host,
               crt_render_stats* stats) {
    if (!ctx) return CRT_ERR_INVALID_ARGUMENT;
    if (!fb_host) return fail(ctx, CRT_ERR_INVALID_ARGUMENT, "crt_render: fb_host is NULL");
    if (!ctx->have_cam) return fail(ctx, CRT_ERR_STATE, "crt_render: no camera set");
    const auto t0 = std::chrono::steady_clock::now();
    CRT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t floats = 3 * size_t(ctx->cam.image_width) * size_t(ctx->cam.image_height);
    if (ctx->fb_floats < floats) {
        if (ctx->d_fb) { (void)hipFree(ctx->d_fb); ctx->d_fb = nullptr; }
        if (ctx->d_sums) { (void)hipFree(ctx->d_sums); ctx->d_sums = nullptr; }
        ctx->fb_floats = 0;
        CRT_HIP_TRY(ctx, hipMalloc((void**)&ctx->d_fb, floats * sizeof(float)));
        CRT_HIP_TRY(ctx, hipMalloc((void**)&ctx->d_sums, floats * sizeof(unsigned long long)));
        ctx->fb_floats = floats;
    }
    CRT_HIP_TRY(ctx, hipMemsetAsync(ctx->d_sums, 0, floats * sizeof(unsigned long long), nullptr));
    if (int rc = crt_render_accumulate(ctx, world, sample_offset, spp, max_depth, (int64_t*)ctx->d_sums, nullptr)) return rc;
    if (int rc = crt_resolve(ctx, (const int64_t*)ctx->d_sums, spp, ctx->d_fb, nullptr)) return rc;  // camera.h:76 scale
    CRT_HIP_TRY(ctx, hipMemcpy(fb_host, ctx->d_fb, floats * sizeof(float), hipMemcpyDeviceToHost));
    ctx->wall_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (stats) return crt_get_stats(ctx, stats);
    return CRT_OK;
}

int crt_trace_samples(crt_context* ctx, int32_t world, int32_t count, const int32_t* px, const int32_t* py,
                      const int32_t* sample, int32_t max_depth, float* ray7_out, float* rgb_out, int32_t* segments_out) {
    if (!ctx) return CRT_ERR_INVALID_ARGUMENT;
    if (count < 1 || !px || !py || !sample || !ray7_out || !rgb_out || !segments_out)
        return fail(ctx, CRT_ERR_INVALID_ARGUMENT, "crt_trace_samples: bad arguments");
    crt_render_args a;
    memset(&a, 0, sizeof a);
    if (int rc = fill_args(ctx, world, 512, 0, 1, max_depth, 1, 0, nullptr, ctx->d_counters, a)) return rc;
    for (int i = 0; i < count; ++i)
        if (px[i] < 0 || px[i] >= ctx->cam.image_width || py[i] < 0 || py[i] >= ctx->cam.image_height || sample[i] < 0)
            return fail(ctx, CRT_ERR_INVALID_ARGUMENT, "crt_trace_samples: pixel or sample out of range");
    CRT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->dbg_cap < count) {
        void** bufs[] = {(void**)&ctx->d_px, (void**)&ctx->d_py, (void**)&ctx->d_smp, (void**)&ctx->d_seg, (void**)&ctx->d_ray7, (void**)&ctx->d_rgb};
        for (void** b : bufs) if (*b) { (void)hipFree(*b); *b = nullptr; }
        ctx->dbg_cap = 0;
        CRT_HIP_TRY(ctx, hipMalloc((void**)&ctx->d_px, size_t(count) * 4));
        CRT_HIP_TRY(ctx, hipMalloc((void**)&ctx->d_py, size_t(count) * 4));
        CRT_HIP_TRY(ctx, hipMalloc((void**)&ctx->d_smp, size_t(count) * 4));
        CRT_HIP_TRY(ctx, hipMalloc((void**)&ctx->d_seg, size_t(count) * 4));
        CRT_HIP_TRY(ctx, hipMalloc((void**)&ctx->d_ray7, size_t(count) * 28));
        CRT_HIP_TRY(ctx, hipMalloc((void**)&ctx->d_rgb, size_t(count) * 12));
        ctx->dbg_cap = count;
    }
    CRT_HIP_TRY(ctx, hipMemcpy(ctx->d_px, px, size_t(count) * 4, hipMemcpyHostToDevice));
    CRT_HIP_TRY(ctx, hipMemcpy(ctx->d_py, py, size_t(count) * 4, hipMemcpyHostToDevice));
    CRT_HIP_TRY(ctx, hipMemcpy(ctx->d_smp, sample, size_t(count) * 4, hipMemcpyHostToDevice));
    CRT_HIP_TRY(ctx, (ctx->arith == CRT_ARITH_FAST ? crt_launch_trace_samples_fast : crt_launch_trace_samples)(a, world, count, ctx->d_px, ctx->d_py, ctx->d_smp, ctx->d_ray7, ctx->d_rgb, ctx->d_seg, nullptr));
    CRT_HIP_TRY(ctx, hipDeviceSynchronize());
    CRT_HIP_TRY(ctx, hipMemcpy(ray7_out, ctx->d_ray7, size_t(count) * 28, hipMemcpyDeviceToHost));
    CRT_HIP_TRY(ctx, hipMemcpy(rgb_out, ctx->d_rgb, size_t(count) * 12, hipMemcpyDeviceToHost));
    CRT_HIP_TRY(ctx, hipMemcpy(segments_out, ctx->d_seg, size_t(count) * 4, hipMemcpyDeviceToHost));
    return CRT_OK;
}


// ---- several GPUs of one node behind one handle (include/crt_hip.h "crt_group") ---------------------------

}  // extern "C"

struct crt_group {
    std::vector<crt_context*> members;
    std::vector<hipStream_t> streams;            // one per member, on its device
    std::vector<hipEvent_t> done;                // member's sums (or their copy on members[0]'s device) are complete
    std::vector<unsigned long long*> sums;       // member's own int64 frame, on its device
    std::vector<unsigned long long*> staging;    // [i-1]: copy of member i's frame on members[0]'s device
    float* d_fb = nullptr;                       // on members[0]'s device
    size_t floats = 0;
    std::string err;
};

namespace {

std::string g_group_create_error;

int gfail(crt_group* g, int code, const std::string& msg) {
    if (g) g->err = msg; else g_group_create_error = msg;
    return code;
}

// fn(member index) on one host thread per member (the set-up calls are synchronous per context); first failure wins
template <class F>
int for_each_member(crt_group* g, const char* what, F fn) {
    const size_t n = g->members.size();
    std::vector<int> rc(n, CRT_OK);
    if (n == 1) rc[0] = fn(0);
    else {
        std::vector<std::thread> th;
        for (size_t i = 0; i < n; ++i) th.emplace_back([&, i] { rc[i] = fn(int(i)); });
        for (auto& t : th) t.join();
    }
    for (size_t i = 0; i < n; ++i)
        if (rc[i] != CRT_OK)
            return gfail(g, rc[i], std::string(what) + ": member " + std::to_string(i) + " (device " +
                                       std::to_string(g->members[i]->device) + "): " + g->members[i]->err);
    return CRT_OK;
}

void group_free_frames(crt_group* g) {
    for (size_t i = 0; i < g->members.size(); ++i) {
        (void)hipSetDevice(g->members[i]->device);
        if (i < g->sums.size() && g->sums[i]) { (void)hipFree(g->sums[i]); g->sums[i] = nullptr; }
    }
    if (!g->members.empty()) (void)hipSetDevice(g->members[0]->device);
    for (auto& p : g->staging) if (p) { (void)hipFree(p); p = nullptr; }
    if (g->d_fb) { (void)hipFree(g->d_fb); g->d_fb = nullptr; }
    g->floats = 0;
}

#define CRT_GROUP_TRY(g, expr)                                                                       \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return gfail(g, CRT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));         \
    } while (0)

}  // namespace

extern "C" {

const char* crt_group_last_error(const crt_group* g) { return g ? g->err.c_str() : g_group_create_error.c_str(); }

int crt_group_create(const int32_t* devices, int32_t n, uint64_t seed, crt_group** out) {
    if (!out) return gfail(nullptr, CRT_ERR_INVALID_ARGUMENT, "crt_group_create: out is NULL");
    *out = nullptr;
    if (!devices || n < 1 || n > CRT_GROUP_MAX)
        return gfail(nullptr, CRT_ERR_INVALID_ARGUMENT, "crt_group_create: device list empty or longer than 16");
    crt_group* g = new crt_group;
    for (int i = 0; i < n; ++i) {
        crt_context* c = nullptr;
        const int rc = crt_create(devices[i], seed, &c);
        if (rc != CRT_OK) {
            const std::string msg = std::string("crt_group_create: device ") + std::to_string(devices[i]) + ": " + crt_last_error(nullptr);
            crt_group_destroy(g);
            return gfail(nullptr, rc, msg);
        }
        g->members.push_back(c);
        hipStream_t st = nullptr; hipEvent_t ev = nullptr;
        if (hipSetDevice(devices[i]) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
            if (st) (void)hipStreamDestroy(st);
            crt_group_destroy(g);
            return gfail(nullptr, CRT_ERR_HIP, "crt_group_create: stream / event creation failed");
        }
        g->streams.push_back(st); g->done.push_back(ev);
    }
    // peer access between members[0]'s device and the others, BOTH ways: crt_group_render issues member i's frame copy
    // on member i's stream (device i writes into the staging buffer on devices[0]), and devices[0] may read the others;
    // with it the copies are direct xGMI transfers.  They work without it (staged by the runtime), so a refusal is
    // not an error.
    for (int i = 1; i < n; ++i) {
        if (devices[i] == devices[0]) continue;
        const int pair[2][2] = {{devices[0], devices[i]}, {devices[i], devices[0]}};  // {accessing device, peer}
        for (const auto& pr : pair) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, pr[0], pr[1]) != hipSuccess || !can) { (void)hipGetLastError(); continue; }
            if (hipSetDevice(pr[0]) != hipSuccess || hipDeviceEnablePeerAccess(pr[1], 0) != hipSuccess)
                (void)hipGetLastError();  // already enabled, or not possible: clear the sticky error
        }
    }
    (void)hipSetDevice(devices[0]);
    g->sums.assign(size_t(n), nullptr);
    g->staging.assign(size_t(n > 1 ? n - 1 : 0), nullptr);
    *out = g;
    return CRT_OK;
}

int crt_group_destroy(crt_group* g) {
    if (!g) return CRT_OK;
    for (size_t i = 0; i < g->members.size(); ++i) {
        (void)hipSetDevice(g->members[i]->device);
        if (i < g->streams.size() && g->streams[i]) { (void)hipStreamSynchronize(g->streams[i]); (void)hipStreamDestroy(g->streams[i]); }
        if (i < g->done.size() && g->done[i]) (void)hipEventDestroy(g->done[i]);
    }
    group_free_frames(g);
    for (crt_context* c : g->members) crt_destroy(c);
    delete g;
    return CRT_OK;
}

int32_t crt_group_size(const crt_group* g) { return g ? int32_t(g->members.size()) : 0; }

crt_context* crt_group_member(crt_group* g, int32_t i) {
    return (g && i >= 0 && size_t(i) < g->members.size()) ? g->members[size_t(i)] : nullptr;
}

int crt_group_upload_scene(crt_group* g, int32_t n, const float* center, const float* motion, const float* radius,
                           const int32_t* mat_id, int32_t m, const int32_t* mat_type, const float* mat_albedo,
                           const float* mat_param) {
    if (!g) return CRT_ERR_INVALID_ARGUMENT;
    return for_each_member(g, "crt_group_upload_scene", [&](int i) {
        return crt_upload_scene(g->members[size_t(i)], n, center, motion, radius, mat_id, m, mat_type, mat_albedo, mat_param);
    });
}

int crt_group_build_bvh(crt_group* g, uint32_t axis_seed, int32_t* num_nodes_out) {
    if (!g) return CRT_ERR_INVALID_ARGUMENT;
    std::vector<int32_t> nn(g->members.size(), 0);
    const int rc = for_each_member(g, "crt_group_build_bvh", [&](int i) { return crt_build_bvh(g->members[size_t(i)], axis_seed, &nn[size_t(i)]); });
    if (rc == CRT_OK && num_nodes_out) *num_nodes_out = nn[0];
    return rc;
}

int crt_group_upload_bvh_nodes(crt_group* g, const crt_bvh_node* nodes, int32_t count) {
    if (!g) return CRT_ERR_INVALID_ARGUMENT;
    return for_each_member(g, "crt_group_upload_bvh_nodes", [&](int i) { return crt_upload_bvh_nodes(g->members[size_t(i)], nodes, count); });
}

int crt_group_set_camera(crt_group* g, const crt_camera* cam) {
    if (!g) return CRT_ERR_INVALID_ARGUMENT;
    for (size_t i = 0; i < g->members.size(); ++i)
        if (int rc = crt_set_camera(g->members[i], cam))
            return gfail(g, rc, std::string("crt_group_set_camera: ") + g->members[i]->err);
    return CRT_OK;
}

int crt_group_render(crt_group* g, int32_t world, int32_t spp, int32_t sample_offset, int32_t max_depth, int32_t split,
                     float* fb_host, crt_render_stats* stats) {
    if (!g) return CRT_ERR_INVALID_ARGUMENT;
    if (!fb_host) return gfail(g, CRT_ERR_INVALID_ARGUMENT, "crt_group_render: fb_host is NULL");
    if (split != CRT_SPLIT_SAMPLES && split != CRT_SPLIT_TILES) return gfail(g, CRT_ERR_INVALID_ARGUMENT, "crt_group_render: unknown split");
    if (spp < 1) return gfail(g, CRT_ERR_INVALID_ARGUMENT, "crt_group_render: spp < 1");
    crt_context* c0 = g->members[0];
    if (!c0->have_cam) return gfail(g, CRT_ERR_STATE, "crt_group_render: no camera set");
    const auto t0 = std::chrono::steady_clock::now();
    const int G = int(g->members.size());
    const size_t floats = 3 * size_t(c0->cam.image_width) * size_t(c0->cam.image_height);
    const size_t bytes = floats * sizeof(unsigned long long);
    if (g->floats != floats) {  // frame buffers: every member's own sums; on members[0]'s device the copies and the fp32 frame
        group_free_frames(g);
        for (int i = 0; i < G; ++i) {
            CRT_GROUP_TRY(g, hipSetDevice(g->members[size_t(i)]->device));
            CRT_GROUP_TRY(g, hipMalloc((void**)&g->sums[size_t(i)], bytes));
        }
        CRT_GROUP_TRY(g, hipSetDevice(c0->device));
        for (int i = 1; i < G; ++i) CRT_GROUP_TRY(g, hipMalloc((void**)&g->staging[size_t(i) - 1], bytes));
        CRT_GROUP_TRY(g, hipMalloc((void**)&g->d_fb, floats * sizeof(float)));
        g->floats = floats;
    }
    // every member's share, asynchronously on its own stream: zero, trace, copy to members[0]'s device
    std::vector<bool> has_work(size_t(G), false);
    for (int i = 0; i < G; ++i) {
        crt_context* c = g->members[size_t(i)];
        CRT_GROUP_TRY(g, hipSetDevice(c->device));
        hipStream_t st = g->streams[size_t(i)];
        CRT_GROUP_TRY(g, hipMemsetAsync(g->sums[size_t(i)], 0, bytes, st));
        int rc = CRT_OK;
        if (split == CRT_SPLIT_SAMPLES) {
            const int lo = int(int64_t(i) * spp / G), hi = int(int64_t(i + 1) * spp / G);
            if (hi > lo) {
                rc = crt_render_accumulate(c, world, sample_offset + lo, hi - lo, max_depth, (int64_t*)g->sums[size_t(i)], st);
                has_work[size_t(i)] = true;
            }
        } else {
            rc = crt_render_accumulate_tiles(c, world, sample_offset, spp, max_depth, G, i, (int64_t*)g->sums[size_t(i)], st);
            has_work[size_t(i)] = true;
        }
        if (rc != CRT_OK) return gfail(g, rc, std::string("crt_group_render: member ") + std::to_string(i) + ": " + c->err);
        if (i > 0)
            CRT_GROUP_TRY(g, hipMemcpyPeerAsync(g->staging[size_t(i) - 1], c0->device, g->sums[size_t(i)], c->device, bytes, st));
        CRT_GROUP_TRY(g, hipEventRecord(g->done[size_t(i)], st));
    }
    // members[0]: wait for the copies, add them to its own sums, resolve, read back
    CRT_GROUP_TRY(g, hipSetDevice(c0->device));
    hipStream_t s0 = g->streams[0];
    crt_frame_list others; memset(&others, 0, sizeof others);
    for (int i = 1; i < G; ++i) {
        CRT_GROUP_TRY(g, hipStreamWaitEvent(s0, g->done[size_t(i)], 0));
        others.p[others.n++] = g->staging[size_t(i) - 1];
    }
    if (others.n > 0) CRT_GROUP_TRY(g, crt_launch_sum_frames(g->sums[0], others, floats, s0));
    CRT_GROUP_TRY(g, crt_launch_resolve(g->sums[0], g->d_fb, floats, spp, s0));  // camera.h:76 scale
    CRT_GROUP_TRY(g, hipMemcpyAsync(fb_host, g->d_fb, floats * sizeof(float), hipMemcpyDeviceToHost, s0));
    CRT_GROUP_TRY(g, hipStreamSynchronize(s0));
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (stats) {
        memset(stats, 0, sizeof *stats);
        for (int i = 0; i < G; ++i) {
            if (!has_work[size_t(i)]) continue;
            crt_render_stats st;
            if (int rc = crt_get_stats(g->members[size_t(i)], &st))
                return gfail(g, rc, std::string("crt_group_render: statistics of member ") + std::to_string(i) + ": " + g->members[size_t(i)]->err);
            stats->rays += st.rays; stats->samples += st.samples;
            stats->pair_visits += st.pair_visits; stats->sphere_tests += st.sphere_tests;
            if (st.kernel_seconds > stats->kernel_seconds) stats->kernel_seconds = st.kernel_seconds;
        }
        stats->wall_seconds = wall;
    }
    return CRT_OK;
}

}  // extern "C"
