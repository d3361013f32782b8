// crt_host.cpp -- host-side pieces of the path that need no GPU: camera derivation, the two
// scene definitions, the BVH builder (ordering + complete tree), the PPM writer.
// Part of libcrt_hip.so (product).  Nothing here includes or links oracle/.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <numeric>
#include <vector>

#include "../../include/crt_hip.h"
#include "crt_host.h"

#pragma STDC FP_CONTRACT OFF

namespace crt {

// ---- Philox4x32-10 on the host (scene stream: counter = (block, 0, 0, 1)) -------------------
static void philox_block(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
    for (int round = 0; round < 10; ++round) {
        const uint64_t a = 0xD2511F53ull * c0, b = 0xCD9E8D57ull * c2;
        const uint32_t t0 = uint32_t(b >> 32) ^ c1 ^ k0, t2 = uint32_t(a >> 32) ^ c3 ^ k1;
        c1 = uint32_t(b); c3 = uint32_t(a); c0 = t0; c2 = t2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

class scene_rng {
  public:
    explicit scene_rng(uint64_t seed) : k0_(uint32_t(seed)), k1_(uint32_t(seed >> 32)) {}
    // [0,1): 1 - u with u in (0,1]  (rtweekend.h:20-23)
    float unit() {
        if (have_ == 0) { philox_block(next_block_++, 0u, 0u, 1u, k0_, k1_, words_); have_ = 4; }
        const uint32_t w = words_[4 - have_];
        --have_;
        const float u = 2.3283064e-10f + float(w) * 2.3283064e-10f;
        return 1.0f - u;
    }
    float range(float lo, float hi) { return lo + (hi - lo) * unit(); }  // rtweekend.h:25-28

  private:
    uint32_t k0_, k1_, next_block_ = 0, words_[4] = {0, 0, 0, 0};
    int have_ = 0;
};

struct f3 { float x, y, z; };
static inline f3 sub(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline f3 add(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline f3 scl(float t, f3 a) { return {t * a.x, t * a.y, t * a.z}; }
static inline f3 crs(f3 a, f3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static inline float nrm2(f3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
static inline f3 normalized(f3 a) { return scl(1.0f / sqrtf(nrm2(a)), a); }
static inline f3 rd(const float* p) { return {p[0], p[1], p[2]}; }
static inline void wr(float* p, f3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }

static inline float to_radians(float deg) {
    const float pi_f = 3.1415926535897932385f;
    return float(double(deg * pi_f) / 180.0);  // rtweekend.h:16-18
}

// camera::initialize, camera.h:72-107
void camera_initialize(const crt_camera_params& in, crt_camera& out) {
    const int W = in.image_width;
    int H = int(float(W) / in.aspect_ratio);
    if (H < 1) H = 1;
    out.image_width = W;
    out.image_height = H;
    out.pixel_samples_scale = float(1.0 / double(in.samples_per_pixel));
    const f3 from = rd(in.lookfrom), at = rd(in.lookat), up = rd(in.vup);
    const float focus = in.focus_dist;
    const float half_h = tanf(to_radians(in.vfov) / 2.0f);
    const float vp_h = 2.0f * half_h * focus;
    const double vp_w = double(vp_h) * (double(W) / double(H));
    const f3 w = normalized(sub(from, at));
    const f3 u = normalized(crs(up, w));
    const f3 v = crs(w, u);
    const f3 edge_u = scl(float(vp_w), u);
    const f3 edge_v = scl(vp_h, f3{-v.x, -v.y, -v.z});
    const f3 du = scl(1.0f / float(W), edge_u);
    const f3 dv = scl(1.0f / float(H), edge_v);
    const f3 corner = sub(sub(sub(from, scl(focus, w)), scl(1.0f / 2.0f, edge_u)), scl(1.0f / 2.0f, edge_v));
    const f3 p00 = add(corner, scl(0.5f, add(du, dv)));
    const float lens_r = focus * tanf(to_radians(in.defocus_angle / 2.0f));
    wr(out.center, from); wr(out.pixel00_loc, p00);
    wr(out.pixel_delta_u, du); wr(out.pixel_delta_v, dv);
    wr(out.defocus_disk_u, scl(lens_r, u)); wr(out.defocus_disk_v, scl(lens_r, v));
    out.defocus_angle = in.defocus_angle;
}

// ---- scenes -----------------------------------------------------------------------------
struct scene_writer {
    float *center, *motion, *radius, *albedo, *param;
    int32_t *mat_id, *mat_type;
    int count = 0;
    void push(f3 c, f3 mo, float r, int type, f3 alb, float prm) {
        wr(center + 3 * count, c); wr(motion + 3 * count, mo); radius[count] = r;
        mat_id[count] = count; mat_type[count] = type; wr(albedo + 3 * count, alb); param[count] = prm;
        ++count;
    }
};

// helper.h:14-31
int scene_small(float* center, float* motion, float* radius, int32_t* mat_id, int32_t* mat_type, float* albedo,
                float* param) {
    scene_writer s{center, motion, radius, albedo, param, mat_id, mat_type};
    const f3 still{0, 0, 0}, white{1, 1, 1};
    s.push({0.0f, -100.5f, -1.0f}, still, 100.0f, CRT_LAMBERTIAN, {0.8f, 0.2f, 0.2f}, 0.0f);  // ground
    s.push({0.0f, 0.0f, -1.2f}, still, 0.5f, CRT_LAMBERTIAN, {0.1f, 0.2f, 0.5f}, 0.0f);       // center
    s.push({-1.0f, 0.0f, -1.0f}, still, 0.5f, CRT_DIELECTRIC, white, 1.50f);                  // left
    s.push({-1.0f, 0.0f, -1.0f}, still, 0.4f, CRT_DIELECTRIC, white, float(1.00 / 1.50));     // bubble
    s.push({1.0f, 0.0f, -1.0f}, still, 0.5f, CRT_METAL, {0.7f, 0.7f, 0.7f}, 0.2f);            // right
    return s.count;
}

// helper.h:37-102; draw order per cell: choose_mat, x jitter, z jitter, then the material's draws
int scene_random(uint64_t seed, int half, float ground_radius, float* center, float* motion, float* radius,
                 int32_t* mat_id, int32_t* mat_type, float* albedo, float* param) {
    scene_writer s{center, motion, radius, albedo, param, mat_id, mat_type};
    scene_rng rng(seed);
    const f3 still{0, 0, 0}, white{1, 1, 1};
    s.push({0.0f, -ground_radius, 0.0f}, still, ground_radius, CRT_LAMBERTIAN, {0.5f, 0.5f, 0.5f}, 0.0f);
    s.push({0.0f, 1.0f, 0.0f}, still, 1.0f, CRT_DIELECTRIC, white, 1.5f);
    s.push({-4.0f, 1.0f, 0.0f}, still, 1.0f, CRT_LAMBERTIAN, {0.4f, 0.2f, 0.1f}, 0.0f);
    s.push({4.0f, 1.0f, 0.0f}, still, 1.0f, CRT_METAL, {0.7f, 0.6f, 0.5f}, 0.0f);
    for (int a = -half; a < half; ++a) {
        for (int b = -half; b < half; ++b) {
            const float pick = rng.unit();
            const float cx = float(double(a) + 0.9 * double(rng.unit()));
            const float cz = float(double(b) + 0.9 * double(rng.unit()));
            const f3 c{cx, 0.2f, cz};
            if (double(sqrtf(nrm2(sub(c, f3{4.0f, 0.2f, 0.0f})))) > 0.9) {
                if (double(pick) < 0.8) {  // diffuse, moving
                    float p[6];
                    for (float& q : p) q = rng.unit();
                    const f3 alb{p[0] * p[3], p[1] * p[4], p[2] * p[5]};
                    const f3 c2 = add(c, f3{0.0f, rng.range(0.0f, 0.5f), 0.0f});
                    s.push(c, sub(c2, c), 0.2f, CRT_LAMBERTIAN, alb, 0.0f);
                } else if (double(pick) < 0.95) {  // metal
                    f3 alb;
                    alb.x = rng.range(0.5f, 1.0f); alb.y = rng.range(0.5f, 1.0f); alb.z = rng.range(0.5f, 1.0f);
                    const float fuzz = rng.range(0.0f, 0.5f);
                    s.push(c, still, 0.2f, CRT_METAL, alb, fuzz < 1.0f ? fuzz : 1.0f);
                } else {  // glass
                    s.push(c, still, 0.2f, CRT_DIELECTRIC, white, 1.5f);
                }
            }
        }
    }
    return s.count;
}

// ---- BVH --------------------------------------------------------------------------------
// sphere.h:10-25: static box c -+ r; moving box = hull of the boxes at time 0 and time 1
static void sphere_box(const float* c, const float* mo, float r, float box[6]) {
    for (int ax = 0; ax < 3; ++ax) {
        const float p0 = c[ax] + 0.0f * mo[ax], p1 = c[ax] + 1.0f * mo[ax];
        float lo0 = p0 - r, hi0 = p0 + r, lo1 = p1 - r, hi1 = p1 + r;
        if (!(lo0 <= hi0)) std::swap(lo0, hi0);  // aabb(point, point) orders extrema, aabb.h:13-20
        if (!(lo1 <= hi1)) std::swap(lo1, hi1);
        box[2 * ax] = lo0 <= lo1 ? lo0 : lo1;      // interval(a, b), interval.h:12-16
        box[2 * ax + 1] = hi0 >= hi1 ? hi0 : hi1;
    }
}

int bvh_depth(int n) {
    int d = 0;
    while ((1ll << d) < n) ++d;  // == ceil(log2(n)) for n >= 1 (bvh.h:139)
    return d;
}

// The reference draws one axis per recursive call, pre-order (bvh.h:257).  A call on a span
// of s primitives makes 2s-1 calls in total, so the call on [start, start+span) with draw
// index d has children at d+1 and d+2*floor(span/2): the recursion is replayed with an
// explicit work list and plain index sorting.
void sphere_boxes(int n, const float* center, const float* motion, const float* radius, float* boxes) {
    for (int i = 0; i < n; ++i) sphere_box(center + 3 * size_t(i), motion + 3 * size_t(i), radius[i], boxes + 6 * size_t(i));
}

// minstd_rand + uniform_int_distribution<int>(0,2), as thrust maps it
void axis_table(int n, uint32_t axis_seed, uint8_t* axis) {
    uint64_t x = axis_seed % 2147483647u;
    if (x == 0) x = 1;
    for (size_t i = 0; i < 2 * size_t(n) - 1; ++i) {
        x = (x * 48271u) % 2147483647u;
        axis[i] = uint8_t(int(double(uint32_t(x) - 1u) / 2147483646.0 * 3.0));
    }
}

int bvh_build(int n, const float* center, const float* motion, const float* radius, uint32_t axis_seed,
              crt_bvh_node* nodes) {
    if (n <= 0) return 0;
    std::vector<float> boxes(6 * size_t(n));
    sphere_boxes(n, center, motion, radius, boxes.data());
    std::vector<uint8_t> axis(2 * size_t(n) - 1);
    axis_table(n, axis_seed, axis.data());

    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    struct job { int start, span; size_t draw; };
    std::vector<job> todo;
    todo.push_back({0, n, 0});
    std::vector<std::pair<float, int>> keyed;
    while (!todo.empty()) {
        const job j = todo.back();
        todo.pop_back();
        if (j.span < 2) continue;
        const int ax = axis[j.draw];
        keyed.resize(j.span);
        for (int i = 0; i < j.span; ++i) keyed[i] = {boxes[6 * size_t(order[j.start + i]) + 2 * ax], order[j.start + i]};
        // strict '<' on the key, stable: equal keys keep their current order
        std::stable_sort(keyed.begin(), keyed.end(), [](const std::pair<float, int>& a, const std::pair<float, int>& b) { return a.first < b.first; });
        for (int i = 0; i < j.span; ++i) order[j.start + i] = keyed[i].second;
        const int left = j.span / 2;
        todo.push_back({j.start + left, j.span - left, j.draw + 2 * size_t(left)});
        todo.push_back({j.start, left, j.draw + 1});
    }

    // complete tree in heap order, root at 0 (bvh.h:64-100)
    const int depth = bvh_depth(n);
    const int L = 1 << depth;
    const float inf = std::numeric_limits<float>::infinity();
    for (int i = 0; i < L; ++i) {
        crt_bvh_node& nd = nodes[L - 1 + i];
        if (i < n) {
            const float* b = &boxes[6 * size_t(order[i])];
            nd = {b[0], b[1], b[2], b[3], b[4], b[5], order[i], -1, -1};
        } else {
            nd = {inf, -inf, inf, -inf, inf, -inf, -1, -1, -1};
        }
    }
    for (int i = L - 2; i >= 0; --i) {
        const crt_bvh_node &l = nodes[2 * i + 1], &r = nodes[2 * i + 2];
        nodes[i] = {l.xmin <= r.xmin ? l.xmin : r.xmin, l.xmax >= r.xmax ? l.xmax : r.xmax,
                    l.ymin <= r.ymin ? l.ymin : r.ymin, l.ymax >= r.ymax ? l.ymax : r.ymax,
                    l.zmin <= r.zmin ? l.zmin : r.zmin, l.zmax >= r.zmax ? l.zmax : r.zmax,
                    -1, 2 * i + 1, 2 * i + 2};
    }
    return 2 * L - 1;
}

// color.h:17-35 for one component
static inline int to_byte(float linear) {
    float g = float(linear > 0 ? std::sqrt(double(linear)) : 0.0);
    if (g < 0.000f) g = 0.000f;
    if (g > 0.999f) g = 0.999f;
    return int(256 * g);
}

int64_t write_ppm(const float* fb, int width, int height, char* buf, int64_t cap) {
    int64_t used = 0;
    char line[64];
    auto put = [&](int len) {
        if (buf && used + len <= cap) memcpy(buf + used, line, size_t(len));
        used += len;
    };
    put(snprintf(line, sizeof line, "P3\n%d %d\n255\n", width, height));  // camera.h:46
    const size_t npx = size_t(width) * size_t(height);
    for (size_t i = 0; i < npx; ++i)
        put(snprintf(line, sizeof line, "%d %d %d\n", to_byte(fb[3 * i]), to_byte(fb[3 * i + 1]), to_byte(fb[3 * i + 2])));
    return used;
}

}  // namespace crt

// ---- C ABI (host-only entry points) ---------------------------------------------------------
extern "C" {

int crt_camera_initialize(const crt_camera_params* params, crt_camera* out) {
    if (!params || !out || params->image_width < 1 || params->samples_per_pixel < 1 || !(params->aspect_ratio > 0))
        return CRT_ERR_INVALID_ARGUMENT;
    crt::camera_initialize(*params, *out);
    return CRT_OK;
}

int crt_scene_small(float* center, float* motion, float* radius, int32_t* mat_id, int32_t* mat_type, float* mat_albedo,
                    float* mat_param, int32_t* n_out) {
    if (!center || !motion || !radius || !mat_id || !mat_type || !mat_albedo || !mat_param || !n_out)
        return CRT_ERR_INVALID_ARGUMENT;
    *n_out = crt::scene_small(center, motion, radius, mat_id, mat_type, mat_albedo, mat_param);
    return CRT_OK;
}

int crt_scene_random(uint64_t seed, int32_t half, float ground_radius, int32_t capacity, float* center, float* motion,
                     float* radius, int32_t* mat_id, int32_t* mat_type, float* mat_albedo, float* mat_param,
                     int32_t* n_out) {
    if (!center || !motion || !radius || !mat_id || !mat_type || !mat_albedo || !mat_param || !n_out || half < 0)
        return CRT_ERR_INVALID_ARGUMENT;
    if (int64_t(capacity) < 4 + 4 * int64_t(half) * half) return CRT_ERR_TOO_LARGE;
    *n_out = crt::scene_random(seed, half, ground_radius, center, motion, radius, mat_id, mat_type, mat_albedo, mat_param);
    return CRT_OK;
}

int64_t crt_write_ppm(const float* fb, int32_t width, int32_t height, char* buf, int64_t cap) {
    if (!fb || width < 0 || height < 0) return -1;
    return crt::write_ppm(fb, width, height, buf, cap);
}

int32_t crt_bvh_num_nodes(int32_t n) {
    if (n <= 0) return 0;
    return (2 << crt::bvh_depth(n)) - 1;
}

int crt_bvh_build_host(int32_t n, const float* center, const float* motion, const float* radius, uint32_t axis_seed,
                       crt_bvh_node* nodes) {
    if (n < 1 || !center || !motion || !radius || !nodes) return CRT_ERR_INVALID_ARGUMENT;
    crt::bvh_build(n, center, motion, radius, axis_seed, nodes);
    return CRT_OK;
}

}  // extern "C"
