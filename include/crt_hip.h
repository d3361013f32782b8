/* crt_hip.h -- C ABI of libcrt_hip.so, the MI355X (gfx950) implementation of the one hot path
 * of ChenVand/CUDA_Ray_Tracer: the per-sample Monte Carlo loop of render_with_cuda.h
 * (get_ray -> closest hit over spheres / flat BVH -> lambertian|metal|dielectric scatter
 * chain -> per-pixel accumulation).
 *
 * The reference has no FFI; its boundary is the host C++ call `render()` /
 * `render_experimental()` plus the scene classes the CLI constructs.  The
 * host classes in cuda_ray_tracer_amd/host/ keep those names and flatten onto the entry
 * points below.  Every entry point cites the reference interface it replaces
 * (file:line relative to the reference root).  Plain pointers and sizes only.
 *
 * Conventions: every function returns 0 (CRT_OK) or a CRT_ERR_* code and never calls
 * exit(); crt_last_error() gives the message.  A context is not re-entrant; separate
 * contexts are independent.  All float data is fp32.  Framebuffers are W*H*3 floats,
 * row-major, RGB interleaved -- the layout of the reference's `vec3 fb[W*H]`
 * (helper.h:139-141).
 */
#ifndef CRT_HIP_H
#define CRT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRT_OK 0
#define CRT_ERR_INVALID_ARGUMENT 1
#define CRT_ERR_NO_DEVICE 2
#define CRT_ERR_HIP 3
#define CRT_ERR_STATE 4
#define CRT_ERR_TOO_LARGE 5

/* material.h:7-98 -- the class hierarchy, flattened to a tag */
#define CRT_LAMBERTIAN 0
#define CRT_METAL 1
#define CRT_DIELECTRIC 2
#define CRT_MATERIAL_BASE 3 /* material.h:11-19: the base class, scatter() returns false -- the path ends black */

/* arithmetic of the render kernels (crt_set_arith) */
#define CRT_ARITH_EXACT 0 /* every fp32 operation rounded separately, in the order of the reference's C++ expressions:
                             a path is bit-identical to the CPU oracle (the parity suite's mode; default) */
#define CRT_ARITH_FAST 1  /* tolerance mode: the slab test of aabb.h:48-62 as one fma per plane, t = fma(plane, 1/d, -o/d), fp
                             contraction, 1-ulp hardware rcp / sqrt / rsq; images agree with EXACT within the tolerance
                             stated in DESIGN.md (PSNR >= 45 dB at 500 spp), not bit for bit; +16 % rays/s */

/* how crt_group_render cuts one image over the group's devices */
#define CRT_SPLIT_SAMPLES 0 /* member g traces samples [g spp/G, (g+1) spp/G) of every pixel (north_star) */
#define CRT_SPLIT_TILES 1   /* member g traces every sample of the 8x8 tiles t = g mod G */

/* which `world` the bounce loop queries */
#define CRT_WORLD_LIST 0 /* hittable_list::hit, hittable_list.h:43-58  (render(),              helper.h:111)             */
#define CRT_WORLD_BVH 1  /* external_hit,       helper_experimental.h:1-50 (render_experimental(), helper_experimental.h:148) */

typedef struct crt_context crt_context;

/* camera.h:11-22 -- the public configuration fields */
typedef struct crt_camera_params {
    float aspect_ratio;
    int32_t image_width;
    int32_t samples_per_pixel;
    float vfov;
    float lookfrom[3];
    float lookat[3];
    float vup[3];
    float defocus_angle;
    float focus_dist;
} crt_camera_params;

/* camera.h:60-68 -- what initialize() derives and the kernel reads through the getters
 * (camera.h:26-40) */
typedef struct crt_camera {
    int32_t image_width;
    int32_t image_height;
    float pixel_samples_scale;
    float center[3];
    float pixel00_loc[3];
    float pixel_delta_u[3];
    float pixel_delta_v[3];
    float defocus_disk_u[3];
    float defocus_disk_v[3];
    float defocus_angle;
} crt_camera;

/* bvh.h:47-62 -- bvh_node, same field order and size (36 B).  Fields the reference
 * leaves uninitialised (children of leaves; loc/children of padding leaves) read -1. */
typedef struct crt_bvh_node {
    float xmin, xmax, ymin, ymax, zmin, zmax; /* aabb: interval x, y, z */
    int32_t loc_for_leaf;
    int32_t left_child_loc;
    int32_t right_child_loc;
} crt_bvh_node;

typedef struct crt_render_stats {
    double kernel_seconds;  /* device time of the render kernel(s) of the last call (HIP events on its stream) */
    double wall_seconds;    /* host wall time of the last crt_render call (helper.h:144-170 timer_seconds) */
    uint64_t rays;          /* closest-hit queries = world.hit calls, render_with_cuda.h:41 */
    uint64_t samples;       /* (pixel, sample) paths traced */
    uint64_t pair_visits;   /* BVH internal nodes visited (two child boxes tested each); counted only with crt_set_count_stats(1) */
    uint64_t sphere_tests;  /* sphere::hit evaluations; counted only with crt_set_count_stats(1) */
} crt_render_stats;

/* ---- host-side helpers (no GPU needed) ------------------------------------------------ */

/* camera::initialize(), camera.h:72-107 */
int crt_camera_initialize(const crt_camera_params* params, crt_camera* out);

/* create_world, helper.h:7-35 (same five spheres as create_objects, main_experimental.cu:38-54).
 * Arrays need room for 5 spheres / 5 materials.  *n_out <- 5. */
int crt_scene_small(float* center, float* motion, float* radius, int32_t* mat_id, int32_t* mat_type,
                    float* mat_albedo, float* mat_param, int32_t* n_out);

/* create_world2, helper.h:37-102, generalised: (2*half)^2 grid cells (half = 11 in the
 * reference), ground sphere of `ground_radius` (1000 in the reference).  One material per
 * sphere.  `capacity` >= 4 + 4*half*half.  The reference seeds cuRAND with 18
 * (helper.h:41); `seed` keys the Philox scene stream. */
int crt_scene_random(uint64_t seed, int32_t half, float ground_radius, int32_t capacity, float* center,
                     float* motion, float* radius, int32_t* mat_id, int32_t* mat_type, float* mat_albedo,
                     float* mat_param, int32_t* n_out);

/* camera::display_frame + write_color, camera.h:44-57, color.h:9-35: P3 text PPM of a
 * normalised framebuffer.  Returns bytes needed; writes at most `cap` bytes when buf != NULL. */
int64_t crt_write_ppm(const float* fb, int32_t width, int32_t height, char* buf, int64_t cap);

/* bvh_world ctor on plain host arrays (bvh.h:135-192, :251-274, :64-100): the node array
 * for n spheres, without a context or a GPU.  nodes needs crt_bvh_num_nodes(n) entries. */
int32_t crt_bvh_num_nodes(int32_t n);
int crt_bvh_build_host(int32_t n, const float* center, const float* motion, const float* radius,
                       uint32_t axis_seed, crt_bvh_node* nodes);

/* ---- context ---------------------------------------------------------------------------- */

/* Replaces the implicit CUDA context + time(0) seeding (helper.h:134).  `seed` keys every
 * per-(pixel, sample) Philox stream. */
int crt_create(int32_t device, uint64_t seed, crt_context** out);
int crt_destroy(crt_context* ctx);
/* message of the last failure on ctx (or of the last failed crt_create when ctx == NULL) */
const char* crt_last_error(const crt_context* ctx);
int crt_set_seed(crt_context* ctx, uint64_t seed);

/* Replaces create_world<<<1,1>>> / create_objects<<<1,1>>> + the device-side `new`s
 * (helper.h:7-102, main_experimental.cu:38-54, :121-132): n spheres as flat arrays
 * (center[3n], motion[3n] = center2 - center1 or 0, radius[n], mat_id[n]) and m materials
 * (type[m], albedo[3m], param[m] = fuzz | refraction_index).  The library copies; the
 * caller keeps ownership.  Drops any BVH built before. */
int crt_upload_scene(crt_context* ctx, int32_t n, const float* center, const float* motion, const float* radius,
                     const int32_t* mat_id, int32_t m, const int32_t* mat_type, const float* mat_albedo,
                     const float* mat_param);

/* bvh_world::bvh_world(object_list, n), bvh.h:135-192.  axis_seed is 17 in the reference
 * (bvh.h:144).  *num_nodes_out <- 2^(depth+1)-1.  Built on the GPU (one stable sort per
 * tree level); the node array equals crt_bvh_build_host's bit for bit. */
int crt_build_bvh(crt_context* ctx, uint32_t axis_seed, int32_t* num_nodes_out);
/* read-back of bvh_world::d_nodes (bvh.h:131) for the bit-exact check */
int crt_get_bvh_nodes(const crt_context* ctx, crt_bvh_node* out, int32_t capacity);
/* The other half of "build or upload": hands the context a node array the caller already holds (the public
 * bvh_world::d_nodes / num_nodes of bvh.h:129-131, e.g. from crt_bvh_build_host or read back from another
 * context) instead of building it again.  `count` must be crt_bvh_num_nodes(n) for the uploaded scene, nodes in
 * heap order (root 0, children of i at 2i+1 / 2i+2, bvh.h:87-95); leaves carry loc_for_leaf in [0, n) or an
 * empty box (min > max, bvh.h:82).  Boxes are taken as given. */
int crt_upload_bvh_nodes(crt_context* ctx, const crt_bvh_node* nodes, int32_t count);

/* the managed `camera*` the kernels dereference (main_experimental.cu:99-116) */
int crt_set_camera(crt_context* ctx, const crt_camera* cam);

/* ---- the hot path ----------------------------------------------------------------------- */

/* render(), helper.h:111-171 (world = CRT_WORLD_LIST) and render_experimental(),
 * helper_experimental.h:148-213 (world = CRT_WORLD_BVH): traces samples
 * [sample_offset, sample_offset + spp) of every pixel and writes the normalised framebuffer
 * (sum * 1/spp) to the caller's host buffer.
 * max_depth is the bounce limit (50 in render_with_cuda.h:33). */
int crt_render(crt_context* ctx, int32_t world, int32_t spp, int32_t sample_offset, int32_t max_depth,
               float* fb_host, crt_render_stats* stats);

/* The same launch without the host round trip: ADDS the un-normalised per-pixel sums of samples
 * [sample_offset, sample_offset + nsamples) to the device buffer d_sums on `stream` (a
 * hipStream_t; NULL = default stream).  Asynchronous.  d_sums is W*H*3 int64, row-major, RGB
 * interleaved, in 2^-40 fixed point: each sample's radiance component r contributes
 * (int64)(min(r, 4096) * 2^40).  Integer sums are associative, so the buffer -- and the image --
 * is bit-identical however the samples are split over calls, workgroups or GPUs.  This is what
 * multi-GPU sample splitting and progressive accumulation call.
 * Launches of ONE context share its work-queue counter and statistics: issue them on one stream
 * (or otherwise ordered); launches that overlap on the GPU need a context each. */
int crt_render_accumulate(crt_context* ctx, int32_t world, int32_t sample_offset, int32_t nsamples,
                          int32_t max_depth, int64_t* d_sums, void* stream);

/* crt_render_accumulate restricted to a share of the frame's 8x8 tiles (row-major tile index t over
 * ceil(W/8) x ceil(H/8)): traces the tiles t = tile_offset + j * tile_stride only.  tile_stride >= 1,
 * 0 <= tile_offset < tile_stride; (1, 0) is the whole frame.  The multi-GPU tile split (BASELINE.json
 * configs[4] "per-GPU tile/sample balancing"): member g of G calls it with (G, g); the shares are disjoint and
 * their union is the frame.  No counterpart in the reference (single GPU, helper.h:111-171). */
int crt_render_accumulate_tiles(crt_context* ctx, int32_t world, int32_t sample_offset, int32_t nsamples,
                                int32_t max_depth, int32_t tile_stride, int32_t tile_offset, int64_t* d_sums,
                                void* stream);

/* crt_render for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  Same image as crt_render, bit for bit, written
 * as W*H*3 fp32 to the caller's DEVICE buffer d_fb, asynchronously on `stream`.  The work-queue counter, the
 * statistics and the int64 sums of the frame live in a static per-GPU scratch array inside the library (about 48 MiB,
 * held by every GPU this entry point has been called on until the process ends) instead of allocated buffers: the
 * call makes no hipMalloc / hipFree.  As with any kernel, the FIRST call on a GPU makes the runtime load code objects
 * (the render kernels' and the array's): make one warm-up call before entering a region that must not allocate.  Limit: frames of at most CRT_STATIC_MAX_PIXELS pixels; larger ones are rejected
 * with CRT_ERR_TOO_LARGE.  All calls on one GPU share that GPU's array -- calls that overlap on one GPU (two streams,
 * two contexts, two threads) race; order them.  Every state / argument check runs before any stream work: a
 * rejected call enqueues nothing (under capture it adds no node).  A call made while `stream` is capturing records no
 * timing events (they would become graph nodes): crt_get_stats then returns the counters with kernel_seconds = 0 and
 * crt_get_kernel_times reports 0 for that launch; graph replays are never timed by the library. */
#define CRT_STATIC_MAX_PIXELS (1 << 21)
int crt_render_static(crt_context* ctx, int32_t world, int32_t spp, int32_t sample_offset, int32_t max_depth,
                      float* d_fb, void* stream);

/* cam.pixel_samples_scale (camera.h:76, render_with_cuda.h:164): device buffer of fixed-point
 * sums over `spp` samples -> fp32 frame, fb[i] = (float)((double)d_sums[i] * (2^-40 / spp)). */
int crt_resolve(crt_context* ctx, const int64_t* d_sums, int32_t spp, float* d_fb, void* stream);

/* Waits for the last launch and returns its statistics. */
int crt_get_stats(crt_context* ctx, crt_render_stats* stats);

/* Device time (HIP events recorded on the launch stream around each render launch) of the
 * `count` most recent crt_render / crt_render_accumulate launches, oldest first; the library keeps
 * the last 64.  Waits for those launches. */
int crt_get_kernel_times(crt_context* ctx, int32_t count, double* seconds_out);

/* enable (1) / disable (0) the pair_visits / sphere_tests counters (slower kernel variant) */
int crt_set_count_stats(crt_context* ctx, int32_t enable);

/* CRT_ARITH_EXACT (default; the environment variable CRT_ARITH=fast changes the default) or CRT_ARITH_FAST */
int crt_set_arith(crt_context* ctx, int32_t arith);

/* Which code ran: template instance of the render kernel of the last launch on ctx ("render_kernel<...>", static
 * string) and a hash of the kernel sources this library was built from (bench.py ties committed profiles to it). */
const char* crt_last_kernel_name(const crt_context* ctx);
const char* crt_build_id(void);

/* Self-test of the exact mode's reciprocal, square root and reciprocal square root.  The reference divides and calls
 * sqrt (1 / direction component, aabb.h:48; v / t = (1/t) * v and unit_vector, vec3.h:114-132; 1 / radius and the
 * discriminant's root, sphere.h:55-66; material.h:73,  vec3.h:141).  For arguments of ordinary size -- |x| in
 * [2^-100, 2^100] for 1/x, x in [2^-96, inf) for the root -- the kernels use short sequences (v_rcp_f32 + one Newton
 * step, 3 instructions for the general division's 11; v_sqrt_f32 + the two one-ulp neighbours tried by their
 * residuals, 9 for 16) and the general ones for the rest.  This call proves on ctx's GPU that the results are the
 * same: it evaluates both forms of all three functions for EVERY one of the 2^32 arguments.  The fourth function is the
 * conversion of a sample's radiance to the frame's 2^-40 fixed point (render_with_cuda.h:128-165 accumulates floats;
 * DESIGN.md "Accumulation contract"): for a radiance that is not negative the kernels split the integer with one floor
 * and one exact fma instead of the compiler's signed float -> int64 sequence; every such argument is compared.
 * out12 = for 1/x, sqrt(x), 1/sqrt(x), fixed(x) in turn: {number of arguments whose results differ as bit patterns
 * (two NaNs count as equal) -- must be 0, the smallest such argument's bits (~0 if none), number of arguments on the
 * short sequence}.  No counterpart in the reference.  ~2 s. */
int crt_selftest_exact_math(crt_context* ctx, uint64_t* out12);

/* Debug/parity: primary ray (7 floats), radiance (3 floats) and number of closest-hit
 * queries of single (pixel, sample) paths, computed by the same device functions as the
 * render kernel.  px/py/sample are arrays of `count` entries; outputs are host buffers. */
int crt_trace_samples(crt_context* ctx, int32_t world, int32_t count, const int32_t* px, const int32_t* py,
                      const int32_t* sample, int32_t max_depth, float* ray7_out, float* rgb_out,
                      int32_t* segments_out);

/* ---- several GPUs of one node behind one handle ------------------------------------------- */

/* The reference is single-GPU (helper_experimental.h:148-213); BASELINE.json configs[2] splits one image over the
 * GPUs of a node.  A group holds one context per entry of `devices` (an index may repeat: contexts then share that
 * GPU); scene, BVH and camera calls go to every member (on one host thread each), crt_group_render cuts the image
 * by `split`, traces the shares concurrently, sums the int64 buffers on devices[0] (peer copies over xGMI + one
 * add kernel; integer sums, so the frame is bit-identical to a single context's for every group size and either
 * split) and writes the normalised frame to the caller's host buffer like crt_render.  stats: rays / samples /
 * counters summed over the members, kernel_seconds = the slowest member's launch. */
typedef struct crt_group crt_group;
int crt_group_create(const int32_t* devices, int32_t n, uint64_t seed, crt_group** out);
int crt_group_destroy(crt_group* grp);
int32_t crt_group_size(const crt_group* grp);
crt_context* crt_group_member(crt_group* grp, int32_t i); /* for the per-context calls (knobs, statistics) */
const char* crt_group_last_error(const crt_group* grp);   /* grp == NULL: last failed crt_group_create */
int crt_group_upload_scene(crt_group* grp, int32_t n, const float* center, const float* motion, const float* radius,
                           const int32_t* mat_id, int32_t m, const int32_t* mat_type, const float* mat_albedo,
                           const float* mat_param);
int crt_group_build_bvh(crt_group* grp, uint32_t axis_seed, int32_t* num_nodes_out);
int crt_group_upload_bvh_nodes(crt_group* grp, const crt_bvh_node* nodes, int32_t count);
int crt_group_set_camera(crt_group* grp, const crt_camera* cam);
int crt_group_render(crt_group* grp, int32_t world, int32_t spp, int32_t sample_offset, int32_t max_depth,
                     int32_t split, float* fb_host, crt_render_stats* stats);

#ifdef __cplusplus
}
#endif
#endif /* CRT_HIP_H */
