// crt_device.h -- device-side data layout shared by the kernels and the host launcher.
//
// HBM layout (all arrays 16-byte aligned, read-only during a render):
//
//  pairs   : float4[4 * L]   L = number of leaves of the complete tree (power of two).
//            Record k (1 <= k < L, 64 B) holds the TWO children of heap node k
//            (1-based heap: children of k are 2k and 2k+1; the reference's 0-based
//            node i is k = i+1, bvh.h:87-95):
//              q0 = { L.xmin, L.xmax, R.xmin, R.xmax }
//              q1 = { L.ymin, L.ymax, R.ymin, R.ymax }
//              q2 = { L.zmin, L.zmax, R.zmin, R.zmax }
//              q3 = { tagL, tagR, -, - } (int bits)  tag >= 0: sphere index of a leaf,
//                   CRT_TAG_INTERNAL: internal node, CRT_TAG_EMPTY: padding / all-padding subtree.
//            Child indices are implicit, so the traversal keeps no node stack: a node index
//            and one 32-bit "right sibling pending" mask replace the reference's
//            bvh_node* stack[32] (helper_experimental.h:11-13).
//  spheres : float4[2 * n]   { cx, cy, cz, radius }, { mx, my, mz, mat_id | type << 28 }
//            (sphere.h:38-41: center ray origin, direction, radius, mat_ptr)
//  mats    : float4[m]       lambertian/metal { r, g, b, fuzz }; dielectric { ior, 1/ior, r0(ior), r0(1/ior) }, r0(x) = ((1-x)/(1+x))^2 (material.h:94-95)
//  sums    : int64[3 * W * H] un-normalised RGB sums in 2^-40 fixed point, row-major, RGB interleaved
//            (the element order of helper.h:139-141's vec3 fb[W*H])
#ifndef CRT_DEVICE_H
#define CRT_DEVICE_H

#include <stdint.h>

#define CRT_TAG_INTERNAL (-1)
#define CRT_TAG_EMPTY (-2)
#define CRT_MAT_TYPE_SHIFT 28
#define CRT_MAT_ID_MASK 0x0fffffff

// scheduled kernel: sphere records staged in LDS (1) or read from HBM through L1/L2 (0)
#ifndef CRT_SCHED_SPHERES_IN_LDS
#define CRT_SCHED_SPHERES_IN_LDS 0
#endif

// counter buffer of a launch (unsigned long long words):
//   [0, 32)                            census, work-queue counter [15], phase cycles, timeline sums (crt_render_args::counters);
//                                      [29] exchange watchdog of SCHED=3 (must stay 0)
//   [CRT_STATS_BASE, + 4 * STRIPES)    rays, samples, pair visits, sphere tests -- striped over 64 slots by workgroup: every
//                                      wave adds its totals once, and thousands of atomics on ONE address serialise in L2
//                                      (10 ns each: 0.28 ms for the 28 000 of a small frame's launch)
//   [CRT_TIMELINE_BASE, + 3 * 8192)    (start, queue empty, exit) of the first 8192 waves (census only)
#define CRT_STAT_STRIPES 64
#define CRT_STATS_BASE 32
#define CRT_TIMELINE_BASE (CRT_STATS_BASE + 4 * CRT_STAT_STRIPES)
#define CRT_COUNTER_WORDS (CRT_TIMELINE_BASE + 3 * 8192)

// SCHED=3 exchange area of one owner wave in LDS (crt_kernels.hip walker_loop): a request ring of CRT_RING_CAP entries of two
// float4 (SoA), one float4 of control words {tail, head, done, -}, 64 mailbox words
#define CRT_RING_CAP 32u
#define CRT_XCHG_FLOAT4_PER_OWNER (2 * CRT_RING_CAP + 1 + 16)

struct crt_dev_camera {
    float center[3], p00[3], du[3], dv[3], ddu[3], ddv[3];
    float defocus_angle;
    int width, height;
};

struct crt_render_args {
    crt_dev_camera cam;
    const void* pairs;
    const void* spheres;
    const void* mats;
    int num_leaves;  // L
    int n;           // spheres
    int m;           // materials
    uint32_t key0, key1;
    int s0, ns;      // first sample index, sample count of this call
    int tiles_per_wg;  // scheduled kernel: 8x8 tiles per workgroup (power of two)
    int tiles_x, tiles_y;
    // tile share of this launch (multi-GPU tile split): it traces the 8x8 tiles t = tile_offset + j * tile_stride,
    // j in [0, n_tiles); the whole frame is stride 1, offset 0, n_tiles = tiles_x * tiles_y
    int tile_stride, tile_offset, n_tiles;
    int simple_chunks;       // simple loop (list world, SCHED=0): a tile's ns samples are cut into this many chunks, one wave each
    int max_depth;
    int sched;               // 1: wave-scheduled BVH tracer, 0: simple per-lane loop, 3: wave-scheduled with walker waves
    int census;              // COUNT variant only: also count scheduler iterations / served lanes per phase
    int run_shift;           // leave the box-walking loop once n_run >> run_shift lanes have left it
    int leaf_num, shade_num; // run the sphere / shade phase once >= num/16 of the live lanes wait for it
    int persist;             // 1: persistent waves pulling (tile, sample chunk) units from counters[15]
    // persistent: a tile's ns samples are cut into unit_chunks chunks of unit_samples (covering
    // samples [0, ns_coarse)) followed by fine_chunks chunks of fine_samples (the rest); all coarse
    // units come first (tile-major), then all fine units: small units at the end of the launch
    // keep its tail short, large ones before that keep the flush traffic low
    int unit_samples;        // samples per coarse unit (1..64)
    int unit_chunks;         // coarse units per tile
    int ns_coarse;           // samples covered by the coarse units (unit_chunks * unit_samples, <= ns)
    int n_coarse;            // n_tiles * unit_chunks
    int fine_samples;        // samples per fine unit (1..64)
    int fine_chunks;         // fine units per tile = ceil((ns - ns_coarse) / fine_samples)
    int n_units;             // n_coarse + n_tiles * fine_chunks
    int top_pairs;           // scheduled, scene in HBM: pair records [0, top_pairs) are also staged in LDS (power of two <= L, or 0)
    int tail_pass;           // persistent: whole-wave speculative draws for the last rejection-sampling lanes
    // SCHED=3 (sched == 3): the first n_walkers waves of a workgroup only walk boxes for the others (crt_kernels.hip walker_loop)
    int n_walkers;           // walker waves per workgroup
    int refill_min;          // a walker refills its idle lanes from the request rings once this many are idle
    int owner_sleep;         // an owner wave (or a wholly idle walker) with nothing to do sleeps this many times 64 clocks between polls
    int walker_backoff;      // a walker that found its rings empty walks this many steps before it looks again
    int walker_prio;         // 1: walker waves run at raised priority (s_setprio 2)
    int walker_min_busy;     // a walker with fewer rays on board polls for more (at most walker_patience times in a row) before it steps; 0: off
    int walker_patience;
    unsigned long long* sums;  // [3 * W * H] fixed-point (2^-40) RGB sums, row-major
    unsigned long long* counters;  // layout above: [4..14] census [15] next unit [16..21] phase cycles [22..28] timeline census; statistics striped from CRT_STATS_BASE; per-wave timeline from CRT_TIMELINE_BASE
};

#endif
