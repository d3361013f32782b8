// rt_render.h -- the drop-in boundary: the host calls the reference's CLI makes.
//   render()               helper.h:111-171              brute-force hittable_list world
//   render_experimental()  helper_experimental.h:148-213 bvh_world
//   create_world / create_world2 / destroy_world          helper.h:7-109 (host functions here,
//                                                         <<<1,1>>> kernels in the reference)
// Same argument lists.  Both render calls flatten the scene, hand it to libcrt_hip.so, print the
// PPM through cam->display_frame() and return the seconds around launch + sync by reference.
// Errors behave like cudaCheckErrors (main_experimental.cu:13-23): message on stderr, exit(1).
// The reference seeds cuRAND with time(0); here the seed is crt_host_seed() (default 1, or the
// environment variable CRT_SEED) so that runs are reproducible.  With CRT_DEVICES="0,1,..." (rt_main --gpus N) one
// image is split over several GPUs of the node (crt_group_*, BASELINE.json configs[2]); the frame is bit-identical.
#ifndef CRT_HOST_RT_RENDER_H
#define CRT_HOST_RT_RENDER_H

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/crt_hip.h"
#include "rt_camera.h"
#include "rt_scene.h"

inline int crt_host_split() {  // CRT_SPLIT=tiles: 8x8-tile shares instead of sample ranges (several GPUs only)
    const char* s = getenv("CRT_SPLIT");
    return (s && strcmp(s, "tiles") == 0) ? CRT_SPLIT_TILES : CRT_SPLIT_SAMPLES;
}

// shared tail of render() / render_experimental(): `group` holds the scene (and the BVH for the BVH world)
inline void crt_render_group(crt_group* group, int world_kind, camera* cam, float& timer_seconds) {
    for (int32_t i = 0; i < crt_group_size(group); ++i) crt_set_seed(crt_group_member(group, i), crt_host_seed());
    int rc = crt_group_set_camera(group, &cam->get_derived());
    crtCheckErrors(crt_group_last_error(group), rc, "camera upload failed");
    std::vector<vec3> frame_buffer(size_t(cam->image_width) * size_t(cam->get_image_height()));
    crt_render_stats stats;
    rc = crt_group_render(group, world_kind, cam->samples_per_pixel, 0, cam->max_depth, crt_host_split(),
                          reinterpret_cast<float*>(frame_buffer.data()), &stats);
    crtCheckErrors(crt_group_last_error(group), rc, "kernel launch error");
    cam->display_frame(frame_buffer.data());
    timer_seconds = float(stats.wall_seconds);
}

// helper.h:111-115
inline void render(int pixels_per_block_x, int pixels_per_block_y, camera* cam, hittable** world, float& timer_seconds) {
    (void)pixels_per_block_x; (void)pixels_per_block_y;  // launch geometry is the library's business
    crt_flat_scene flat;
    (*world)->flatten(flat);
    const std::vector<int32_t>& devices = crt_host_devices();
    crt_group* group = nullptr;
    int rc = crt_group_create(devices.data(), int32_t(devices.size()), crt_host_seed(), &group);
    crtCheckErrors(crt_group_last_error(nullptr), rc, "device context creation failed");
    rc = crt_group_upload_scene(group, flat.num_objects(), flat.center.data(), flat.motion.data(), flat.radius.data(),
                                flat.mat_id.data(), int32_t(flat.mat_type.size()), flat.mat_type.data(),
                                flat.mat_albedo.data(), flat.mat_param.data());
    crtCheckErrors(crt_group_last_error(group), rc, "scene upload failed");
    crt_render_group(group, CRT_WORLD_LIST, cam, timer_seconds);
    crt_group_destroy(group);
}

// helper_experimental.h:148-152.  The scene and the tree are already on the GPUs (bvh_world's constructor).
inline void render_experimental(int pixels_per_block_x, int pixels_per_block_y, camera* cam, bvh_world* world,
                                float& timer_seconds) {
    (void)pixels_per_block_x; (void)pixels_per_block_y;
    if (!world->group) {
        fprintf(stderr, "Fatal error: device context creation failed (bvh_world was built without a GPU at %s:%d)\n", __FILE__, __LINE__);
        fprintf(stderr, "*** FAILED - ABORTING\n");
        exit(1);
    }
    crt_render_group(world->group, CRT_WORLD_BVH, cam, timer_seconds);
}

// Builds host objects from the flat arrays of the library's scene generators.
inline void crt_objects_from_flat(int n, const float* center, const float* motion, const float* radius,
                                  const int32_t* mat_type, const float* albedo, const float* param,
                                  hittable** world, material_list** mat_lst) {
    material** materials = new material*[n];
    hittable** objects = new hittable*[n];
    for (int i = 0; i < n; ++i) {
        const color a(albedo[3 * i], albedo[3 * i + 1], albedo[3 * i + 2]);
        if (mat_type[i] == CRT_LAMBERTIAN) materials[i] = new lambertian(a);
        else if (mat_type[i] == CRT_METAL) materials[i] = new metal(a, param[i]);
        else materials[i] = new dielectric(param[i]);
        const point3 c(center[3 * i], center[3 * i + 1], center[3 * i + 2]);
        const vec3 mo(motion[3 * i], motion[3 * i + 1], motion[3 * i + 2]);
        if (mo.x() == 0 && mo.y() == 0 && mo.z() == 0) objects[i] = new sphere(c, radius[i], materials[i]);
        else objects[i] = new sphere(c, c + mo, radius[i], materials[i]);
    }
    *mat_lst = new material_list(materials, n);
    *world = new hittable_list(objects, n);
}

// helper.h:7-35: ground + three spheres (one hollow)
inline void create_world(hittable** world, material_list** mat_lst) {
    float center[15], motion[15], radius[5], albedo[15], param[5];
    int32_t mat_id[5], mat_type[5], n = 0;
    crt_scene_small(center, motion, radius, mat_id, mat_type, albedo, param, &n);
    crt_objects_from_flat(n, center, motion, radius, mat_type, albedo, param, world, mat_lst);
}

// helper.h:37-102: the random-spheres scene
inline void create_world2(hittable** world, material_list** mat_lst, uint64_t scene_seed = 18, int half = 11,
                          float ground_radius = 1000.0f) {
    const int cap = 4 + 4 * half * half;
    std::vector<float> center(3 * cap), motion(3 * cap), radius(cap), albedo(3 * cap), param(cap);
    std::vector<int32_t> mat_id(cap), mat_type(cap);
    int32_t n = 0;
    crt_scene_random(scene_seed, half, ground_radius, cap, center.data(), motion.data(), radius.data(), mat_id.data(),
                     mat_type.data(), albedo.data(), param.data(), &n);
    crt_objects_from_flat(n, center.data(), motion.data(), radius.data(), mat_type.data(), albedo.data(), param.data(),
                          world, mat_lst);
}

// helper.h:104-109
inline void destroy_world(hittable** world, material_list** mat_lst) {
    hittable_list* list = static_cast<hittable_list*>(*world);
    for (int i = 0; i < list->size; ++i) delete list->objects[i];
    delete[] list->objects;
    delete *world;
    delete *mat_lst;
}

#endif
