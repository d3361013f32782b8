// host_api_check.cpp -- compiled and run by tests/test_host_cpp_api.py (CPU only): the C++ mirror
// of the reference's scene API must flatten to exactly what the library's generators produce, and
// bvh_world must hold the library's node array.
#include <cstdio>
#include <cstring>
#include <vector>

#include "../cuda_ray_tracer_amd/host/rt_render.h"

static int fails = 0;
#define CHECK(c) do { if (!(c)) { printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

static void check_scene(bool random_scene) {
    hittable* world = nullptr;
    material_list* mats = nullptr;
    if (random_scene) create_world2(&world, &mats); else create_world(&world, &mats);
    crt_flat_scene flat;
    world->flatten(flat);
    const int cap = 4 + 4 * 11 * 11;
    std::vector<float> center(3 * cap), motion(3 * cap), radius(cap), albedo(3 * cap), param(cap);
    std::vector<int32_t> mat_id(cap), mat_type(cap);
    int32_t n = 0;
    if (random_scene) crt_scene_random(18, 11, 1000.0f, cap, center.data(), motion.data(), radius.data(), mat_id.data(), mat_type.data(), albedo.data(), param.data(), &n);
    else crt_scene_small(center.data(), motion.data(), radius.data(), mat_id.data(), mat_type.data(), albedo.data(), param.data(), &n);
    CHECK(flat.num_objects() == n);
    CHECK(memcmp(flat.center.data(), center.data(), 12 * n) == 0);
    CHECK(memcmp(flat.motion.data(), motion.data(), 12 * n) == 0);
    CHECK(memcmp(flat.radius.data(), radius.data(), 4 * n) == 0);
    CHECK(memcmp(flat.mat_type.data(), mat_type.data(), 4 * n) == 0);
    CHECK(memcmp(flat.mat_param.data(), param.data(), 4 * n) == 0);
    hittable_list* list = static_cast<hittable_list*>(world);
    bvh_world bw(list->objects, list->size);
    CHECK(bw.num_objects == n && bw.num_nodes == crt_bvh_num_nodes(n));
    std::vector<crt_bvh_node> nodes(bw.num_nodes);
    crt_bvh_build_host(n, center.data(), motion.data(), radius.data(), 17u, nodes.data());
    CHECK(memcmp(nodes.data(), bw.d_nodes, sizeof(crt_bvh_node) * nodes.size()) == 0);
    // the list's own bounding box is the union of the spheres' boxes == the BVH root box
    const aabb b = world->bounding_box();
    CHECK(b.x.min == nodes[0].xmin && b.x.max == nodes[0].xmax && b.y.min == nodes[0].ymin && b.z.max == nodes[0].zmax);
    destroy_world(&world, &mats);
}

int main() {
    check_scene(false);
    check_scene(true);
    camera cam;
    cam.aspect_ratio = 16.0 / 9.0; cam.image_width = 400; cam.samples_per_pixel = 10; cam.vfov = 20;
    cam.lookfrom = point3(13, 2, 3); cam.lookat = point3(0, 0, 0); cam.defocus_angle = 0.6; cam.focus_dist = 10.0;
    cam.initialize();
    CHECK(cam.get_image_height() == 225);
    CHECK(cam.get_pixel_samples_scale() == float(1.0 / 10));
    CHECK(cam.get_center().x() == 13 && cam.get_center().y() == 2);
    sphere moving(point3(0, 0, 0), point3(0, 0.5f, 0), 0.2f, nullptr), copy(moving);
    CHECK(copy.bounding_box().y.max == 0.7f && copy.bounding_box().y.min == -0.2f);
    material base;  // material.h:11-19: never scatters
    CHECK(base.flatten().type == CRT_MATERIAL_BASE);
    metal m(color(1, 1, 1), 3.0f);
    CHECK(m.flatten().param == 1.0f);  // fuzz clamp, material.h:46
    // shared material pointers flatten to one material
    lambertian shared(color(0.5f, 0.5f, 0.5f));
    sphere s1(point3(0, 0, 0), 1, &shared), s2(point3(3, 0, 0), 1, &shared);
    hittable* objs[2] = {&s1, &s2};
    hittable_list two(objs, 2);
    crt_flat_scene flat;
    two.flatten(flat);
    CHECK(flat.mat_type.size() == 1 && flat.mat_id[0] == 0 && flat.mat_id[1] == 0);
    printf(fails ? "FAILED %d\n" : "OK\n", fails);
    return fails != 0;
}
