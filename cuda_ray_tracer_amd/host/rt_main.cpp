// rt_main.cpp -- the reference's CLI (main_experimental.cu:64-174) on the MI355X path.
//   --width N     image width (default 400)
//   --samples N   samples per pixel, rounded UP to a multiple of 32 like main_experimental.cu:72
//                 (default 128); --exact-samples keeps N as given
// and, beyond the reference: --scene small|random|grid:HALF, --world bvh|list, --depth N, --seed N, --gpu N,
// --gpus N (one image split over GPUs 0..N-1; CRT_SPLIT=tiles for tile shares instead of sample ranges).
// The PPM goes to stdout, the two status lines to stderr, as in the reference.
#include <chrono>
#include <cstring>
#include <iostream>

#include "rt_render.h"

int main(int argc, char* argv[]) {
    size_t g_image_width = 400;
    size_t g_samples_per_pixel = 128;
    bool exact_samples = false, use_bvh = true, random_scene = false;
    int depth = 50, grid_half = 11;
    for (int i = 1; i < argc; ++i) {
        if (strcmp(argv[i], "--width") == 0 && i + 1 < argc) { g_image_width = atoi(argv[++i]); }
        else if (strcmp(argv[i], "--samples") == 0 && i + 1 < argc) { g_samples_per_pixel = atoi(argv[++i]); }
        else if (strcmp(argv[i], "--exact-samples") == 0) { exact_samples = true; }
        else if (strcmp(argv[i], "--depth") == 0 && i + 1 < argc) { depth = atoi(argv[++i]); }
        else if (strcmp(argv[i], "--seed") == 0 && i + 1 < argc) { crt_host_seed() = strtoull(argv[++i], nullptr, 10); }
        else if (strcmp(argv[i], "--gpu") == 0 && i + 1 < argc) { crt_host_devices().assign(1, atoi(argv[++i])); }
        else if (strcmp(argv[i], "--gpus") == 0 && i + 1 < argc) {
            const int n = atoi(argv[++i]);
            if (n < 1 || n > 16) { std::cerr << "Invalid --gpus\n"; return 1; }
            crt_host_devices().clear();
            for (int g = 0; g < n; ++g) crt_host_devices().push_back(g);
        }
        else if (strcmp(argv[i], "--scene") == 0 && i + 1 < argc) {
            const char* v = argv[++i];
            random_scene = strcmp(v, "random") == 0 || strncmp(v, "grid:", 5) == 0;
            if (strncmp(v, "grid:", 5) == 0) grid_half = atoi(v + 5);  // the generator on a (2 HALF)^2 grid (C4: grid:500)
        }
        else if (strcmp(argv[i], "--world") == 0 && i + 1 < argc) { use_bvh = strcmp(argv[++i], "list") != 0; }
        else {
            std::cerr << "Unknown argument: " << argv[i] << "\n";
            return 1;
        }
    }
    if (!exact_samples) g_samples_per_pixel = (g_samples_per_pixel + 31) / 32 * 32;  // main_experimental.cu:72
    if (g_image_width < 1 || g_samples_per_pixel < 1 || depth < 1 || grid_half < 1) {
        std::cerr << "Invalid --width / --samples / --depth / --scene grid:HALF\n";
        return 1;
    }
    const int g_threads_x = int(g_samples_per_pixel);
    const int g_threads_y = std::min(8, std::max(1, 1024 / g_threads_x));

    camera cam;  // main_experimental.cu:103-116
    cam.aspect_ratio = 16.0 / 9.0;
    cam.image_width = int(g_image_width);
    cam.samples_per_pixel = int(g_samples_per_pixel);
    cam.max_depth = depth;
    cam.vfov = 20;
    cam.lookfrom = point3(13, 2, 3);
    cam.lookat = point3(0, 0, 0);
    cam.vup = vec3(0, 1, 0);
    cam.defocus_angle = 0.6;
    cam.focus_dist = 10.0;
    cam.initialize();

    hittable* world_list = nullptr;
    material_list* materials = nullptr;
    if (random_scene && grid_half != 11) create_world2(&world_list, &materials, 18, grid_half, grid_half > 100 ? 1.0e5f : 1000.0f);
    else if (random_scene) create_world2(&world_list, &materials);
    else create_world(&world_list, &materials);

    const int pixels_per_block_x = int((g_threads_x + g_samples_per_pixel - 1) / g_samples_per_pixel);
    const int pixels_per_block_y = g_threads_y;
    std::cerr << "Rendering width " << cam.image_width << " image ";
    std::cerr << "with " << pixels_per_block_x * cam.samples_per_pixel << "x" << pixels_per_block_y << " blocks.\n";

    float buffer_gen_time = 0;
    if (use_bvh) {
        hittable_list* list = static_cast<hittable_list*>(world_list);
        const auto t_build = std::chrono::steady_clock::now();
        bvh_world* world = new bvh_world(list->objects, list->size);  // main_experimental.cu:137
        if (getenv("CRT_VERBOSE"))
            std::cerr << "bvh_world: " << world->num_objects << " objects, " << world->num_nodes << " nodes in "
                      << std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_build).count() << " ms\n";
        render_experimental(pixels_per_block_x, pixels_per_block_y, &cam, world, buffer_gen_time);
        delete world;
    } else {
        render(pixels_per_block_x, pixels_per_block_y, &cam, &world_list, buffer_gen_time);  // main.cu:113
    }
    std::cerr << "Buffer creation took " << buffer_gen_time << " seconds.\n";
    destroy_world(&world_list, &materials);
    return 0;
}
