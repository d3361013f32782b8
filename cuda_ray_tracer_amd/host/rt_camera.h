// rt_camera.h -- the reference's `camera` (camera.h:9-107): same public fields, initialize(),
// getters and display_frame().  initialize() derives the viewport through crt_camera_initialize
// (bit-identical arithmetic); display_frame() writes the P3 PPM to std::cout through
// crt_write_ppm (color.h:17-35).
#ifndef CRT_HOST_RT_CAMERA_H
#define CRT_HOST_RT_CAMERA_H

#include <iostream>
#include <string>

#include "../../include/crt_hip.h"
#include "rt_math.h"

class camera {
  public:
    float aspect_ratio = 1.0;       // Ratio of image width over height
    int image_width = 100;          // Rendered image width in pixel count
    int samples_per_pixel = 32;     // Count of random samples for each pixel
    int max_depth = 50;             // bounce limit; commented out in camera.h:14, constant 50 in render_with_cuda.h:33

    float vfov = 90;
    point3 lookfrom = point3(0, 0, 0);
    point3 lookat = point3(0, 0, -1);
    vec3 vup = vec3(0, 1, 0);

    float defocus_angle = 0;
    float focus_dist = 10;

    void initialize() {
        crt_camera_params p;
        p.aspect_ratio = aspect_ratio; p.image_width = image_width; p.samples_per_pixel = samples_per_pixel;
        p.vfov = vfov; p.defocus_angle = defocus_angle; p.focus_dist = focus_dist;
        for (int k = 0; k < 3; ++k) { p.lookfrom[k] = lookfrom[k]; p.lookat[k] = lookat[k]; p.vup[k] = vup[k]; }
        if (crt_camera_initialize(&p, &derived) != CRT_OK) {
            std::cerr << "Fatal error: camera::initialize: invalid camera parameters\n*** FAILED - ABORTING\n";
            exit(1);
        }
    }

    int get_image_height() const { return derived.image_height; }
    float get_pixel_samples_scale() const { return derived.pixel_samples_scale; }
    point3 get_center() const { return v(derived.center); }
    point3 get_pixel00_loc() const { return v(derived.pixel00_loc); }
    point3 get_pixel_delta_u() const { return v(derived.pixel_delta_u); }
    point3 get_pixel_delta_v() const { return v(derived.pixel_delta_v); }
    point3 get_defocus_disk_u() const { return v(derived.defocus_disk_u); }
    point3 get_defocus_disk_v() const { return v(derived.defocus_disk_v); }
    const crt_camera& get_derived() const { return derived; }

    void display_frame(vec3* frame_buffer) {  // camera.h:44-57
        const float* fb = reinterpret_cast<const float*>(frame_buffer);
        const int64_t need = crt_write_ppm(fb, image_width, derived.image_height, nullptr, 0);
        std::string text(size_t(need), '\0');
        crt_write_ppm(fb, image_width, derived.image_height, &text[0], need);
        std::cout << text;
    }

  private:
    static point3 v(const float* p) { return point3(p[0], p[1], p[2]); }
    crt_camera derived{};
};

#endif
