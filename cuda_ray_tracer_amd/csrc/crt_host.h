// crt_host.h -- host-only helpers of libcrt_hip.so (crt_host.cpp), used by crt_api.cpp
#ifndef CRT_HOST_H
#define CRT_HOST_H

#include <cstdint>

#include "../../include/crt_hip.h"

namespace crt {
void camera_initialize(const crt_camera_params& in, crt_camera& out);
int scene_small(float* center, float* motion, float* radius, int32_t* mat_id, int32_t* mat_type, float* albedo,
                float* param);
int scene_random(uint64_t seed, int half, float ground_radius, float* center, float* motion, float* radius,
                 int32_t* mat_id, int32_t* mat_type, float* albedo, float* param);
int bvh_depth(int n);
// the two inputs of the BVH construction that are computed on the host for both builders:
// sphere boxes (6 floats each, sphere.h:10-25) and the 2n-1 pre-order split axes (bvh.h:144-147,257)
void sphere_boxes(int n, const float* center, const float* motion, const float* radius, float* boxes);
void axis_table(int n, uint32_t axis_seed, uint8_t* axis);
int bvh_build(int n, const float* center, const float* motion, const float* radius, uint32_t axis_seed,
              crt_bvh_node* nodes);
int64_t write_ppm(const float* fb, int width, int height, char* buf, int64_t cap);
}  // namespace crt

#endif
