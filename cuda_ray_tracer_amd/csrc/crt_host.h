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
int bvh_build(int n, const float* center, const float* motion, const float* radius, uint32_t axis_seed,
              crt_bvh_node* nodes);
int64_t write_ppm(const float* fb, int width, int height, char* buf, int64_t cap);
}  // namespace crt

#endif
