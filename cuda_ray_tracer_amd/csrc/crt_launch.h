// crt_launch.h -- launchers defined in crt_kernels.hip, called from crt_api.cpp
#ifndef CRT_LAUNCH_H
#define CRT_LAUNCH_H

#include <hip/hip_runtime_api.h>

#include "crt_device.h"

hipError_t crt_launch_render(const crt_render_args& a, int world, bool in_lds, bool count, int block, int grid,
                             size_t lds_bytes, hipStream_t stream);
// the same kernels compiled with fp contraction and the fma slab test (crt_kernels.hip with CRT_FAST=1)
hipError_t crt_launch_render_fast(const crt_render_args& a, int world, bool in_lds, bool count, int block, int grid,
                                  size_t lds_bytes, hipStream_t stream);
hipError_t crt_launch_trace_samples_fast(const crt_render_args& a, int world, int count, const int* px, const int* py,
                                         const int* samples, float* ray7, float* rgb, int* segments, hipStream_t stream);
// dst[i] += sum over the listed frames (int64 fixed-point sums of a group's other members)
#define CRT_GROUP_MAX 16
struct crt_frame_list { const unsigned long long* p[CRT_GROUP_MAX]; int n; };
hipError_t crt_launch_sum_frames(unsigned long long* dst, const crt_frame_list& src, size_t count, hipStream_t stream);
// crt_static.hip: the per-GPU static scratch of crt_render_static (counters first, then the frame's int64 sums)
#define CRT_STATIC_COUNTER_WORDS CRT_COUNTER_WORDS   // = the size of a context's counter buffer; 198 912 B, a multiple of 256
#define CRT_STATIC_MAX_PIXELS_ (1u << 21)          // == CRT_STATIC_MAX_PIXELS of include/crt_hip.h
hipError_t crt_static_scratch(int device, unsigned long long** out);
hipError_t crt_launch_resolve(const void* sums, float* fb, size_t count, int spp, hipStream_t stream);
// out9 (device; {0, ~0, 0} x 3 on entry): for rcp_, sqrt_, rsqrt_ {arguments that differ from the general sequence, smallest such, arguments on the short one}
hipError_t crt_launch_math_selftest(unsigned long long* out12, hipStream_t stream);
hipError_t crt_launch_trace_samples(const crt_render_args& a, int world, int count, const int* px, const int* py,
                                    const int* samples, float* ray7, float* rgb, int* segments, hipStream_t stream);

#endif
