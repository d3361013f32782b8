// crt_bvh.h -- GPU BVH builder (crt_bvh.hip), called from crt_api.cpp
#ifndef CRT_BVH_H
#define CRT_BVH_H

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/crt_hip.h"

// n spheres, depth = ceil(log2 n); d_boxes: 6 floats per sphere (xmin,xmax,ymin,ymax,zmin,zmax);
// d_axis: the 2n-1 pre-order split axes; d_nodes: 2^(depth+1)-1 nodes (heap order, out);
// d_pairs: 16 floats per heap pair, 2^depth records (out).  Synchronous on `stream`.
hipError_t crt_gpu_bvh_build(int n, int depth, const float* d_boxes, const uint8_t* d_axis, crt_bvh_node* d_nodes,
                             float* d_pairs, hipStream_t stream);

// the child-pair records (crt_device.h) of a complete node array with L leaves; asynchronous on `stream`
hipError_t crt_gpu_pairs_from_nodes(const crt_bvh_node* d_nodes, float* d_pairs, int L, hipStream_t stream);

#endif
