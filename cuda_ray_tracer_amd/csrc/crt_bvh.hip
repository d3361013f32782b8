// crt_bvh.hip -- the reference's BVH construction (bvh.h:135-192, 251-274, 64-100) on the GPU,
// producing the node array of the host builder (crt_host.cpp bvh_build) bit for bit.
//
// The reference recursion sorts the span of every internal call with a stable sort on the lower
// box bound of a randomly drawn axis, then splits it at floor(span/2).  All calls of one recursion
// depth work on disjoint, contiguous spans, so a whole depth is ONE stable sort of all n entries on
// the composite key (span start, sortable bits of the float key): 52 key bits, rocPRIM radix sort
// (LSD, stable), ceil(log2 n) sorts in all.  Which span a position belongs to at a given depth, and
// which of the 2n-1 pre-order axis draws that call made, follow from n alone: a call on a span of s
// makes 2s-1 draws, so its children draw at d+1 and d+2*floor(s/2) ("BVH build cost").
// The axis table itself (minstd_rand is sequential) is generated on the host and uploaded.
//
// Ties: the reference's comparator is a strict '<' on floats under a stable sort -- equal keys keep
// their current order.  The radix key maps -0.0f to +0.0f first, so "equal as floats" and "equal as
// keys" coincide; box bounds are never NaN.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "crt_bvh.h"
#include "crt_device.h"

namespace {

struct span_t { uint32_t start, size, draw; };

// the call of recursion depth `level` whose span contains position p
__device__ __forceinline__ span_t span_of(uint32_t p, uint32_t n, int level) {
    span_t s; s.start = 0; s.size = n; s.draw = 0;
    for (int j = 0; j < level && s.size >= 2; ++j) {
        const uint32_t left = s.size >> 1;
        if (p < s.start + left) { s.size = left; s.draw += 1; }
        else { s.start += left; s.size -= left; s.draw += 2 * left; }
    }
    return s;
}

__global__ __launch_bounds__(256) void iota_kernel(uint32_t* order, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) order[i] = i;
}

__global__ __launch_bounds__(256) void keys_kernel(uint64_t* keys, const uint32_t* order, const float* boxes,
                                                   const uint8_t* axis, uint32_t n, int level) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n) return;
    const span_t s = span_of(p, n, level);
    uint32_t low = 0;
    if (s.size >= 2) {
        const float f = boxes[6 * (size_t)order[p] + 2 * axis[s.draw]] + 0.0f;  // -0 -> +0
        const uint32_t u = __float_as_uint(f);
        low = (u & 0x80000000u) ? ~u : (u | 0x80000000u);                      // order-preserving bits
    }
    keys[p] = ((uint64_t)s.start << 32) | low;
}

// complete tree in heap order, root at 0 (bvh.h:64-100): leaves, then one level at a time upwards
__global__ __launch_bounds__(256) void leaves_kernel(crt_bvh_node* nodes, const uint32_t* order, const float* boxes,
                                                     uint32_t n, uint32_t L) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= L) return;
    crt_bvh_node nd;
    if (i < n) {
        const uint32_t o = order[i];
        const float* b = boxes + 6 * (size_t)o;
        nd.xmin = b[0]; nd.xmax = b[1]; nd.ymin = b[2]; nd.ymax = b[3]; nd.zmin = b[4]; nd.zmax = b[5];
        nd.loc_for_leaf = (int32_t)o;
    } else {
        const float inf = __builtin_huge_valf();
        nd.xmin = inf; nd.xmax = -inf; nd.ymin = inf; nd.ymax = -inf; nd.zmin = inf; nd.zmax = -inf;
        nd.loc_for_leaf = -1;
    }
    nd.left_child_loc = -1; nd.right_child_loc = -1;
    nodes[(size_t)L - 1 + i] = nd;
}

__global__ __launch_bounds__(256) void level_kernel(crt_bvh_node* nodes, uint32_t first, uint32_t count) {
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= count) return;
    const size_t i = (size_t)first + j;
    const crt_bvh_node l = nodes[2 * i + 1], r = nodes[2 * i + 2];
    crt_bvh_node nd;  // interval(a, b), interval.h:12-16
    nd.xmin = l.xmin <= r.xmin ? l.xmin : r.xmin; nd.xmax = l.xmax >= r.xmax ? l.xmax : r.xmax;
    nd.ymin = l.ymin <= r.ymin ? l.ymin : r.ymin; nd.ymax = l.ymax >= r.ymax ? l.ymax : r.ymax;
    nd.zmin = l.zmin <= r.zmin ? l.zmin : r.zmin; nd.zmax = l.zmax >= r.zmax ? l.zmax : r.zmax;
    nd.loc_for_leaf = -1; nd.left_child_loc = (int32_t)(2 * i + 1); nd.right_child_loc = (int32_t)(2 * i + 2);
    nodes[i] = nd;
}

__device__ __forceinline__ int tag_of(const crt_bvh_node& nd) {
    if (nd.xmin > nd.xmax) return CRT_TAG_EMPTY;
    return nd.loc_for_leaf >= 0 ? nd.loc_for_leaf : CRT_TAG_INTERNAL;
}

// child-pair records (crt_device.h): record k (1-based heap) holds nodes 2k-1, 2k (0-based)
__global__ __launch_bounds__(256) void pairs_kernel(float4* pairs, const crt_bvh_node* nodes, uint32_t L) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= L) return;
    float4 q0 = make_float4(0, 0, 0, 0), q1 = q0, q2 = q0, q3 = q0;
    if (k == 0) {
        if (L == 1) { const int t = tag_of(nodes[0]); q3.x = __int_as_float(t); q3.y = __int_as_float(t); }
    } else {
        const crt_bvh_node l = nodes[2 * (size_t)k - 1], r = nodes[2 * (size_t)k];
        const int tl = tag_of(l), tr = tag_of(r);
        const float BIG = 3.0e38f;  // empty child: degenerate far box (crt_api.cpp)
        const bool el = tl == CRT_TAG_EMPTY, er = tr == CRT_TAG_EMPTY;
        q0 = make_float4(el ? BIG : l.xmin, el ? BIG : l.xmax, er ? BIG : r.xmin, er ? BIG : r.xmax);
        q1 = make_float4(el ? BIG : l.ymin, el ? BIG : l.ymax, er ? BIG : r.ymin, er ? BIG : r.ymax);
        q2 = make_float4(el ? BIG : l.zmin, el ? BIG : l.zmax, er ? BIG : r.zmin, er ? BIG : r.zmax);
        q3.x = __int_as_float(tl); q3.y = __int_as_float(tr);
    }
    pairs[4 * (size_t)k] = q0; pairs[4 * (size_t)k + 1] = q1; pairs[4 * (size_t)k + 2] = q2; pairs[4 * (size_t)k + 3] = q3;
}

inline unsigned blocks(uint64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

hipError_t crt_gpu_pairs_from_nodes(const crt_bvh_node* d_nodes, float* d_pairs, int L, hipStream_t stream) {
    hipLaunchKernelGGL(pairs_kernel, dim3(blocks((uint64_t)L)), dim3(256), 0, stream, (float4*)d_pairs, d_nodes, (uint32_t)L);
    return hipGetLastError();
}

#define BVH_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return e_; } } while (0)

hipError_t crt_gpu_bvh_build(int n_, int depth, const float* d_boxes, const uint8_t* d_axis, crt_bvh_node* d_nodes,
                             float* d_pairs, hipStream_t stream) {
    const uint32_t n = (uint32_t)n_, L = 1u << depth;
    uint32_t *order[2] = {nullptr, nullptr};
    uint64_t *keys[2] = {nullptr, nullptr};
    void* temp = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(order[0]); (void)hipFree(order[1]); (void)hipFree(keys[0]); (void)hipFree(keys[1]); (void)hipFree(temp);
    };
    for (int i = 0; i < 2; ++i) {
        BVH_TRY(hipMalloc((void**)&order[i], sizeof(uint32_t) * (size_t)n));
        BVH_TRY(hipMalloc((void**)&keys[i], sizeof(uint64_t) * (size_t)n));
    }
    int seg_bits = 0;
    while ((1ull << seg_bits) < (uint64_t)n) ++seg_bits;  // span starts are < n
    const unsigned end_bit = 32u + (unsigned)seg_bits;
    size_t temp_bytes = 0;
    BVH_TRY(rocprim::radix_sort_pairs(nullptr, temp_bytes, keys[0], keys[1], order[0], order[1], (size_t)n, 0u, end_bit, stream));
    BVH_TRY(hipMalloc(&temp, temp_bytes ? temp_bytes : 16));
    hipLaunchKernelGGL(iota_kernel, dim3(blocks(n)), dim3(256), 0, stream, order[0], n);
    int cur = 0;
    for (int level = 0; level < depth; ++level) {
        hipLaunchKernelGGL(keys_kernel, dim3(blocks(n)), dim3(256), 0, stream, keys[0], order[cur], d_boxes, d_axis, n, level);
        BVH_TRY(hipGetLastError());
        BVH_TRY(rocprim::radix_sort_pairs(temp, temp_bytes, keys[0], keys[1], order[cur], order[cur ^ 1], (size_t)n, 0u, end_bit, stream));
        cur ^= 1;
    }
    hipLaunchKernelGGL(leaves_kernel, dim3(blocks(L)), dim3(256), 0, stream, d_nodes, order[cur], d_boxes, n, L);
    for (int level = depth - 1; level >= 0; --level) {
        const uint32_t first = (1u << level) - 1u, count = 1u << level;
        hipLaunchKernelGGL(level_kernel, dim3(blocks(count)), dim3(256), 0, stream, d_nodes, first, count);
    }
    hipLaunchKernelGGL(pairs_kernel, dim3(blocks(L)), dim3(256), 0, stream, (float4*)d_pairs, d_nodes, L);
    BVH_TRY(hipGetLastError());
    BVH_TRY(hipStreamSynchronize(stream));
    cleanup();
    return hipSuccess;
}
