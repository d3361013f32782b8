// pin_thrust_ties.hip -- what does a real Thrust do with the reference's comparator on TIES?  
// bvh.h:30-45 compares primitives with `a.min <= b.min`, which is not a strict weak order; bvh.h:266 hands it to
// thrust::stable_sort(thrust::device, ...).  The tie behaviour is implementation-defined.  This program runs that
// comparator through rocThrust's device stable_sort (the ROCm image's Thrust; the reference's CUDA Thrust is not here)
// on tie-heavy inputs and prints the resulting index order beside the order under this repository's contract
// (strict `<`, stable: ties keep their current order).  TEST INFRASTRUCTURE; run on a GPU box:
//   hipcc --offload-arch=gfx950 -O1 -o pin_thrust_ties pin_thrust_ties.hip && ./pin_thrust_ties
#include <hip/hip_runtime.h>
#include <thrust/device_vector.h>
#include <thrust/execution_policy.h>
#include <thrust/sort.h>

#include <algorithm>
#include <cstdio>
#include <vector>

struct prim { size_t idx; float key; float pad[6]; };  // 32 bytes like bvh_primitive {size_t, aabb}

struct cmp_le {  // the reference's comparator shape, bvh.h:43
    __host__ __device__ bool operator()(const prim& a, const prim& b) const { return a.key <= b.key; }
};
struct cmp_lt {
    __host__ __device__ bool operator()(const prim& a, const prim& b) const { return a.key < b.key; }
};

static void run(const char* name, const std::vector<float>& keys) {
    std::vector<prim> h(keys.size());
    for (size_t i = 0; i < keys.size(); ++i) { h[i].idx = i; h[i].key = keys[i]; }
    thrust::device_vector<prim> d(h.begin(), h.end());
    thrust::stable_sort(thrust::device, d.begin(), d.end(), cmp_le());
    std::vector<prim> le(keys.size());
    thrust::copy(d.begin(), d.end(), le.begin());
    std::vector<prim> lt = h;
    std::stable_sort(lt.begin(), lt.end(), cmp_lt());
    bool same = true, sorted = true;
    for (size_t i = 0; i < keys.size(); ++i) same = same && le[i].idx == lt[i].idx;
    for (size_t i = 1; i < keys.size(); ++i) sorted = sorted && le[i - 1].key <= le[i].key;
    printf("%s: n=%zu keys-sorted=%d  same-order-as-contract(strict<,stable)=%d\n", name, keys.size(), (int)sorted, (int)same);
    if (keys.size() <= 16) {
        printf("   rocThrust device stable_sort with <= :");
        for (auto& p : le) printf(" %zu", p.idx);
        printf("\n   contract (strict <, stable)          :");
        for (auto& p : lt) printf(" %zu", p.idx);
        printf("\n");
    }
}

int main() {
    run("six keys 1,0,1,0,1,0 (SURVEY D3 probe)", {1, 0, 1, 0, 1, 0});
    run("config-1 y-min (helper.h:28-31: spheres 1,2,4 tie at -0.5)", {-200.5f, -0.5f, -0.5f, -0.4f, -0.5f});
    std::vector<float> many(4096);
    for (size_t i = 0; i < many.size(); ++i) many[i] = float((i * 2654435761u >> 7) % 13);
    run("4096 keys, 13 distinct values", many);
    std::vector<float> big(1 << 20);
    for (size_t i = 0; i < big.size(); ++i) big[i] = float((i * 2654435761u >> 5) % 1000);
    run("1M keys, 1000 distinct values", big);
    return 0;
}
