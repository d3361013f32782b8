// rt_scene.h -- host-side scene classes with the reference's names and constructors:
//   material / lambertian / metal / dielectric / material_list   (material.h:7-115)
//   hit_record / hittable / sphere / hittable_list               (hittable.h, sphere.h, hittable_list.h)
//   bvh_node / bvh_world                                          (bvh.h:47-62, :126-192)
// In the reference these objects live in device memory and are called through device-side
// virtuals.  Here they are plain host objects that FLATTEN into the SoA arrays of
// crt_upload_scene; intersection and scattering run in libcrt_hip.so.
#ifndef CRT_HOST_RT_SCENE_H
#define CRT_HOST_RT_SCENE_H

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "../../include/crt_hip.h"
#include "rt_math.h"

class material;

// hittable.h:8-23 -- kept for source compatibility; filled by the GPU, not by host code
class hit_record {
  public:
    point3 p;
    vec3 normal;
    material* mat_ptr = nullptr;
    float t = 0;
    bool front_face = false;
};

// what a material flattens to (crt_upload_scene: type, albedo, param)
struct crt_flat_material {
    int32_t type;
    color albedo;
    float param;
};

class material {  // material.h:7-20
  public:
    virtual ~material() = default;
    // a material of the base class never scatters (material.h:11-19): the path returns black at once
    virtual crt_flat_material flatten() const { return {CRT_MATERIAL_BASE, color(0, 0, 0), 0.0f}; }
};

class lambertian : public material {  // material.h:22-42
  public:
    lambertian(const color& albedo) : albedo(albedo) {}
    crt_flat_material flatten() const override { return {CRT_LAMBERTIAN, albedo, 0.0f}; }

  private:
    color albedo;
};

class metal : public material {  // material.h:44-60
  public:
    metal(const color& albedo, float fuzz) : albedo(albedo), fuzz(fuzz < 1 ? fuzz : 1) {}
    crt_flat_material flatten() const override { return {CRT_METAL, albedo, fuzz}; }

  private:
    color albedo;
    float fuzz;
};

class dielectric : public material {  // material.h:62-98
  public:
    dielectric(float refraction_index) : refraction_index(refraction_index) {}
    crt_flat_material flatten() const override { return {CRT_DIELECTRIC, color(1, 1, 1), refraction_index}; }

  private:
    float refraction_index;
};

class material_list {  // material.h:100-115: owner of the materials
  public:
    material** materials;
    int size;
    material_list(material** material_list_list, int length) : materials(material_list_list), size(length) {}
    ~material_list() {
        for (int i = 0; i < size; ++i) delete materials[i];
        delete[] materials;
    }
};

// flattened scene under construction
struct crt_flat_scene {
    std::vector<float> center, motion, radius, mat_albedo, mat_param;
    std::vector<int32_t> mat_id, mat_type;
    std::unordered_map<const material*, int32_t> mat_ptrs;  // one flat material per distinct pointer
    int32_t material_index(const material* m) {
        auto it = mat_ptrs.find(m);
        if (it != mat_ptrs.end()) return it->second;
        const crt_flat_material f = m->flatten();
        const int32_t id = int32_t(mat_type.size());
        mat_ptrs.emplace(m, id);
        mat_type.push_back(f.type);
        for (int k = 0; k < 3; ++k) mat_albedo.push_back(f.albedo[k]);
        mat_param.push_back(f.param);
        return id;
    }
    int32_t num_objects() const { return int32_t(radius.size()); }
};

class hittable {  // hittable.h:43-51
  public:
    virtual ~hittable() = default;
    virtual aabb bounding_box() const = 0;
    virtual void flatten(crt_flat_scene& out) const = 0;  // replaces the device-side virtual hit()
};

class sphere : public hittable {  // sphere.h:6-42
  public:
    sphere(const point3& static_center, float radius, material* new_mat)
        : center(static_center, vec3(0, 0, 0)), radius(fmaxf(0, radius)), mat_ptr(new_mat) {
        const vec3 rvec(radius, radius, radius);
        bbox = aabb(static_center - rvec, static_center + rvec);
    }
    sphere(const point3& center1, const point3& center2, float radius, material* new_mat)
        : center(center1, center2 - center1), radius(fmaxf(0, radius)), mat_ptr(new_mat) {
        const vec3 rvec(radius, radius, radius);
        const aabb box1(center.at(0) - rvec, center.at(0) + rvec);
        const aabb box2(center.at(1) - rvec, center.at(1) + rvec);
        bbox = aabb(box1, box2);
    }
    // unlike sphere.h:28-29 the copy keeps its bounding box
    sphere(const sphere& other) = default;

    aabb bounding_box() const override { return bbox; }
    void flatten(crt_flat_scene& out) const override {
        for (int k = 0; k < 3; ++k) { out.center.push_back(center.orig[k]); out.motion.push_back(center.dir[k]); }
        out.radius.push_back(radius);
        out.mat_id.push_back(out.material_index(mat_ptr));
    }

    ray center;
    float radius;
    material* mat_ptr;
    aabb bbox;
};

class hittable_list : public hittable {  // hittable_list.h:7-41
  public:
    hittable** objects;
    int size;
    hittable_list(hittable** object_list, int length) : objects(object_list), size(length) {
        aabb box;
        if (length > 0) {
            box = object_list[0]->bounding_box();
            for (int i = 1; i < length; ++i) box = aabb(box, object_list[i]->bounding_box());
        }
        bbox = box;
    }
    aabb bounding_box() const override { return bbox; }
    void flatten(crt_flat_scene& out) const override {
        for (int i = 0; i < size; ++i) objects[i]->flatten(out);
    }

  private:
    aabb bbox;
};

// bvh.h:47-62 -- same fields, same 36-byte layout
class bvh_node {
  public:
    aabb bbox;
    int loc_for_leaf = -1;
    int left_child_loc = -1;
    int right_child_loc = -1;
};
static_assert(sizeof(bvh_node) == sizeof(crt_bvh_node), "bvh_node must match crt_bvh_node");

// Process-wide settings of the host mirror.  The reference seeds cuRAND with time(0) (helper.h:134) and uses the
// current CUDA device; here the seed is crt_host_seed() (default 1, or the environment variable CRT_SEED) so that
// runs are reproducible, and the GPUs are crt_host_devices(): CRT_DEVICES="0,1,2,3" (one image split over them), else
// the single device CRT_DEVICE (default 0).
inline uint64_t& crt_host_seed() {
    static uint64_t seed = getenv("CRT_SEED") ? strtoull(getenv("CRT_SEED"), nullptr, 10) : 1ull;
    return seed;
}
inline std::vector<int32_t>& crt_host_devices() {
    static std::vector<int32_t> devices = [] {
        std::vector<int32_t> d;
        if (const char* list = getenv("CRT_DEVICES")) {
            for (const char* p = list; *p;) {
                char* end = nullptr;
                const long v = strtol(p, &end, 10);
                if (end == p) break;
                d.push_back(int32_t(v));
                p = (*end == ',') ? end + 1 : end;
            }
        }
        if (d.empty()) d.push_back(getenv("CRT_DEVICE") ? atoi(getenv("CRT_DEVICE")) : 0);
        return d;
    }();
    return devices;
}

// errors behave like cudaCheckErrors (main_experimental.cu:13-23): message on stderr, exit(1)
#define crtCheckErrors(msg_source, rc, msg)                                                                  \
    do {                                                                                                     \
        if ((rc) != CRT_OK) {                                                                                \
            fprintf(stderr, "Fatal error: %s (%s at %s:%d)\n", msg, (msg_source), __FILE__, __LINE__);        \
            fprintf(stderr, "*** FAILED - ABORTING\n");                                                      \
            exit(1);                                                                                         \
        }                                                                                                    \
    } while (0)

// bvh.h:126-192.  Like the reference's, the constructor builds the tree ON THE DEVICE: it flattens the objects,
// hands them to the GPUs of crt_host_devices() (crt_group), builds the node array there (crt_bvh.hip: one stable
// sort per tree level, bit-identical to the reference's recursion under the documented tie rule) and reads it back
// into the public d_nodes (host memory here).  The group keeps scene and tree, so render_experimental() renders
// without building anything again.  A process that has no GPU at all (host-side tools, the CPU test-suite) gets the
// same node array from the library's host builder and no group; rendering then fails like any launch without a device.
class bvh_world {
  public:
    int num_objects;
    hittable** d_objects;
    int num_nodes;
    bvh_node* d_nodes;
    crt_flat_scene flat;
    crt_group* group = nullptr;

    bvh_world(hittable** object_list, int list_length)
        : num_objects(list_length), d_objects(object_list), num_nodes(0), d_nodes(nullptr) {
        for (int i = 0; i < list_length; ++i) object_list[i]->flatten(flat);
        num_nodes = crt_bvh_num_nodes(flat.num_objects());
        if (num_nodes <= 0) return;
        d_nodes = new bvh_node[num_nodes];
        const std::vector<int32_t>& devices = crt_host_devices();
        int rc = crt_group_create(devices.data(), int32_t(devices.size()), crt_host_seed(), &group);
        if (rc == CRT_ERR_NO_DEVICE) {
            group = nullptr;
            crt_bvh_build_host(flat.num_objects(), flat.center.data(), flat.motion.data(), flat.radius.data(), 17u,
                               reinterpret_cast<crt_bvh_node*>(d_nodes));  // seed 17: bvh.h:144
            return;
        }
        crtCheckErrors(crt_group_last_error(nullptr), rc, "device context creation failed");
        rc = crt_group_upload_scene(group, flat.num_objects(), flat.center.data(), flat.motion.data(), flat.radius.data(),
                                    flat.mat_id.data(), int32_t(flat.mat_type.size()), flat.mat_type.data(),
                                    flat.mat_albedo.data(), flat.mat_param.data());
        crtCheckErrors(crt_group_last_error(group), rc, "scene upload failed");
        int32_t built = 0;
        rc = crt_group_build_bvh(group, 17u, &built);
        crtCheckErrors(crt_group_last_error(group), rc, "BVH build failed");
        rc = crt_get_bvh_nodes(crt_group_member(group, 0), reinterpret_cast<crt_bvh_node*>(d_nodes), num_nodes);
        crtCheckErrors(crt_last_error(crt_group_member(group, 0)), rc, "BVH read-back failed");
    }
    ~bvh_world() {  // bvh.h:194
        delete[] d_nodes;
        crt_group_destroy(group);
    }
    bvh_world(const bvh_world&) = delete;
    bvh_world& operator=(const bvh_world&) = delete;
};

#endif
