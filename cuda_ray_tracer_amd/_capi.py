"""ctypes binding of libcrt_hip.so (include/crt_hip.h).

This is plumbing for tests and bench.py: the product is the C-ABI library.  Importing this
module loads the in-tree shared library and fails loudly when it is missing -- there is no
Python or CPU fallback for the hot path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CRT_LIB_PATH") or os.path.join(_HERE, "libcrt_hip.so")  # CRT_LIB_PATH: A/B of two builds (tools)

HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "crt_hip.h")


def header_exports(path=HEADER_PATH):
    """Every function include/crt_hip.h declares, read from its prototypes (so the export check of
    __graft_entry__.build() and tests/test_host_parity.py cannot drift from the header)."""
    import re
    text = re.sub(r"/\*.*?\*/", " ", open(path).read(), flags=re.S)   # comments out
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)                  # preprocessor lines out
    text = re.sub(r"\{[^{}]*\}", " ", text)                           # struct / enum bodies out (none nest)
    names = re.findall(r"\b(crt_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


# every symbol include/crt_hip.h declares
EXPORTS = header_exports()

CRT_LAMBERTIAN, CRT_METAL, CRT_DIELECTRIC, CRT_MATERIAL_BASE = 0, 1, 2, 3
CRT_WORLD_LIST, CRT_WORLD_BVH = 0, 1
CRT_ARITH_EXACT, CRT_ARITH_FAST = 0, 1
CRT_SPLIT_SAMPLES, CRT_SPLIT_TILES = 0, 1


class CameraParams(C.Structure):  # crt_camera_params, camera.h:11-22
    _fields_ = [("aspect_ratio", C.c_float), ("image_width", C.c_int32), ("samples_per_pixel", C.c_int32),
                ("vfov", C.c_float), ("lookfrom", C.c_float * 3), ("lookat", C.c_float * 3), ("vup", C.c_float * 3),
                ("defocus_angle", C.c_float), ("focus_dist", C.c_float)]


class Camera(C.Structure):  # crt_camera, camera.h:60-68
    _fields_ = [("image_width", C.c_int32), ("image_height", C.c_int32), ("pixel_samples_scale", C.c_float),
                ("center", C.c_float * 3), ("pixel00_loc", C.c_float * 3), ("pixel_delta_u", C.c_float * 3),
                ("pixel_delta_v", C.c_float * 3), ("defocus_disk_u", C.c_float * 3), ("defocus_disk_v", C.c_float * 3),
                ("defocus_angle", C.c_float)]


class BvhNode(C.Structure):  # crt_bvh_node, bvh.h:47-62
    _fields_ = [("xmin", C.c_float), ("xmax", C.c_float), ("ymin", C.c_float), ("ymax", C.c_float),
                ("zmin", C.c_float), ("zmax", C.c_float), ("loc_for_leaf", C.c_int32), ("left_child_loc", C.c_int32),
                ("right_child_loc", C.c_int32)]


class RenderStats(C.Structure):  # crt_render_stats
    _fields_ = [("kernel_seconds", C.c_double), ("wall_seconds", C.c_double), ("rays", C.c_uint64),
                ("samples", C.c_uint64), ("pair_visits", C.c_uint64), ("sphere_tests", C.c_uint64)]


def load(path=LIB_PATH):
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no fallback path.")
    # One HIP runtime per process: PyTorch bundles its own libamdhip64.so (SONAME libamdhip64.so.7,
    # the same as /opt/rocm's).  Two copies in one process each open the GPU and the second one
    # finds "no HIP GPUs".  Importing torch first makes libcrt_hip.so's DT_NEEDED libamdhip64.so.7
    # resolve to the copy torch already loaded, so device pointers, streams and RCCL buffers are
    # shared with torch.  Without torch the library binds to /opt/rocm's runtime (its RUNPATH).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    fp, ip, vp = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_void_p
    lib.crt_last_error.restype = C.c_char_p
    lib.crt_last_error.argtypes = [vp]
    lib.crt_camera_initialize.argtypes = [C.POINTER(CameraParams), C.POINTER(Camera)]
    lib.crt_scene_small.argtypes = [fp, fp, fp, ip, ip, fp, fp, ip]
    lib.crt_scene_random.argtypes = [C.c_uint64, C.c_int32, C.c_float, C.c_int32, fp, fp, fp, ip, ip, fp, fp, ip]
    lib.crt_write_ppm.restype = C.c_int64
    lib.crt_write_ppm.argtypes = [fp, C.c_int32, C.c_int32, C.c_char_p, C.c_int64]
    lib.crt_bvh_num_nodes.restype = C.c_int32
    lib.crt_bvh_num_nodes.argtypes = [C.c_int32]
    lib.crt_bvh_build_host.argtypes = [C.c_int32, fp, fp, fp, C.c_uint32, C.POINTER(BvhNode)]
    lib.crt_create.argtypes = [C.c_int32, C.c_uint64, C.POINTER(vp)]
    lib.crt_destroy.argtypes = [vp]
    lib.crt_set_seed.argtypes = [vp, C.c_uint64]
    lib.crt_upload_scene.argtypes = [vp, C.c_int32, fp, fp, fp, ip, C.c_int32, ip, fp, fp]
    lib.crt_build_bvh.argtypes = [vp, C.c_uint32, ip]
    lib.crt_get_bvh_nodes.argtypes = [vp, C.POINTER(BvhNode), C.c_int32]
    lib.crt_set_camera.argtypes = [vp, C.POINTER(Camera)]
    lib.crt_render.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, fp, C.POINTER(RenderStats)]
    lib.crt_render_accumulate.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, vp]
    lib.crt_get_stats.argtypes = [vp, C.POINTER(RenderStats)]
    lib.crt_get_kernel_times.argtypes = [vp, C.c_int32, C.POINTER(C.c_double)]
    lib.crt_resolve.argtypes = [vp, vp, C.c_int32, vp, vp]
    lib.crt_set_count_stats.argtypes = [vp, C.c_int32]
    lib.crt_trace_samples.argtypes = [vp, C.c_int32, C.c_int32, ip, ip, ip, C.c_int32, fp, fp, ip]
    lib.crt_upload_bvh_nodes.argtypes = [vp, C.POINTER(BvhNode), C.c_int32]
    lib.crt_render_accumulate_tiles.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, vp]
    lib.crt_render_static.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, vp]
    lib.crt_set_arith.argtypes = [vp, C.c_int32]
    lib.crt_last_kernel_name.restype = C.c_char_p
    lib.crt_last_kernel_name.argtypes = [vp]
    lib.crt_build_id.restype = C.c_char_p
    lib.crt_build_id.argtypes = []
    lib.crt_group_create.argtypes = [ip, C.c_int32, C.c_uint64, C.POINTER(vp)]
    lib.crt_group_destroy.argtypes = [vp]
    lib.crt_group_size.restype = C.c_int32
    lib.crt_group_size.argtypes = [vp]
    lib.crt_group_member.restype = vp
    lib.crt_group_member.argtypes = [vp, C.c_int32]
    lib.crt_group_last_error.restype = C.c_char_p
    lib.crt_group_last_error.argtypes = [vp]
    lib.crt_group_upload_scene.argtypes = [vp, C.c_int32, fp, fp, fp, ip, C.c_int32, ip, fp, fp]
    lib.crt_group_build_bvh.argtypes = [vp, C.c_uint32, ip]
    lib.crt_group_upload_bvh_nodes.argtypes = [vp, C.POINTER(BvhNode), C.c_int32]
    lib.crt_group_set_camera.argtypes = [vp, C.POINTER(Camera)]
    lib.crt_group_render.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, fp, C.POINTER(RenderStats)]
    try:  # absent from older builds (CRT_LIB_PATH A/Bs)
        lib.crt_selftest_exact_math.argtypes = [vp, C.POINTER(C.c_uint64)]
    except AttributeError:
        pass
    lib.crt_debug_counters.argtypes = [vp, C.POINTER(C.c_uint64)]  # internal, not in the header
    try:  # absent from older builds (CRT_LIB_PATH A/Bs)
        lib.crt_debug_timeline.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int32]  # internal, not in the header
    except AttributeError:
        pass
    return lib
