"""ctypes binding of oracle/liboracle.so -- the CPU restatement of the reference path.

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
import this module; the product never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")


class CameraParams(C.Structure):
    _fields_ = [("aspect_ratio", C.c_float), ("image_width", C.c_int32), ("samples_per_pixel", C.c_int32),
                ("vfov", C.c_float), ("lookfrom", C.c_float * 3), ("lookat", C.c_float * 3), ("vup", C.c_float * 3),
                ("defocus_angle", C.c_float), ("focus_dist", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("image_width", C.c_int32), ("image_height", C.c_int32), ("pixel_samples_scale", C.c_float),
                ("center", C.c_float * 3), ("pixel00_loc", C.c_float * 3), ("pixel_delta_u", C.c_float * 3),
                ("pixel_delta_v", C.c_float * 3), ("defocus_disk_u", C.c_float * 3), ("defocus_disk_v", C.c_float * 3),
                ("defocus_angle", C.c_float)]


class BvhNode(C.Structure):
    _fields_ = [("xmin", C.c_float), ("xmax", C.c_float), ("ymin", C.c_float), ("ymax", C.c_float),
                ("zmin", C.c_float), ("zmax", C.c_float), ("loc_for_leaf", C.c_int32), ("left_child_loc", C.c_int32),
                ("right_child_loc", C.c_int32)]


_lib = None
fp, ip, vp = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_void_p


def build():
    subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    L = C.CDLL(LIB_PATH)
    u32p = C.POINTER(C.c_uint32)
    L.orc_philox4x32_10.argtypes = [u32p, u32p, u32p]
    L.orc_u01.restype = C.c_float
    L.orc_u01.argtypes = [C.c_uint32]
    L.orc_sample_stream.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, fp]
    L.orc_seq_stream.argtypes = [C.c_uint64, C.c_int, fp]
    L.orc_axis_sequence.argtypes = [C.c_uint32, C.c_int, ip]
    L.orc_camera_initialize.argtypes = [C.POINTER(CameraParams), C.POINTER(Camera)]
    L.orc_scene_small.argtypes = [fp, fp, fp, ip, ip, fp, fp]
    L.orc_scene_random.argtypes = [C.c_uint64, C.c_int, C.c_float, fp, fp, fp, ip, ip, fp, fp]
    L.orc_bvh_num_nodes.argtypes = [C.c_int]
    L.orc_bvh_build.argtypes = [C.c_int, fp, fp, fp, C.c_uint32, C.POINTER(BvhNode), ip]
    scene_args = [C.c_int, fp, fp, fp, ip, C.c_int, ip, fp, fp]
    L.orc_render.argtypes = scene_args + [C.POINTER(BvhNode), C.c_int, C.POINTER(Camera), C.c_uint64,
                                          C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.POINTER(C.c_int64), C.POINTER(C.c_uint64), C.c_int]
    L.orc_resolve.argtypes = [C.POINTER(C.c_int64), C.c_size_t, C.c_int, fp]
    L.orc_trace_sample.argtypes = scene_args + [C.POINTER(BvhNode), C.c_int, C.POINTER(Camera), C.c_uint64,
                                                C.c_int, C.c_int, C.c_int, C.c_int, fp, fp, ip]
    L.orc_closest_hit.argtypes = [C.c_int, fp, fp, fp, C.POINTER(BvhNode), C.c_int, fp, fp, fp, fp]
    L.orc_write_ppm.restype = C.c_long
    L.orc_write_ppm.argtypes = [fp, C.c_int, C.c_int, C.c_char_p, C.c_long]
    _lib = L
    return L


def _fp(a):
    return a.ctypes.data_as(fp)


def _ip(a):
    return a.ctypes.data_as(ip)


class Scene:
    def __init__(self, center, motion, radius, mat_id, mat_type, mat_albedo, mat_param):
        self.center = np.ascontiguousarray(center, np.float32).reshape(-1, 3)
        self.motion = np.ascontiguousarray(motion, np.float32).reshape(-1, 3)
        self.radius = np.ascontiguousarray(radius, np.float32).reshape(-1)
        self.mat_id = np.ascontiguousarray(mat_id, np.int32).reshape(-1)
        self.mat_type = np.ascontiguousarray(mat_type, np.int32).reshape(-1)
        self.mat_albedo = np.ascontiguousarray(mat_albedo, np.float32).reshape(-1, 3)
        self.mat_param = np.ascontiguousarray(mat_param, np.float32).reshape(-1)

    @property
    def n(self):
        return int(self.radius.shape[0])

    @property
    def m(self):
        return int(self.mat_type.shape[0])

    def args(self):
        return (self.n, _fp(self.center), _fp(self.motion), _fp(self.radius), _ip(self.mat_id), self.m,
                _ip(self.mat_type), _fp(self.mat_albedo), _fp(self.mat_param))


def _alloc(cap):
    return (np.zeros((cap, 3), np.float32), np.zeros((cap, 3), np.float32), np.zeros(cap, np.float32),
            np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros((cap, 3), np.float32), np.zeros(cap, np.float32))


def scene_small():
    a = _alloc(5)
    n = lib().orc_scene_small(_fp(a[0]), _fp(a[1]), _fp(a[2]), _ip(a[3]), _ip(a[4]), _fp(a[5]), _fp(a[6]))
    return Scene(*[x[:n] for x in a])


def scene_random(seed=18, half=11, ground_radius=1000.0):
    a = _alloc(4 + 4 * half * half)
    n = lib().orc_scene_random(seed, half, ground_radius, _fp(a[0]), _fp(a[1]), _fp(a[2]), _ip(a[3]), _ip(a[4]),
                               _fp(a[5]), _fp(a[6]))
    return Scene(*[x[:n].copy() for x in a])


def camera(image_width, samples_per_pixel, aspect_ratio=16.0 / 9.0, vfov=20.0, lookfrom=(13, 2, 3), lookat=(0, 0, 0),
           vup=(0, 1, 0), defocus_angle=0.6, focus_dist=10.0):
    p = CameraParams()
    p.aspect_ratio = aspect_ratio
    p.image_width = image_width
    p.samples_per_pixel = samples_per_pixel
    p.vfov = vfov
    p.lookfrom[:] = lookfrom
    p.lookat[:] = lookat
    p.vup[:] = vup
    p.defocus_angle = defocus_angle
    p.focus_dist = focus_dist
    out = Camera()
    lib().orc_camera_initialize(C.byref(p), C.byref(out))
    return out


def bvh_build(scene, axis_seed=17):
    nn = lib().orc_bvh_num_nodes(scene.n)
    nodes = (BvhNode * nn)()
    order = np.zeros(scene.n, np.int32)
    lib().orc_bvh_build(scene.n, _fp(scene.center), _fp(scene.motion), _fp(scene.radius), axis_seed, nodes, _ip(order))
    return nodes, order


def nodes_to_array(nodes):
    """(N, 9) uint32 view of a node array (bit patterns), for bit-exact comparison."""
    return np.frombuffer(bytes(nodes), dtype=np.uint32).reshape(-1, 9).copy()


def render(scene, cam, seed, s0, ns, max_depth, nodes=None, region=None, nthreads=1):
    """Fixed-point (2^-40) un-normalised sums, int64 (H, W, 3) + stats [rays, pair_visits, sphere_tests, samples]."""
    W, H = cam.image_width, cam.image_height
    x0, y0, x1, y1 = region if region else (0, 0, W, H)
    sums = np.zeros((H, W, 3), np.int64)
    stats = (C.c_uint64 * 4)()
    nn = len(nodes) if nodes is not None else 0
    lib().orc_render(*scene.args(), nodes, nn, C.byref(cam), seed, x0, y0, x1, y1, s0, ns, max_depth,
                     sums.ctypes.data_as(C.POINTER(C.c_int64)), stats, nthreads)
    return sums, [int(v) for v in stats]


def resolve(sums, spp):
    """fp32 frame of fixed-point sums: float(double(sum) * (2^-40 / spp)) -- crt_resolve's formula."""
    sums = np.ascontiguousarray(sums, np.int64)
    fb = np.empty(sums.shape, np.float32)
    lib().orc_resolve(sums.ctypes.data_as(C.POINTER(C.c_int64)), sums.size, spp, _fp(fb))
    return fb


def trace_sample(scene, cam, seed, px, py, sample, max_depth, nodes=None):
    ray7 = np.zeros(7, np.float32)
    rgb = np.zeros(3, np.float32)
    seg = C.c_int(0)
    nn = len(nodes) if nodes is not None else 0
    lib().orc_trace_sample(*scene.args(), nodes, nn, C.byref(cam), seed, px, py, sample, max_depth, _fp(ray7), _fp(rgb),
                           C.byref(seg))
    return ray7, rgb, seg.value


def closest_hit(scene, ray7, nodes=None):
    t = C.c_float(0)
    ray7 = np.ascontiguousarray(ray7, np.float32)
    nn = len(nodes) if nodes is not None else 0
    obj = lib().orc_closest_hit(scene.n, _fp(scene.center), _fp(scene.motion), _fp(scene.radius), nodes, nn, _fp(ray7),
                                C.byref(t), None, None)
    return obj, t.value


def write_ppm(fb):
    fb = np.ascontiguousarray(fb, np.float32)
    h, w = fb.shape[0], fb.shape[1]
    need = lib().orc_write_ppm(_fp(fb), w, h, None, 0)
    buf = C.create_string_buffer(need)
    lib().orc_write_ppm(_fp(fb), w, h, buf, need)
    return buf.raw[:need]
