"""cuda_ray_tracer_amd -- MI355X-native implementation of the render hot path of
ChenVand/CUDA_Ray_Tracer behind a C ABI (include/crt_hip.h, libcrt_hip.so).

The Python here is only a thin convenience layer over the C ABI for tests and bench.py:
numpy arrays in, numpy arrays out.  The C++ host API that mirrors the reference's scene
classes lives in cuda_ray_tracer_amd/host/.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import (CRT_ARITH_EXACT, CRT_ARITH_FAST, CRT_DIELECTRIC, CRT_LAMBERTIAN, CRT_MATERIAL_BASE, CRT_METAL, CRT_SPLIT_SAMPLES,
                    CRT_SPLIT_TILES, CRT_WORLD_BVH, CRT_WORLD_LIST, BvhNode, Camera, CameraParams, RenderStats)

__all__ = ["lib", "Scene", "camera", "reference_camera", "scene_small", "scene_random", "bvh_build_host",
           "write_ppm", "Renderer", "Group", "build_info", "CRT_WORLD_LIST", "CRT_WORLD_BVH", "CRT_ARITH_EXACT",
           "CRT_ARITH_FAST", "CRT_SPLIT_SAMPLES", "CRT_SPLIT_TILES", "CrtError"]

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _capi.load()
    return _lib


class CrtError(RuntimeError):
    pass


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class Scene:
    """Flat scene arrays: what crt_upload_scene takes."""

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


def _alloc_scene(cap):
    return (np.zeros((cap, 3), np.float32), np.zeros((cap, 3), np.float32), np.zeros(cap, np.float32),
            np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros((cap, 3), np.float32), np.zeros(cap, np.float32))


def _check(rc, what, ctx=None):
    if rc != 0:
        msg = lib().crt_last_error(ctx)
        raise CrtError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def scene_small():
    """helper.h:7-35 (create_world)."""
    a = _alloc_scene(5)
    n = C.c_int32(0)
    _check(lib().crt_scene_small(_fp(a[0]), _fp(a[1]), _fp(a[2]), _ip(a[3]), _ip(a[4]), _fp(a[5]), _fp(a[6]), C.byref(n)),
           "crt_scene_small")
    return Scene(*[x[:n.value] for x in a])


def scene_random(seed=18, half=11, ground_radius=1000.0):
    """helper.h:37-102 (create_world2), generalised to a (2*half)^2 grid."""
    cap = 4 + 4 * half * half
    a = _alloc_scene(cap)
    n = C.c_int32(0)
    _check(lib().crt_scene_random(seed, half, ground_radius, cap, _fp(a[0]), _fp(a[1]), _fp(a[2]), _ip(a[3]), _ip(a[4]),
                                  _fp(a[5]), _fp(a[6]), C.byref(n)), "crt_scene_random")
    return Scene(*[x[:n.value].copy() for x in a])


def camera(image_width, samples_per_pixel, aspect_ratio=16.0 / 9.0, vfov=20.0, lookfrom=(13, 2, 3), lookat=(0, 0, 0),
           vup=(0, 1, 0), defocus_angle=0.6, focus_dist=10.0):
    """camera fields + initialize() (camera.h:11-22, 72-107); defaults = main_experimental.cu:103-114."""
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
    _check(lib().crt_camera_initialize(C.byref(p), C.byref(out)), "crt_camera_initialize")
    return out


reference_camera = camera


def bvh_build_host(scene, axis_seed=17):
    """bvh_world ctor on host arrays; returns a numpy structured view of crt_bvh_node[]."""
    nn = lib().crt_bvh_num_nodes(scene.n)
    nodes = (BvhNode * nn)()
    _check(lib().crt_bvh_build_host(scene.n, _fp(scene.center), _fp(scene.motion), _fp(scene.radius), axis_seed, nodes),
           "crt_bvh_build_host")
    return nodes


def write_ppm(fb):
    fb = np.ascontiguousarray(fb, np.float32)
    h, w = fb.shape[0], fb.shape[1]
    need = lib().crt_write_ppm(_fp(fb), w, h, None, 0)
    buf = C.create_string_buffer(need)
    lib().crt_write_ppm(_fp(fb), w, h, buf, need)
    return buf.raw[:need]


def build_info():
    """Hash of the kernel sources the loaded library was built from + the default arithmetic mode."""
    import os
    return {"build_id": lib().crt_build_id().decode(), "arith": "fast" if os.environ.get("CRT_ARITH") == "fast" else "exact"}


class Renderer:
    """One crt_context: device + seed + scene + BVH + camera."""

    def __init__(self, device=0, seed=1, _ctx=None):
        self._owned = _ctx is None
        self._ctx = C.c_void_p() if _ctx is None else C.c_void_p(_ctx)
        if _ctx is None:
            _check(lib().crt_create(device, seed, C.byref(self._ctx)), "crt_create")
        self.cam = None
        self.scene = None

    def close(self):
        if self._ctx and self._owned:
            lib().crt_destroy(self._ctx)
        self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_seed(self, seed):
        _check(lib().crt_set_seed(self._ctx, seed), "crt_set_seed", self._ctx)

    def upload_scene(self, scene):
        s = scene
        _check(lib().crt_upload_scene(self._ctx, s.n, _fp(s.center), _fp(s.motion), _fp(s.radius), _ip(s.mat_id), s.m,
                                      _ip(s.mat_type), _fp(s.mat_albedo), _fp(s.mat_param)), "crt_upload_scene", self._ctx)
        self.scene = scene

    def build_bvh(self, axis_seed=17):
        nn = C.c_int32(0)
        _check(lib().crt_build_bvh(self._ctx, axis_seed, C.byref(nn)), "crt_build_bvh", self._ctx)
        return nn.value

    def bvh_nodes(self):
        nn = lib().crt_bvh_num_nodes(self.scene.n)
        nodes = (BvhNode * nn)()
        _check(lib().crt_get_bvh_nodes(self._ctx, nodes, nn), "crt_get_bvh_nodes", self._ctx)
        return nodes

    def upload_bvh_nodes(self, nodes):
        _check(lib().crt_upload_bvh_nodes(self._ctx, nodes, len(nodes)), "crt_upload_bvh_nodes", self._ctx)

    def set_camera(self, cam):
        _check(lib().crt_set_camera(self._ctx, C.byref(cam)), "crt_set_camera", self._ctx)
        self.cam = cam

    def set_arith(self, arith):
        _check(lib().crt_set_arith(self._ctx, int(arith)), "crt_set_arith", self._ctx)

    def kernel_name(self):
        return lib().crt_last_kernel_name(self._ctx).decode()

    def selftest_exact_math(self):
        """{"rcp" | "sqrt" | "rsqrt" | "fixed": (arguments whose short sequence differs from the general one, smallest such
        argument's bits, arguments on the short sequence)} over all 2^32 arguments, on this context's GPU
        (crt_selftest_exact_math)."""
        import ctypes as C
        out = (C.c_uint64 * 12)()
        _check(lib().crt_selftest_exact_math(self._ctx, out), "crt_selftest_exact_math", self._ctx)
        return {name: (out[3 * i], out[3 * i + 1], out[3 * i + 2]) for i, name in enumerate(("rcp", "sqrt", "rsqrt", "fixed"))}

    def set_count_stats(self, enable):
        _check(lib().crt_set_count_stats(self._ctx, int(enable)), "crt_set_count_stats", self._ctx)

    def render(self, world, spp, sample_offset=0, max_depth=50):
        """crt_render: normalised framebuffer (H, W, 3) + stats."""
        fb = np.empty((self.cam.image_height, self.cam.image_width, 3), np.float32)
        st = RenderStats()
        _check(lib().crt_render(self._ctx, world, spp, sample_offset, max_depth, _fp(fb), C.byref(st)), "crt_render", self._ctx)
        return fb, st

    def render_accumulate(self, world, sample_offset, nsamples, max_depth, d_sums_ptr, stream=None, tile_stride=1,
                          tile_offset=0):
        if tile_stride == 1 and tile_offset == 0:
            _check(lib().crt_render_accumulate(self._ctx, world, sample_offset, nsamples, max_depth, C.c_void_p(d_sums_ptr),
                                               C.c_void_p(stream or 0)), "crt_render_accumulate", self._ctx)
        else:
            _check(lib().crt_render_accumulate_tiles(self._ctx, world, sample_offset, nsamples, max_depth, tile_stride,
                                                     tile_offset, C.c_void_p(d_sums_ptr), C.c_void_p(stream or 0)),
                   "crt_render_accumulate_tiles", self._ctx)

    def render_static(self, world, spp, sample_offset, max_depth, d_fb_ptr, stream=None):
        """crt_render_static: the frame of crt_render into the caller's device buffer, no allocation."""
        _check(lib().crt_render_static(self._ctx, world, spp, sample_offset, max_depth, C.c_void_p(d_fb_ptr),
                                       C.c_void_p(stream or 0)), "crt_render_static", self._ctx)

    def resolve(self, d_sums_ptr, spp, d_fb_ptr, stream=None):
        _check(lib().crt_resolve(self._ctx, C.c_void_p(d_sums_ptr), spp, C.c_void_p(d_fb_ptr), C.c_void_p(stream or 0)),
               "crt_resolve", self._ctx)

    def stats(self):
        st = RenderStats()
        _check(lib().crt_get_stats(self._ctx, C.byref(st)), "crt_get_stats", self._ctx)
        return st

    def kernel_times(self, count):
        out = (C.c_double * count)()
        _check(lib().crt_get_kernel_times(self._ctx, count, out), "crt_get_kernel_times", self._ctx)
        return list(out)

    def trace_samples(self, world, px, py, sample, max_depth=50):
        px = np.ascontiguousarray(px, np.int32)
        py = np.ascontiguousarray(py, np.int32)
        sample = np.ascontiguousarray(sample, np.int32)
        n = px.shape[0]
        ray7 = np.empty((n, 7), np.float32)
        rgb = np.empty((n, 3), np.float32)
        seg = np.empty(n, np.int32)
        _check(lib().crt_trace_samples(self._ctx, world, n, _ip(px), _ip(py), _ip(sample), max_depth, _fp(ray7), _fp(rgb),
                                       _ip(seg)), "crt_trace_samples", self._ctx)
        return ray7, rgb, seg


class Group:
    """crt_group: one context per entry of `devices`; one image cut over them (include/crt_hip.h)."""

    def __init__(self, devices, seed=1):
        self._grp = C.c_void_p()
        devs = np.ascontiguousarray(devices, np.int32)
        rc = lib().crt_group_create(_ip(devs), len(devs), seed, C.byref(self._grp))
        if rc != 0:
            raise CrtError(f"crt_group_create failed ({rc}): {lib().crt_group_last_error(None).decode()}")
        self.cam = None
        self.scene = None

    def _check(self, rc, what):
        if rc != 0:
            raise CrtError(f"{what} failed ({rc}): {lib().crt_group_last_error(self._grp).decode()}")

    def close(self):
        if self._grp:
            lib().crt_group_destroy(self._grp)
            self._grp = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return lib().crt_group_size(self._grp)

    def member(self, i):
        """Borrowed Renderer view of member i (knobs, statistics); the group keeps ownership."""
        ptr = lib().crt_group_member(self._grp, i)
        if not ptr:
            raise IndexError(i)
        r = Renderer(_ctx=ptr)
        r.cam, r.scene = self.cam, self.scene
        return r

    def upload_scene(self, scene):
        s = scene
        self._check(lib().crt_group_upload_scene(self._grp, s.n, _fp(s.center), _fp(s.motion), _fp(s.radius), _ip(s.mat_id), s.m,
                                                 _ip(s.mat_type), _fp(s.mat_albedo), _fp(s.mat_param)), "crt_group_upload_scene")
        self.scene = scene

    def build_bvh(self, axis_seed=17):
        nn = C.c_int32(0)
        self._check(lib().crt_group_build_bvh(self._grp, axis_seed, C.byref(nn)), "crt_group_build_bvh")
        return nn.value

    def upload_bvh_nodes(self, nodes):
        self._check(lib().crt_group_upload_bvh_nodes(self._grp, nodes, len(nodes)), "crt_group_upload_bvh_nodes")

    def set_camera(self, cam):
        self._check(lib().crt_group_set_camera(self._grp, C.byref(cam)), "crt_group_set_camera")
        self.cam = cam

    def render(self, world, spp, sample_offset=0, max_depth=50, split=CRT_SPLIT_SAMPLES):
        fb = np.empty((self.cam.image_height, self.cam.image_width, 3), np.float32)
        st = RenderStats()
        self._check(lib().crt_group_render(self._grp, world, spp, sample_offset, max_depth, split, _fp(fb), C.byref(st)),
                    "crt_group_render")
        return fb, st
