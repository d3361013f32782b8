"""CPU tests of the product's host-side logic (libcrt_hip.so, no GPU needed) against the oracle:
camera derivation, scene definitions, BVH node array (bit-exact), PPM writer; plus the C-ABI
surface itself (every declared symbol exported, error behaviour without a device).
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bits(struct):
    return bytes(struct)


def test_library_exports_every_declared_symbol(crt):
    from cuda_ray_tracer_amd import _capi
    header = open(os.path.join(ROOT, "include", "crt_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(crt_[a-z_0-9]+)\s*\(", header)))
    assert declared == sorted(_capi.EXPORTS)
    lib = crt.lib()
    for name in declared:
        assert hasattr(lib, name), name


def test_struct_layouts_match_header(crt):
    assert C.sizeof(crt.BvhNode) == 36
    assert C.sizeof(crt.CameraParams) == 4 * 15
    assert C.sizeof(crt.Camera) == 4 * 22
    assert C.sizeof(crt.RenderStats) == 48


@pytest.mark.parametrize("width,spp", [(400, 10), (1200, 500), (1920, 64), (3840, 1024), (37, 3), (1, 1)])
def test_camera_bit_exact(crt, orc, width, spp):
    a = crt.camera(width, spp)
    b = orc.camera(width, spp)
    assert _bits(a) == _bits(b)
    a = crt.camera(width, spp, aspect_ratio=1.5, vfov=63.0, lookfrom=(-2, 2, 1), lookat=(0, 0, -1), vup=(0.1, 1, 0),
                   defocus_angle=0.0, focus_dist=3.4)
    b = orc.camera(width, spp, aspect_ratio=1.5, vfov=63.0, lookfrom=(-2, 2, 1), lookat=(0, 0, -1), vup=(0.1, 1, 0),
                   defocus_angle=0.0, focus_dist=3.4)
    assert _bits(a) == _bits(b)


def _same_scene(a, b):
    for f in ("center", "motion", "radius", "mat_albedo", "mat_param"):
        x, y = getattr(a, f), getattr(b, f)
        assert x.shape == y.shape and np.array_equal(x.view(np.uint32), y.view(np.uint32)), f
    assert np.array_equal(a.mat_id, b.mat_id) and np.array_equal(a.mat_type, b.mat_type)


def test_scene_small_bit_exact(crt, orc):
    _same_scene(crt.scene_small(), orc.scene_small())


@pytest.mark.parametrize("seed,half,ground", [(18, 11, 1000.0), (1, 11, 1000.0), (2 ** 40 + 5, 3, 1000.0), (7, 40, 1e5), (9, 0, 10.0)])
def test_scene_random_bit_exact(crt, orc, seed, half, ground):
    _same_scene(crt.scene_random(seed, half, ground), orc.scene_random(seed, half, ground))


def test_scene_random_capacity_error(crt):
    a = [np.zeros((4, 3), np.float32), np.zeros((4, 3), np.float32), np.zeros(4, np.float32), np.zeros(4, np.int32),
         np.zeros(4, np.int32), np.zeros((4, 3), np.float32), np.zeros(4, np.float32)]
    n = C.c_int32(0)
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
    rc = crt.lib().crt_scene_random(18, 11, 1000.0, 4, a[0].ctypes.data_as(fp), a[1].ctypes.data_as(fp),
                                    a[2].ctypes.data_as(fp), a[3].ctypes.data_as(ip), a[4].ctypes.data_as(ip),
                                    a[5].ctypes.data_as(fp), a[6].ctypes.data_as(fp), C.byref(n))
    assert rc == 5  # CRT_ERR_TOO_LARGE


def _bvh_case(crt, orc, sc_c, sc_o, axis_seed=17):
    got = crt.bvh_build_host(sc_c, axis_seed)
    want, _ = orc.bvh_build(sc_o, axis_seed)
    assert len(got) == len(want) == crt.lib().crt_bvh_num_nodes(sc_c.n)
    g = np.frombuffer(bytes(got), np.uint32).reshape(-1, 9)
    w = np.frombuffer(bytes(want), np.uint32).reshape(-1, 9)
    assert np.array_equal(g, w)


def test_bvh_bit_exact_reference_scenes(crt, orc):
    _bvh_case(crt, orc, crt.scene_small(), orc.scene_small())           # 15 nodes
    _bvh_case(crt, orc, crt.scene_random(18, 11), orc.scene_random(18, 11))  # 1023 nodes
    _bvh_case(crt, orc, crt.scene_random(18, 11), orc.scene_random(18, 11), axis_seed=12345)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 31, 33, 100, 257])
def test_bvh_bit_exact_ragged_sizes(crt, orc, n):
    """n = 2^k, 2^k +- 1 (worst padding), single sphere; many equal keys (ties)."""
    rng = np.random.default_rng(n)
    center = rng.integers(-3, 4, (n, 3)).astype(np.float32)  # coarse grid -> lots of ties on every axis
    motion = np.where(rng.random((n, 1)) < 0.3, rng.normal(0, 1, (n, 3)), 0).astype(np.float32)
    radius = rng.choice([0.25, 0.5, 1.0], n).astype(np.float32)
    ids = np.arange(n, dtype=np.int32)
    args = (center, motion, radius, ids, np.zeros(n, np.int32), np.zeros((n, 3), np.float32), np.zeros(n, np.float32))
    _bvh_case(crt, orc, crt.Scene(*args), orc.Scene(*args))


def test_bvh_bit_exact_large(crt, orc):
    sc_c, sc_o = crt.scene_random(3, 60), orc.scene_random(3, 60)  # ~14k spheres, 32767 nodes
    assert sc_c.n > 13000
    _bvh_case(crt, orc, sc_c, sc_o)


def test_ppm_bit_exact(crt, orc):
    rng = np.random.default_rng(1)
    fb = (rng.random((9, 13, 3)) ** 3 * 1.3).astype(np.float32)
    fb[0, 0] = [0, -1, 5]
    assert crt.write_ppm(fb) == orc.write_ppm(fb)
    assert crt.write_ppm(fb).startswith(b"P3\n13 9\n255\n")


def test_no_device_fails_loudly(crt):
    """Without a usable gfx950 device the context cannot be created: no silent CPU fallback."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    ctx = C.c_void_p()
    rc = crt.lib().crt_create(0, 1, C.byref(ctx))
    assert rc in (2, 3) and not ctx.value
    assert crt.lib().crt_last_error(None)
    with pytest.raises(crt.CrtError):
        crt.Renderer(0, 1)
