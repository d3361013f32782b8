"""Round-3 render-loop tests whose expected values do NOT come from oracle/ (closed form, float64 geometry, or
statistics), through the C ABI (crt_trace_samples: primary ray, radiance and segment count of single paths):

  * Snell refraction at ior 1.5 (vec3.h:138-143, material.h:69-81): the doubly refracted ray of a near-normal primary
    ray hits a grid of small absorbers behind the ball exactly where float64 geometry says;
  * Schlick reflectance and total internal reflection (material.h:75-78,92-97): reflected fraction against
    sum of r0 + (1-r0)(1-cos)^5 from outside; from inside, every decision of every bounce predicted from the Philox
    stream of the path (numpy restatement of Philox4x32-10 checked against the Random123 vectors) -- which also pins
    "no block is drawn under total internal reflection" (material.h:78 short circuit) and the block-per-event plan;
  * lambertian scatter is cosine-weighted (material.h:28, vec3.h:145-152): hit fraction of an absorber above the
    surface against the point-to-sphere form factor cos(beta) (R/D)^2, overhead and at a slant;
  * get_ray (render_with_cuda.h:11-26): lens samples uniform on the defocus disk, pixel offsets uniform in the pixel,
    time uniform on [0,1) (chi-square); and the primary ray itself, bit for bit, from the Philox stream and the
    reference's fp32 expressions in numpy float32;
  * the first scatter of a fuzzy metal and of a lambertian ball replayed from the stream (material.h:26-38,48-55).
"""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KNOBS = ("CRT_BLOCK", "CRT_SCHED", "CRT_TILES", "CRT_LDS_LIMIT", "CRT_PERSIST", "CRT_ARITH", "CRT_BVH_HOST", "CRT_TAIL_PASS",
         "CRT_UNIT_SAMPLES")


def _renderer(crt, scene, cam, seed, bvh=True):
    for k in KNOBS:
        os.environ.pop(k, None)
    r = crt.Renderer(0, seed)
    r.upload_scene(scene)
    if bvh:
        r.build_bvh(17)
    r.set_camera(cam)
    return r


def _bits_equal(a, b):
    return a.shape == b.shape and np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


# ---- Philox4x32-10 in numpy (Salmon et al., SC'11; constants of rocrand_philox4x32_10.h) ------------------------

def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over numpy uint64 arrays holding 32-bit values; returns four uint32 arrays."""
    M0, M1, W0, W1, mask = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85, 0xFFFFFFFF
    c0, c1, c2, c3 = (np.asarray(v, np.uint64) & mask for v in (c0, c1, c2, c3))
    k0, k1 = np.uint64(k0 & mask), np.uint64(k1 & mask)
    for _ in range(10):
        m0, m1 = np.uint64(M0) * c0, np.uint64(M1) * c2
        n0 = (m1 >> np.uint64(32)) ^ c1 ^ k0
        n2 = (m0 >> np.uint64(32)) ^ c3 ^ k1
        c0, c1, c2, c3 = n0, m1 & mask, n2, m0 & mask
        k0, k1 = (k0 + np.uint64(W0)) & mask, (k1 + np.uint64(W1)) & mask
    return tuple(v.astype(np.uint32) for v in (c0, c1, c2, c3))


def u01(x):
    """(0,1]: x * 2^-32 + 2^-33 in fp32 (rocRAND's mapping, DESIGN.md RNG)."""
    return np.float32(2.3283064e-10) + x.astype(np.float32) * np.float32(2.3283064e-10)


def test_numpy_philox_known_answers():
    """Random123 kat_vectors, "philox4x32 10" rows -- the restatement used below is the published function."""
    def one(ctr, key):
        return [int(v[0]) for v in philox4x32_10(*[np.array([c], np.uint64) for c in ctr], key[0], key[1])]
    assert one([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert one([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert one([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


# ---- float64 geometry helpers -------------------------------------------------------------------------------------

def _roots(o, d, c, rad):
    """Both intersection parameters of rays o + t d with a sphere (float64), NaN on a miss; and disc / a."""
    oc = c - o
    a = (d * d).sum(1)
    h = (d * oc).sum(1)
    disc = h * h - a * ((oc * oc).sum(1) - rad * rad)
    sq = np.sqrt(np.maximum(disc, 0))
    miss = disc < 0
    return np.where(miss, np.nan, (h - sq) / a), np.where(miss, np.nan, (h + sq) / a), disc / a


def _refract(ud, n, ri):
    """vec3.h:138-143 in float64: ud unit, n against the ray."""
    cos_t = np.minimum(-(ud * n).sum(1), 1.0)
    perp = ri[:, None] * (ud + cos_t[:, None] * n)
    par = -np.sqrt(np.abs(1.0 - (perp * perp).sum(1)))[:, None] * n
    return perp + par


def _unit(v):
    return v / np.linalg.norm(v, axis=1)[:, None]


def _schlick(cos_t, ri):
    r0 = ((1 - ri) / (1 + ri)) ** 2
    return r0 + (1 - r0) * (1 - cos_t) ** 5


# ---- (i) Snell ------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("world", ["bvh", "list"])
def test_snell_double_refraction_lands_where_float64_says(crt, world):
    """Glass ball, ior 1.5, pinhole camera.  A path with exactly 3 segments whose primary ray hits the ball has
    refracted in and out (a reflection at either interface gives 2 or >= 4 segments in this scene), so its third
    segment is the float64 double refraction of the primary ray (vec3.h:138-143 with ri = 1/1.5 going in,
    1.5 coming out; material.h:69-81).  Behind the ball stands a grid of small absorbers (base-class material: the
    path ends black at the hit): such a path is black exactly when that float64 ray hits one of them."""
    glass_c, glass_r, ior = np.array([0.0, 0.0, 0.0]), 1.0, 1.5
    gy, gz = np.meshgrid(np.linspace(-0.9, 0.9, 7), np.linspace(-0.9, 0.9, 7))
    targets = np.stack([np.full(gy.size, -2.2), gy.ravel(), gz.ravel()], axis=1)
    t_rad = 0.09
    n_t = len(targets)
    centers = np.vstack([glass_c[None, :], targets])
    sc = crt.Scene(centers, np.zeros((n_t + 1, 3)), [glass_r] + [t_rad] * n_t, [0] + [1] * n_t,
                   [crt.CRT_DIELECTRIC, crt.CRT_MATERIAL_BASE], [[1, 1, 1], [0, 0, 0]], [ior, 0.0])
    cam = crt.camera(400, 16, vfov=11.0, lookfrom=(9, 0.2, -0.1), lookat=(0, 0, 0), defocus_angle=0.0, focus_dist=9.0)
    r = _renderer(crt, sc, cam, 31, bvh=world == "bvh")
    rng = np.random.default_rng(11)
    n = 200000
    px, py, smp = rng.integers(0, 400, n), rng.integers(0, 225, n), rng.integers(0, 16, n)
    ray7, rgb, seg = r.trace_samples(crt.CRT_WORLD_BVH if world == "bvh" else crt.CRT_WORLD_LIST, px, py, smp, 50)
    r.close()
    o, d = ray7[:, 0:3].astype(np.float64), ray7[:, 3:6].astype(np.float64)
    t_in, _, _ = _roots(o, d, glass_c, glass_r)
    # no absorber in front of the ball on these rays (they all stand behind it), so "hits the ball" = t_in exists
    on_glass = ~np.isnan(t_in)
    p1 = o + np.nan_to_num(t_in)[:, None] * d
    n1 = (p1 - glass_c) / glass_r
    ud = _unit(d)
    cos_in = -(ud * n1).sum(1)
    d2 = _refract(ud, n1, np.full(n, 1.0 / ior))
    _, t_out, _ = _roots(p1, d2, glass_c, glass_r)          # from the surface: the far root
    p2 = p1 + np.nan_to_num(t_out)[:, None] * d2
    n2 = (p2 - glass_c) / glass_r                            # outward; the ray meets it from inside: normal = -n2, ri = ior
    ud2 = _unit(d2)
    sin_out = np.sqrt(np.maximum(0.0, 1.0 - np.minimum((ud2 * n2).sum(1), 1.0) ** 2))
    d3 = _refract(ud2, -n2, np.full(n, ior))
    hit_any = np.zeros(n, bool)
    margin = np.full(n, np.inf)
    for c in targets:
        t0, _, m = _roots(p2, _unit(d3), c, t_rad)
        hit_any |= ~np.isnan(t0) & (np.nan_to_num(t0) > 1e-3)
        margin = np.minimum(margin, np.abs(m))
    use = on_glass & (cos_in > 0.75) & (ior * sin_out < 0.95) & (seg == 3) & (margin > 2e-4)
    black = (rgb == 0).all(axis=1)
    assert use.sum() > 30000 and (use & hit_any).sum() > 2000 and (use & ~hit_any).sum() > 10000
    assert np.array_equal(black[use], hit_any[use])
    # and the straight continuation of the primary ray (no refraction at all) would classify many of them differently
    straight = np.zeros(n, bool)
    for c in targets:
        t0, _, _ = _roots(o, ud, c, t_rad)
        straight |= ~np.isnan(t0)
    assert (straight[use] != hit_any[use]).sum() > 2000


# ---- (ii) Schlick / total internal reflection ----------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("world", ["bvh", "list"])
def test_schlick_reflected_fraction_from_outside(crt, world):
    """One glass ball, nothing else: a primary ray that hits it is REFLECTED at the first interface exactly when its
    path has 2 segments (hit, then the sky: the ball is convex), refracted when it has >= 3.  The number of reflected
    paths must lie within 4 sigma of sum_i R_i, R_i = r0 + (1 - r0)(1 - cos_i)^5 (material.h:92-97) with cos_i the
    float64 incidence cosine of path i -- over the whole disk of the ball, so R_i runs from 0.04 to 1."""
    ior = 1.5
    sc = crt.Scene([[0, 0, 0]], np.zeros((1, 3)), [1.0], [0], [crt.CRT_DIELECTRIC], [[1, 1, 1]], [ior])
    cam = crt.camera(400, 64, vfov=17.0, lookfrom=(7, 0, 0), lookat=(0, 0, 0), defocus_angle=0.0, focus_dist=7.0)
    r = _renderer(crt, sc, cam, 77, bvh=world == "bvh")
    rng = np.random.default_rng(12)
    n = 300000
    px, py, smp = rng.integers(0, 400, n), rng.integers(0, 225, n), rng.integers(0, 64, n)
    ray7, rgb, seg = r.trace_samples(crt.CRT_WORLD_BVH if world == "bvh" else crt.CRT_WORLD_LIST, px, py, smp, 50)
    r.close()
    o, d = ray7[:, 0:3].astype(np.float64), ray7[:, 3:6].astype(np.float64)
    t_in, _, m = _roots(o, d, np.zeros(3), 1.0)
    hit = ~np.isnan(t_in) & (np.abs(m) > 1e-5)
    assert (seg[np.isnan(t_in) & (np.abs(m) > 1e-5)] == 1).all()
    p1 = o + np.nan_to_num(t_in)[:, None] * d
    cos_in = np.minimum(-(_unit(d) * p1).sum(1), 1.0)
    R = _schlick(cos_in, 1.0 / ior)
    assert hit.sum() > 100000 and (seg[hit] >= 2).all()
    for lo, hi in ((0.0, 1.01), (0.0, 0.3), (0.3, 0.6), (0.6, 1.01)):  # all, and by band of the incidence cosine
        band = hit & (cos_in >= lo) & (cos_in < hi)
        expect, sigma = R[band].sum(), np.sqrt((R[band] * (1 - R[band])).sum())
        got = int((seg[band] == 2).sum())
        assert band.sum() > 5000 and abs(got - expect) < 4 * sigma, (lo, hi, got, expect, sigma)
    # r0 alone (no (1 - cos)^5 term) or a cubic instead of the fifth power would be far outside:
    r0 = ((1 - 1 / ior) / (1 + 1 / ior)) ** 2
    got, sigma = int((seg[hit] == 2).sum()), np.sqrt((R[hit] * (1 - R[hit])).sum())
    assert abs(got - r0 * hit.sum()) > 20 * sigma
    assert abs(got - (r0 + (1 - r0) * (1 - cos_in[hit]) ** 3).sum()) > 20 * sigma


@pytest.mark.gpu
@pytest.mark.parametrize("world", ["bvh", "list"])
def test_inside_a_glass_ball_every_bounce_follows_the_philox_stream(crt, world):
    """Camera INSIDE a glass ball (ior 1.5).  A ray inside a sphere keeps its incidence angle from bounce to bounce, so
    (a) beyond the critical angle (1.5 sin > 1) it must reflect for ever: the path uses all max_depth segments and is
    black -- total internal reflection is certain, not likely (material.h:75-78);
    (b) below it, bounce j reflects exactly when reflectance(cos, 1.5) > 1 - u_j, u_j = first word of the Philox block
    of event e0 + j of this (pixel, sample) (DESIGN.md RNG / D16: camera block, disk blocks, then ONE block per
    dielectric decision), so the segment count of every path is predicted from the stream.
    Mixed paths -- totally reflected first, decided by a draw later -- are in the next test."""
    ior, depth, seed = 1.5, 40, 123
    sc = crt.Scene([[0, 0, 0]], np.zeros((1, 3)), [1.0], [0], [crt.CRT_DIELECTRIC], [[1, 1, 1]], [ior])
    W, spp = 200, 32
    cam = crt.camera(W, spp, vfov=100.0, lookfrom=(0.85, 0.0, 0.0), lookat=(0.85, 0.0, 1.0), defocus_angle=0.0, focus_dist=1.0)
    r = _renderer(crt, sc, cam, seed, bvh=world == "bvh")
    rng = np.random.default_rng(13)
    n = 60000
    H = cam.image_height
    px, py, smp = rng.integers(0, W, n), rng.integers(0, H, n), rng.integers(0, spp, n)
    ray7, rgb, seg = r.trace_samples(crt.CRT_WORLD_BVH if world == "bvh" else crt.CRT_WORLD_LIST, px, py, smp, depth)
    r.close()
    o, d = ray7[:, 0:3].astype(np.float64), ray7[:, 3:6].astype(np.float64)
    _, t_out, _ = _roots(o, d, np.zeros(3), 1.0)
    p = o + t_out[:, None] * d
    cos_t = np.minimum((_unit(d) * p).sum(1), 1.0)            # against the inward normal -p: -ud . (-p)
    sin_t = np.sqrt(np.maximum(0.0, 1 - cos_t ** 2))
    tir = ior * sin_t > 1 + 1e-4
    sub = ior * sin_t < 1 - 1e-4
    assert tir.sum() > 10000 and sub.sum() > 10000
    black = (rgb == 0).all(axis=1)
    assert (seg[tir] == depth).all() and black[tir].all()                       # (a)
    # (b) the stream: pixel index, sample; event 0 = camera block, then disk blocks until an attempt is accepted
    pixel = (py * W + px).astype(np.uint64)
    sample = smp.astype(np.uint64)
    k0, k1 = seed & 0xffffffff, seed >> 32

    def block(event):
        return philox4x32_10(event.astype(np.uint64), sample, pixel, np.zeros(n, np.uint64), k0, k1)

    def rf11(u):  # rtweekend.h:25-28 in fp32: -1 + 2 * (1 - u)
        return np.float32(-1.0) + np.float32(2.0) * (np.float32(1.0) - u)

    event = np.ones(n, np.uint64)          # first disk block
    open_ = np.ones(n, bool)
    for _ in range(12):                    # vec3.h:154-160, two attempts per block
        b = block(event)
        x0, y0, x1, y1 = (rf11(u01(v)) for v in b)
        ok = (x0 * x0 + y0 * y0 < np.float32(1.0)) | (x1 * x1 + y1 * y1 < np.float32(1.0))
        event = np.where(open_, event + np.uint64(1), event)
        open_ &= ~ok
    assert not open_.any()
    R = _schlick(cos_t, ior)
    pred = np.full(n, depth)
    alive = sub.copy()
    borderline = np.zeros(n, bool)
    for j in range(depth):                 # hit j (segment j+1): reflect, or refract out and escape with segment j+2
        u = u01(block(event + np.uint64(j))[0]).astype(np.float64)
        borderline |= alive & (np.abs(R - (1.0 - u)) < 1e-5)
        leaves = alive & ~(R > 1.0 - u)
        if j + 2 <= depth:
            pred[leaves] = j + 2
        alive &= ~leaves
    use = sub & ~borderline
    assert use.sum() > 10000
    assert np.array_equal(seg[use], pred[use])                                  # (b): every decision of every bounce
    escaped = use & (pred < depth)
    assert (~black[escaped]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("world", ["bvh", "list"])
def test_no_philox_block_is_drawn_under_total_internal_reflection(crt, world):
    """material.h:78: `cannot_refract || reflectance(...) > random_float(state)` short-circuits -- a totally reflected
    hit consumes no random number.  Camera inside a glass ball B (ior 1.5) near its wall, and a fuzz-0 mirror ball M
    inside B: a path typically bounces off B's wall by total internal reflection (the incidence angle would stay the
    same for ever), then meets M, which changes the angle; once sub-critical, its reflect / refract decision uses the
    Philox block of event e0 + (blocks drawn so far) -- camera block, disk blocks, the attempts of every
    random_unit_vector the mirror drew (material.h:51 draws it even with fuzz 0), one block per NON-TIR dielectric hit,
    and nothing for the TIR hits.  The float64 replay below advances the event counter that way and predicts the
    segment count of every path; had the kernel drawn a block on its TIR hits, the paths with decisions AFTER a TIR
    hit (thousands) would come out different.  Decisions closer to their threshold than the replay's accumulated
    rounding (a margin growing with the bounce index) are left out."""
    ior, depth, seed = 1.5, 8, 321
    ball_c, ball_r = np.array([0.0, 0.0, 0.0]), 1.0
    mir_c, mir_r = np.array([-0.5, 0.05, 0.0]), 0.42
    sc = crt.Scene([ball_c, mir_c], np.zeros((2, 3)), [ball_r, mir_r], [0, 1], [crt.CRT_DIELECTRIC, crt.CRT_METAL],
                   [[1, 1, 1], [1, 1, 1]], [ior, 0.0])
    W, spp = 160, 32
    cam = crt.camera(W, spp, vfov=80.0, lookfrom=(0.8, 0.0, 0.0), lookat=(0.8, 1.0, 0.2), defocus_angle=0.0, focus_dist=1.0)
    r = _renderer(crt, sc, cam, seed, bvh=world == "bvh")
    rng = np.random.default_rng(14)
    n = 40000
    H = cam.image_height
    px, py, smp = rng.integers(0, W, n), rng.integers(0, H, n), rng.integers(0, spp, n)
    ray7, rgb, seg = r.trace_samples(crt.CRT_WORLD_BVH if world == "bvh" else crt.CRT_WORLD_LIST, px, py, smp, depth)
    r.close()
    pixel, sample = (py * W + px).astype(np.uint64), smp.astype(np.uint64)
    k0, k1 = seed & 0xffffffff, seed >> 32

    def block(event):
        return philox4x32_10(event.astype(np.uint64), sample, pixel, np.zeros(n, np.uint64), k0, k1)

    def rf11(u):
        return np.float32(-1.0) + np.float32(2.0) * (np.float32(1.0) - u)

    event = np.ones(n, np.uint64)
    open_ = np.ones(n, bool)
    for _ in range(12):
        b = block(event)
        x0, y0, x1, y1 = (rf11(u01(v)) for v in b)
        ok = (x0 * x0 + y0 * y0 < np.float32(1.0)) | (x1 * x1 + y1 * y1 < np.float32(1.0))
        event = np.where(open_, event + np.uint64(1), event)
        open_ &= ~ok
    # float64 replay of ray_color: position, direction, next Philox event; every ray stays inside B until it refracts out
    o, d = ray7[:, 0:3].astype(np.float64), _unit(ray7[:, 3:6].astype(np.float64))
    pred = np.full(n, depth)
    alive = np.ones(n, bool)
    unsure = np.zeros(n, bool)
    tir_hits = np.zeros(n, int)
    drawn_after_tir = np.zeros(n, bool)
    for j in range(depth):
        eps = min(0.2, 2e-6 * 5.0 ** j)                                   # float64 replay vs fp32 kernel: error grows per bounce
        t_m, _, marg_m = _roots(o, d, mir_c, mir_r)
        _, t_b, _ = _roots(o, d, ball_c, ball_r)
        hit_m = ~np.isnan(t_m) & (np.nan_to_num(t_m) > 1e-3)
        unsure |= alive & (np.abs(marg_m) < 50 * eps)                    # grazing the mirror
        t = np.where(hit_m, t_m, t_b)
        p = o + t[:, None] * d
        # mirror (metal fuzz 0, material.h:50-51; it draws one unit vector -> blocks of random_unit_vector: the attempts
        # until acceptance, vec3.h:145-152): direction reflect(d, n); the event counter advances by the attempts
        nm = _unit(p - mir_c)
        refl_m = d - 2 * (d * nm).sum(1)[:, None] * nm
        att = np.zeros(n, np.uint64)
        open_ = alive & hit_m
        for _ in range(40):
            if not open_.any():
                break
            b = block(event + att)
            x, y, z = (rf11(u01(v)) for v in b[:3])
            l2 = x * x + y * y + z * z
            ok = (np.float32(0.0) < l2) & (l2 <= np.float32(1.0))
            att = np.where(open_, att + np.uint64(1), att)
            open_ &= ~ok
        unsure |= open_
        # glass wall from inside
        nb = -p / ball_r                                               # against the ray (front_face false)
        cos_t = np.minimum(-(d * nb).sum(1), 1.0)
        sin_t = np.sqrt(np.maximum(0.0, 1 - cos_t ** 2))
        tir = ior * sin_t > 1.0
        unsure |= alive & ~hit_m & (np.abs(ior * sin_t - 1.0) < 50 * eps)
        u = u01(block(event)[0]).astype(np.float64)
        R = _schlick(cos_t, ior)
        unsure |= alive & ~hit_m & ~tir & (np.abs(R - (1.0 - u)) < 5 * eps)
        reflect_b = tir | (R > 1.0 - u)
        refl_b = d - 2 * (d * nb).sum(1)[:, None] * nb
        wall = alive & ~hit_m
        leaves = wall & ~reflect_b
        if j + 2 <= depth:
            pred[leaves] = j + 2
        drawn_after_tir |= wall & ~tir & (tir_hits > 0)
        tir_hits += (wall & tir).astype(int)
        event = np.where(alive & hit_m, event + att, np.where(wall & ~tir, event + np.uint64(1), event))
        alive &= ~leaves
        d = np.where(hit_m[:, None], _unit(refl_m), _unit(refl_b))
        o = p
    use = ~unsure
    stats = dict(use=float(use.mean()), after_tir=int((use & drawn_after_tir).sum()), mism=int((seg[use] != pred[use]).sum()),
                 tir_paths=int((tir_hits > 0).sum()), after_tir_all=int(drawn_after_tir.sum()))
    assert use.mean() > 0.5, stats
    assert np.array_equal(seg[use], pred[use]), stats
    # the case the test exists for: decisions drawn AFTER totally reflected hits, in thousands of paths
    assert (use & drawn_after_tir).sum() > 2000, stats


# ---- (iii) lambertian scatter is cosine-weighted -------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("target_c,target_r,tell_uniform", [((0.0, 3.2, 0.0), 0.9, True), ((3.1, 1.9, 0.0), 0.9, False)])
def test_lambertian_scatter_is_cosine_weighted(crt, target_c, target_r, tell_uniform):
    """material.h:28: direction = normal + random_unit_vector -- a cosine-weighted direction.  The probability that it
    hits a sphere of radius R at distance D whose centre stands at angle beta from the normal, wholly above the tangent
    plane, is the point-to-sphere form factor cos(beta) (R / D)^2.  A lambertian ball (hit first by the primary ray) and
    one absorber: the number of paths that end black after 2 segments against sum_i cos(beta_i) (R / D_i)^2, within
    4 sigma -- overhead (beta ~ 0..40 deg) and at a slant (beta ~ 35..75 deg).  A uniform hemisphere would give
    (1 - sqrt(1 - (R/D)^2)) ~ (R/D)^2 / 2 instead: overhead that is half the cosine-weighted count, far outside the
    bound (at the slant, cos(beta) ~ 1/2 and the two laws nearly agree, so only the overhead case tells them apart)."""
    lam_c, lam_r = np.array([0.0, 0.0, 0.0]), 1.0
    tc = np.array(target_c, np.float64)
    sc = crt.Scene([lam_c, tc], np.zeros((2, 3)), [lam_r, target_r], [0, 1], [crt.CRT_LAMBERTIAN, crt.CRT_MATERIAL_BASE],
                   [[0.8, 0.8, 0.8], [0, 0, 0]], [0.0, 0.0])
    # looking down on the top of the ball from the side opposite to the slanted target, so neither target is in the way
    cam = crt.camera(320, 64, vfov=6.0, lookfrom=(-6.0, 9.0, 3.0), lookat=(0.0, 0.85, 0.0), defocus_angle=0.0, focus_dist=10.0)
    r = _renderer(crt, sc, cam, 55)
    rng = np.random.default_rng(15)
    n = 400000
    px, py, smp = rng.integers(0, 320, n), rng.integers(0, 180, n), rng.integers(0, 64, n)
    ray7, rgb, seg = r.trace_samples(crt.CRT_WORLD_BVH, px, py, smp, 50)
    r.close()
    o, d = ray7[:, 0:3].astype(np.float64), ray7[:, 3:6].astype(np.float64)
    t_l, _, m_l = _roots(o, d, lam_c, lam_r)
    t_t, _, _ = _roots(o, d, tc, target_r)
    first_lam = ~np.isnan(t_l) & np.isnan(t_t) & (m_l > 1e-3)
    p = o + np.nan_to_num(t_l)[:, None] * d
    nrm = (p - lam_c) / lam_r
    to_c = tc - p
    D = np.linalg.norm(to_c, axis=1)
    cos_b = (to_c * nrm).sum(1) / D
    above = first_lam & (D * cos_b > target_r * 1.02)          # the absorber wholly above the tangent plane at p
    P = cos_b * (target_r / D) ** 2
    assert above.sum() > 60000, int(above.sum())
    black2 = (rgb == 0).all(axis=1) & (seg == 2)
    got, expect = int(black2[above].sum()), P[above].sum()
    sigma = np.sqrt((P[above] * (1 - P[above])).sum())
    assert abs(got - expect) < 4 * sigma, (got, expect, sigma)
    uniform = (1 - np.sqrt(1 - (target_r / D[above]) ** 2)).sum()  # solid angle / 2 pi
    if tell_uniform:
        assert abs(got - uniform) > 12 * sigma, (got, uniform, sigma)


# ---- (iv) get_ray: lens, pixel and time samples ---------------------------------------------------------------------

def _chi2(counts):
    e = counts.sum() / len(counts)
    return float(((counts - e) ** 2 / e).sum())


@pytest.mark.gpu
def test_lens_pixel_and_time_samples_are_uniform(crt):
    """render_with_cuda.h:11-26: the ray origin is center + p.x * defocus_disk_u + p.y * defocus_disk_v with p uniform
    in the unit disk (vec3.h:154-160: r^2 uniform on [0,1), angle uniform); the pixel sample is uniform in the pixel's
    footprint; the time is uniform on [0,1).  Chi-square over 20 bins each, 400 000 paths: bound 60 (19 degrees of
    freedom: p < 1e-5), plus the pairwise independence of lens x and y signs."""
    cam = crt.camera(1200, 500)
    sc = crt.scene_random(18, 3)
    r = _renderer(crt, sc, cam, 9)
    rng = np.random.default_rng(16)
    n = 400000
    px, py, smp = rng.integers(0, 1200, n), rng.integers(0, 675, n), rng.integers(0, 500, n)
    ray7, _, _ = r.trace_samples(crt.CRT_WORLD_BVH, px, py, smp, 1)
    r.close()
    c = np.array(cam.center[:], np.float64)
    ddu, ddv = np.array(cam.defocus_disk_u[:], np.float64), np.array(cam.defocus_disk_v[:], np.float64)
    off = ray7[:, 0:3].astype(np.float64) - c
    A = np.stack([ddu, ddv], axis=1)
    coef, *_ = np.linalg.lstsq(A, off.T, rcond=None)
    x, y = coef
    r2 = x * x + y * y
    assert (r2 < 1 + 1e-4).all()
    bins = 20
    assert _chi2(np.bincount(np.minimum((r2 * bins).astype(int), bins - 1), minlength=bins)) < 60
    ang = (np.arctan2(y, x) + np.pi) / (2 * np.pi)
    assert _chi2(np.bincount(np.minimum((ang * bins).astype(int), bins - 1), minlength=bins)) < 60
    tm = ray7[:, 6].astype(np.float64)
    assert ((tm >= 0) & (tm < 1)).all()
    assert _chi2(np.bincount(np.minimum((tm * bins).astype(int), bins - 1), minlength=bins)) < 60
    p00, du, dv = (np.array(v[:], np.float64) for v in (cam.pixel00_loc, cam.pixel_delta_u, cam.pixel_delta_v))
    target = ray7[:, 0:3].astype(np.float64) + ray7[:, 3:6].astype(np.float64) - p00
    B = np.stack([du, dv], axis=1)
    uv, *_ = np.linalg.lstsq(B, target.T, rcond=None)
    for frac in (uv[0] - px + 0.5, uv[1] - py + 0.5):
        frac = np.clip(frac, 0, 1 - 1e-9)
        assert _chi2(np.bincount((frac * bins).astype(int), minlength=bins)) < 60
    # lens sample, pixel offset and time come from different words / blocks of the stream: no correlation
    for a, b in ((x, y), (x, tm), (uv[0] - px, tm), (r2, uv[1] - py)):
        assert abs(np.corrcoef(a, b)[0, 1]) < 5 / np.sqrt(n)


# ---- (v) the draw plan replayed: primary rays bit for bit, and the first scatter of metal / lambertian -------------

def _stream(seed, W, px, py, smp):
    """block(event) of the Philox stream of the paths (px, py, smp): counter (event, sample, pixel, 0), key = seed."""
    pixel, sample = (np.asarray(py) * W + np.asarray(px)).astype(np.uint64), np.asarray(smp).astype(np.uint64)
    zero = np.zeros(len(pixel), np.uint64)
    return lambda event: philox4x32_10(np.asarray(event, np.uint64), sample, pixel, zero, seed & 0xffffffff, seed >> 32)


def _rf11(u):  # rtweekend.h:25-28 in fp32: -1 + 2 * (1 - u)
    return np.float32(-1.0) + np.float32(2.0) * (np.float32(1.0) - u)


def _disk_sample(block, n):
    """vec3.h:154-160 from event 1 on, two attempts per block: the accepted (x, y) in fp32 and the next event index."""
    one = np.float32(1.0)
    event, open_ = np.ones(n, np.uint64), np.ones(n, bool)
    dx, dy = np.zeros(n, np.float32), np.zeros(n, np.float32)
    for _ in range(16):
        x0, y0, x1, y1 = (_rf11(u01(v)) for v in block(event))
        ok0, ok1 = (x0 * x0 + y0 * y0) < one, (x1 * x1 + y1 * y1) < one
        t0, t1 = open_ & ok0, open_ & ~ok0 & ok1
        dx, dy = np.where(t0, x0, np.where(t1, x1, dx)), np.where(t0, y0, np.where(t1, y1, dy))
        event = np.where(open_, event + np.uint64(1), event)
        open_ = open_ & ~(ok0 | ok1)
    assert not open_.any()
    return dx, dy, event


def _unit_vector(block, event, todo):
    """vec3.h:145-152 from `event` on, one block per attempt: the accepted vector (float64 unit) and the blocks drawn."""
    n = len(event)
    used, open_ = np.zeros(n, np.uint64), todo.copy()
    v = np.zeros((n, 3))
    for _ in range(64):
        if not open_.any():
            break
        x, y, z = (_rf11(u01(w)) for w in block(event + used)[:3])
        l2 = x * x + y * y + z * z
        ok = (np.float32(0.0) < l2) & (l2 <= np.float32(1.0))
        take = open_ & ok
        v[take] = (np.stack([x, y, z], axis=1)[take].astype(np.float64)) / np.sqrt(l2[take].astype(np.float64))[:, None]
        used = np.where(open_, used + np.uint64(1), used)
        open_ = open_ & ~ok
    assert not open_.any()
    return v, used


@pytest.mark.gpu
def test_primary_ray_replayed_from_the_philox_stream(crt):
    """get_ray (render_with_cuda.h:6-29) from the stream alone: the camera block gives the pixel offsets and the time
    (DESIGN.md D16), the disk blocks the lens sample; origin and direction are then the reference's fp32 expressions,
    one rounding per operation, in numpy float32.  200 000 primary rays at C2's camera (defocus 0.6): bit for bit."""
    seed, W, spp = 4242, 1200, 500
    cam = crt.camera(W, spp)
    r = _renderer(crt, crt.scene_random(18, 3), cam, seed)
    rng = np.random.default_rng(21)
    n = 200000
    px, py, smp = rng.integers(0, W, n), rng.integers(0, cam.image_height, n), rng.integers(0, spp, n)
    ray7, _, _ = r.trace_samples(crt.CRT_WORLD_BVH, px, py, smp, 1)
    r.close()
    block = _stream(seed, W, px, py, smp)
    b0 = block(np.zeros(n, np.uint64))
    half, one = np.float32(0.5), np.float32(1.0)
    x_off, y_off, tm = u01(b0[0]) - half, u01(b0[1]) - half, one - u01(b0[2])
    dx, dy, _ = _disk_sample(block, n)
    f = lambda v: np.array(v[:], np.float32)
    c, p00, du, dv, ddu, ddv = (f(v) for v in (cam.center, cam.pixel00_loc, cam.pixel_delta_u, cam.pixel_delta_v,
                                                cam.defocus_disk_u, cam.defocus_disk_v))
    sx, sy = px.astype(np.float32) + x_off, py.astype(np.float32) + y_off
    sample_pt = (p00[None, :] + sx[:, None] * du[None, :]) + sy[:, None] * dv[None, :]
    origin = (c[None, :] + dx[:, None] * ddu[None, :]) + dy[:, None] * ddv[None, :]
    want = np.concatenate([origin, sample_pt - origin, tm[:, None]], axis=1).astype(np.float32)
    assert _bits_equal(ray7, want)


@pytest.mark.gpu
@pytest.mark.parametrize("world", ["bvh", "list"])
@pytest.mark.parametrize("material", ["metal_fuzz", "lambertian"])
def test_first_scatter_replayed_from_the_philox_stream(crt, world, material):
    """material.h:26-38 / 48-55 with the stream's own unit vector: a ball S (metal, fuzz 0.4 -- or lambertian) and an
    absorber T.  For a primary ray that hits S first, the float64 replay draws random_unit_vector from the blocks that
    follow the disk blocks (one block per attempt), forms unit(reflect(d, n)) + 0.4 u (metal: absorbed -- black after
    ONE segment -- exactly when that direction points into the surface) or n + u (lambertian), and follows it: onto T
    (black after two segments) or to the sky (lit, two segments).  Every path's (black, segments) must come out as
    predicted; decisions within rounding of their threshold are left out."""
    seed, W, spp, fuzz = 99, 320, 64, 0.4
    s_c, s_r = np.array([0.0, 0.0, 0.0]), 1.0
    t_c, t_r = np.array([1.9, 1.7, 0.4]), 0.9
    mtype = crt.CRT_METAL if material == "metal_fuzz" else crt.CRT_LAMBERTIAN
    sc = crt.Scene([s_c, t_c], np.zeros((2, 3)), [s_r, t_r], [0, 1], [mtype, crt.CRT_MATERIAL_BASE],
                   [[0.8, 0.7, 0.6], [0, 0, 0]], [fuzz, 0.0])
    cam = crt.camera(W, spp, vfov=16.0, lookfrom=(6.5, 1.2, 0.8), lookat=(0, 0, 0), defocus_angle=0.0, focus_dist=6.5)
    r = _renderer(crt, sc, cam, seed, bvh=world == "bvh")
    rng = np.random.default_rng(22)
    n = 120000
    px, py, smp = rng.integers(0, W, n), rng.integers(0, cam.image_height, n), rng.integers(0, spp, n)
    ray7, rgb, seg = r.trace_samples(crt.CRT_WORLD_BVH if world == "bvh" else crt.CRT_WORLD_LIST, px, py, smp, 50)
    r.close()
    block = _stream(seed, W, px, py, smp)
    _, _, event = _disk_sample(block, n)
    o, d = ray7[:, 0:3].astype(np.float64), ray7[:, 3:6].astype(np.float64)
    t_s, _, m_s = _roots(o, d, s_c, s_r)
    t_t, _, m_t = _roots(o, d, t_c, t_r)
    first_s = ~np.isnan(t_s) & (np.isnan(t_t) | (t_s < t_t)) & (np.abs(m_s) > 1e-4) & (np.isnan(t_t) | (np.abs(m_t) > 1e-4))
    p = o + np.nan_to_num(t_s)[:, None] * d
    nrm = (p - s_c) / s_r
    u, _ = _unit_vector(block, event, first_s)
    if material == "metal_fuzz":
        refl = d - 2 * (d * nrm).sum(1)[:, None] * nrm
        direction = _unit(np.where(first_s[:, None], refl, 1.0)) + fuzz * u
    else:
        direction = nrm + u
    facing = (direction * nrm).sum(1)
    absorbed = (facing <= 0) if material == "metal_fuzz" else np.zeros(n, bool)
    t2, _, m2 = _roots(p, _unit(np.where(first_s[:, None], direction, 1.0)), t_c, t_r)
    on_target = ~np.isnan(t2) & (np.nan_to_num(t2) > 1e-3)
    # (a direction within a few degrees of the tangent plane can meet S again beyond t = 0.001 within fp32 rounding --
    # the reference's own acne, sphere.h:59-64 -- and a metal direction within rounding of the plane can flip the
    # absorption test: both left out)
    grazing = np.abs(facing) < 0.05 * np.linalg.norm(direction, axis=1)
    sure = first_s & ~grazing & (np.abs(m2) > 1e-4) & (np.linalg.norm(direction, axis=1) > 1e-3)
    pred_black = absorbed | on_target
    pred_seg = np.where(absorbed, 1, 2)
    black = (rgb == 0).all(axis=1)
    assert sure.sum() > 40000 and (sure & on_target & ~absorbed).sum() > 1500
    if material == "metal_fuzz":
        assert (sure & absorbed).sum() > 300
    assert np.array_equal(black[sure], pred_black[sure])
    assert np.array_equal(seg[sure], pred_seg[sure])
    lit = sure & ~pred_black                      # one bounce: attenuation = albedo, sky of the PRIMARY ray (D2)
    dd = ray7[:, 3:6]
    len2 = dd[:, 0] * dd[:, 0] + dd[:, 1] * dd[:, 1] + dd[:, 2] * dd[:, 2]
    a = np.float32(0.5) * ((np.float32(1.0) / np.sqrt(len2)) * dd[:, 1] + np.float32(1.0))
    sky = np.stack([(np.float32(1.0) - a) * np.float32(1.0) + a * np.float32(c) for c in (0.5, 0.7, 1.0)], axis=1)
    albedo = np.array([0.8, 0.7, 0.6], np.float32)
    assert _bits_equal(rgb[lit], (albedo[None, :] * sky[lit]).astype(np.float32))


@pytest.mark.gpu
def test_short_reciprocal_and_root_are_the_general_ones_for_every_argument(crt):
    """The exact kernels form 1/x (aabb.h:48, vec3.h:114-116, sphere.h:66) for |x| in [2^-100, 2^100] from v_rcp_f32 and
    one Newton step instead of the compiler's 11-instruction division, and sqrt(x) (sphere.h:55, vec3.h:130,141,
    material.h:73) for x in [2^-96, inf) without the general sequence's scaling and fix-up.  Proof by exhaustion on the
    device: all 2^32 arguments of each, bit patterns compared (crt_selftest_exact_math)."""
    r = crt.Renderer(0, 1)
    try:
        res = r.selftest_exact_math()
    finally:
        r.close()
    for name, (bad, first, short) in res.items():
        assert bad == 0, f"{name}: {bad} arguments differ, the smallest has bits {first:#010x}"
    assert res["rcp"][2] == 2 * (200 * (1 << 23) + 1)  # 2 signs x (the 200 binades of [2^-100, 2^100) x 2^23 mantissas + 2^100 itself)
    assert res["sqrt"][2] == res["rsqrt"][2] == (96 + 128) * (1 << 23)  # the binades of [2^-96, 2^128)
    # the radiance -> 2^-40 fixed point conversion: +0 ... +inf, -0 and the NaNs (which fminf turns into 4096)
    assert res["fixed"][2] == (0x7F800000 + 1) + 1 + 2 * ((1 << 23) - 1)
