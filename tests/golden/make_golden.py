"""Writes tests/golden/oracle_images.npz from the CPU oracle (oracle/liboracle.so).

The reference ships no fixtures and cannot run here (DESIGN.md "Oracle"), so these vectors are
outputs of the oracle itself: they freeze its behaviour (a later edit of rt_oracle.cpp that
changes a bit shows up in tests/test_golden_cpu.py) and give the GPU tests a committed target.
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as orc  # noqa: E402

SEED, W, SPP = 20250220, 64, 4


def main():
    out = {"seed": np.int64(SEED)}
    for name, sc, depth in (("small", orc.scene_small(), 10), ("random", orc.scene_random(18, 11), 50)):
        nodes, order = orc.bvh_build(sc, 17)
        cam = orc.camera(W, SPP)
        a, st_a = orc.render(sc, cam, SEED, 0, SPP, depth, None)
        b, st_b = orc.render(sc, cam, SEED, 0, SPP, depth, nodes)
        out[name + "_list"] = orc.resolve(a, SPP)
        out[name + "_bvh"] = orc.resolve(b, SPP)
        out[name + "_sums_bvh"] = b
        out[name + "_rays"] = np.int64(st_b[0])
        out[name + "_stats_bvh"] = np.array(st_b, np.int64)
        out[name + "_leaf_order"] = order
        out[name + "_nodes"] = orc.nodes_to_array(nodes)
    # individual (pixel, sample) paths at the real C1 / C2 geometry -- primary ray (7 floats),
    # radiance, segment count, through the list and through the BVH
    rng = np.random.default_rng(SEED)
    for name, sc, width, spp, depth in (("c1", orc.scene_small(), 400, 10, 10), ("c2", orc.scene_random(18, 11), 1200, 500, 50)):
        nodes, _ = orc.bvh_build(sc, 17)
        cam = orc.camera(width, spp)
        n = 256
        px = rng.integers(0, cam.image_width, n).astype(np.int32)
        py = rng.integers(0, cam.image_height, n).astype(np.int32)
        smp = rng.integers(0, spp, n).astype(np.int32)
        out[name + "_paths_pixel_sample"] = np.stack([px, py, smp], 1)
        for world, wn in ((None, "list"), (nodes, "bvh")):
            rows = [orc.trace_sample(sc, cam, 1, int(px[i]), int(py[i]), int(smp[i]), depth, world) for i in range(n)]
            out[f"{name}_paths_{wn}_ray"] = np.stack([r[0] for r in rows])
            out[f"{name}_paths_{wn}_rgb"] = np.stack([r[1] for r in rows])
            out[f"{name}_paths_{wn}_segments"] = np.array([r[2] for r in rows], np.int32)
    # the full C1 frame (400x225, 10 spp, depth 10, seed 1), as SHA-256 of its fp32 bits
    import hashlib
    sc = orc.scene_small()
    nodes, _ = orc.bvh_build(sc, 17)
    cam = orc.camera(400, 10)
    for world, wn in ((None, "list"), (nodes, "bvh")):
        sums, st = orc.render(sc, cam, 1, 0, 10, 10, world, nthreads=8)
        out[f"c1_frame_{wn}_sha256"] = np.frombuffer(hashlib.sha256(orc.resolve(sums, 10).tobytes()).digest(), np.uint8).copy()
        out[f"c1_frame_{wn}_rays"] = np.int64(st[0])
    sc = orc.scene_random(18, 11)
    out["random_n"] = np.int64(sc.n)
    out["random_center_head"] = sc.center[:16].copy()
    np.savez_compressed(os.path.join(HERE, "oracle_images.npz"), **out)
    print({k: getattr(v, "shape", v) for k, v in out.items()})


if __name__ == "__main__":
    main()
