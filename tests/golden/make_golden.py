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
    sc = orc.scene_random(18, 11)
    out["random_n"] = np.int64(sc.n)
    out["random_center_head"] = sc.center[:16].copy()
    np.savez_compressed(os.path.join(HERE, "oracle_images.npz"), **out)
    print({k: getattr(v, "shape", v) for k, v in out.items()})


if __name__ == "__main__":
    main()
