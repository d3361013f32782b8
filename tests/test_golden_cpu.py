"""The oracle against its own committed golden outputs (tests/golden/oracle_images.npz, written by
tests/golden/make_golden.py): freezes rt_oracle.cpp bit for bit, on CPU."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def test_oracle_reproduces_golden(orc):
    g = np.load(os.path.join(HERE, "golden", "oracle_images.npz"))
    seed = int(g["seed"])
    for name, sc, depth in (("small", orc.scene_small(), 10), ("random", orc.scene_random(18, 11), 50)):
        nodes, order = orc.bvh_build(sc, 17)
        assert np.array_equal(order, g[name + "_leaf_order"])
        assert np.array_equal(orc.nodes_to_array(nodes), g[name + "_nodes"])
        cam = orc.camera(64, 4)
        a, _ = orc.render(sc, cam, seed, 0, 4, depth, None)
        b, st = orc.render(sc, cam, seed, 0, 4, depth, nodes, nthreads=4)
        assert np.array_equal(orc.resolve(a, 4).view(np.uint32), g[name + "_list"].view(np.uint32))
        assert np.array_equal(orc.resolve(b, 4).view(np.uint32), g[name + "_bvh"].view(np.uint32))
        assert np.array_equal(b, g[name + "_sums_bvh"])
        assert st == g[name + "_stats_bvh"].tolist()
        # the BVH is only an acceleration structure: same image as the brute-force list
        assert np.array_equal(a, b)
    assert orc.scene_random(18, 11).n == int(g["random_n"])


def test_oracle_reproduces_golden_paths_and_c1_frame(orc):
    """individual paths at the C1 / C2 geometry and the full C1 frame's SHA-256."""
    import hashlib
    g = np.load(os.path.join(HERE, "golden", "oracle_images.npz"))
    for name, sc, width, spp, depth in (("c1", orc.scene_small(), 400, 10, 10), ("c2", orc.scene_random(18, 11), 1200, 500, 50)):
        nodes, _ = orc.bvh_build(sc, 17)
        cam = orc.camera(width, spp)
        pps = g[name + "_paths_pixel_sample"]
        for world, wn in ((None, "list"), (nodes, "bvh")):
            for i in range(0, len(pps), 4):  # every fourth vector keeps the CPU suite quick
                ray7, rgb, seg = orc.trace_sample(sc, cam, 1, int(pps[i, 0]), int(pps[i, 1]), int(pps[i, 2]), depth, world)
                assert np.array_equal(ray7.view(np.uint32), g[f"{name}_paths_{wn}_ray"][i].view(np.uint32))
                assert np.array_equal(rgb.view(np.uint32), g[f"{name}_paths_{wn}_rgb"][i].view(np.uint32))
                assert seg == int(g[f"{name}_paths_{wn}_segments"][i])
            if name == "c1":
                sums, st = orc.render(sc, cam, 1, 0, 10, 10, world, nthreads=8)
                assert hashlib.sha256(orc.resolve(sums, 10).tobytes()).digest() == bytes(g[f"c1_frame_{wn}_sha256"])
                assert st[0] == int(g[f"c1_frame_{wn}_rays"])


def test_product_host_bvh_matches_golden(crt):
    g = np.load(os.path.join(HERE, "golden", "oracle_images.npz"))
    for name, sc in (("small", crt.scene_small()), ("random", crt.scene_random(18, 11))):
        nodes = crt.bvh_build_host(sc, 17)
        got = np.frombuffer(bytes(nodes), np.uint32).reshape(-1, 9)
        assert np.array_equal(got, g[name + "_nodes"])
