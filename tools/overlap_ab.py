#!/usr/bin/env python3
"""Does the next image's launch hide the drain tail of the previous one?  K launches of a sample share (default 62 spp =
one rank's share of an 8-GPU split of C2) back to back: one context on one stream, against two contexts alternating on
two streams (their persistent grids overlap on the chip: the second launch's workgroups start as the first one's drain).
    python tools/overlap_ab.py [--spp 62] [--launches 40]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import cuda_ray_tracer_amd as crt  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp", type=int, default=62)
    ap.add_argument("--launches", type=int, default=40)
    ap.add_argument("--width", type=int, default=1200)
    args = ap.parse_args()
    sc = crt.scene_random(18, 11)
    cam = crt.camera(args.width, 500)
    ctxs, streams, bufs = [], [], []
    for _ in range(2):
        r = crt.Renderer(0, 1)
        r.upload_scene(sc)
        r.build_bvh(17)
        r.set_camera(cam)
        ctxs.append(r)
        streams.append(torch.cuda.Stream())
        bufs.append(torch.zeros(cam.image_height, cam.image_width, 3, dtype=torch.int64, device="cuda:0"))

    def run(n_ctx):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.launches):
            i = k % n_ctx
            with torch.cuda.stream(streams[i]):
                bufs[i].zero_()
                ctxs[i].render_accumulate(crt.CRT_WORLD_BVH, 0, args.spp, 50, bufs[i].data_ptr(), streams[i].cuda_stream)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.launches * 1e3

    for n_ctx in (1, 2, 1, 2):
        run(n_ctx)
        ms = min(run(n_ctx) for _ in range(3))
        rays = ctxs[0].stats().rays
        print(f"{n_ctx} context(s): {ms:.3f} ms per {args.spp}-spp launch, {rays / ms / 1e3:.0f} Mrays/s", flush=True)
    a = bufs[0].clone()
    assert torch.equal(a, bufs[1])  # same samples, same sums
    for r in ctxs:
        r.close()


if __name__ == "__main__":
    main()
