#!/usr/bin/env python3
"""One rank's 1/G share of an image, timed both ways on ONE GPU (BASELINE.json configs[4] "tile/sample balancing"):
  samples: rank g traces samples [g spp/G, (g+1) spp/G) of every 8x8 tile   (crt_render_accumulate)
  tiles:   rank g traces all spp samples of the tiles t = g mod G             (crt_render_accumulate_tiles)
Every share g = 0..G-1 is rendered in turn; a node's step takes as long as its slowest share, so the figure of merit
is max over g (then comes the combine: reduce(SUM) for samples, a gather would do for tiles).
    python tools/share_ab.py --width 1200 --spp 500 --gpus 8 [--rounds 3] [--arith fast]"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import cuda_ray_tracer_amd as crt  # noqa: E402
from cuda_ray_tracer_amd.multi_gpu import share_of_rank  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1200)
    ap.add_argument("--spp", type=int, default=500)
    ap.add_argument("--gpus", type=int, nargs="+", default=[2, 4, 8])
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--arith", default="exact")
    args = ap.parse_args()
    r = crt.Renderer(0, 1)
    r.set_arith(crt.CRT_ARITH_FAST if args.arith == "fast" else crt.CRT_ARITH_EXACT)
    r.upload_scene(crt.scene_random(18, 11))
    r.build_bvh(17)
    cam = crt.camera(args.width, args.spp)
    r.set_camera(cam)
    buf = torch.zeros(cam.image_height, cam.image_width, 3, dtype=torch.int64, device="cuda:0")

    def timed(sh):
        ts = []
        for _ in range(args.rounds + 1):
            r.render_accumulate(crt.CRT_WORLD_BVH, sh["first"], sh["nsamples"], args.depth, buf.data_ptr(), 0,
                                tile_stride=sh["tile_stride"], tile_offset=sh["tile_offset"])
            st = r.stats()
            ts.append(st.kernel_seconds * 1e3)
        return statistics.median(ts[1:]), st.rays

    # the same share back to back from TWO contexts alternating on two streams (bench.py's double-buffered steps for
    # N > 1): the next launch's workgroups start in the drain tail of the previous one; ms per launch, wall clock
    import time
    r2 = crt.Renderer(0, 1)
    r2.set_arith(crt.CRT_ARITH_FAST if args.arith == "fast" else crt.CRT_ARITH_EXACT)
    r2.upload_scene(crt.scene_random(18, 11))
    r2.build_bvh(17)
    r2.set_camera(cam)
    buf2 = torch.zeros_like(buf)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def piped(sh, launches=24):
        best = 1e9
        for _ in range(args.rounds + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(launches):
                rr, bb, ss = ((r, buf, streams[0]), (r2, buf2, streams[1]))[k % 2]
                with torch.cuda.stream(ss):
                    bb.zero_()
                    rr.render_accumulate(crt.CRT_WORLD_BVH, sh["first"], sh["nsamples"], args.depth, bb.data_ptr(), ss.cuda_stream,
                                         tile_stride=sh["tile_stride"], tile_offset=sh["tile_offset"])
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / launches * 1e3)
        return best

    full_ms, full_rays = timed(share_of_rank(args.spp, 1, 0))
    print(f"{args.width}x{cam.image_height} {args.spp} spp, arith {args.arith}: full frame {full_ms:.2f} ms, {full_rays} rays, "
          f"{full_rays / full_ms / 1e3:.0f} Mrays/s", flush=True)
    for G in args.gpus:
        for split in ("samples", "tiles"):
            per = [timed(share_of_rank(args.spp, G, g, split)) for g in range(G)]
            ms = [p[0] for p in per]
            rays = [p[1] for p in per]
            assert sum(rays) == full_rays
            print(f"  G={G} {split:8s}: share ms min {min(ms):.2f} max {max(ms):.2f} (ideal {full_ms / G:.2f}); rays/share min {min(rays)} "
                  f"max {max(rays)} (imbalance {max(rays) / (full_rays / G) - 1:+.2%}); node rate before the combine "
                  f"{full_rays / max(ms) / 1e3:.0f} Mrays/s = {full_ms / max(ms) / G:.1%} of linear", flush=True)
            if split == "samples":
                g_slow = ms.index(max(ms))
                pm = piped(share_of_rank(args.spp, G, g_slow, split))
                print(f"  G={G} samples, double-buffered (two contexts, two streams), slowest share: {pm:.2f} ms per launch incl. "
                      f"zeroing = {full_ms / pm / G:.1%} of linear", flush=True)
    r.close()
    r2.close()


if __name__ == "__main__":
    main()
