#!/usr/bin/env python3
"""A/B timing of render-kernel variants in ONE process (interleaved rounds, median and min).

Each variant is a set of CRT_* environment knobs read by crt_create (block size, scheduler on/off,
phase thresholds).  Workload: the bench scene at 1200x675, --spp samples, depth 50.
    python tools/tune.py --spp 100 --rounds 3 "CRT_SCHED=0" "CRT_SCHED=1" "CRT_SCHED=1 CRT_LEAF_NUM=2"
"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

import cuda_ray_tracer_amd as crt  # noqa: E402

KNOBS = ("CRT_BLOCK", "CRT_TILES", "CRT_SCHED", "CRT_LEAF_NUM", "CRT_SHADE_NUM", "CRT_LDS_LIMIT", "CRT_RUN_SHIFT", "CRT_PERSIST", "CRT_UNIT_SAMPLES", "CRT_TAIL_PASS", "CRT_TOP_PAIRS", "CRT_UNIT_FINE", "CRT_UNIT_FINE_PCT", "CRT_ARITH", "CRT_WALKERS", "CRT_REFILL_MIN", "CRT_OWNER_SLEEP", "CRT_WALKER_BACKOFF", "CRT_WALKER_PRIO", "CRT_WALKER_MIN_BUSY", "CRT_WALKER_PATIENCE")


def make(variant, args):
    for k in KNOBS:
        os.environ.pop(k, None)
    for kv in variant.split():
        k, v = kv.split("=")
        os.environ[k] = v
    r = crt.Renderer(0, 1)
    sc = crt.scene_random(18, args.half, args.ground)
    r.upload_scene(sc)
    r.build_bvh(17)
    r.set_camera(crt.camera(args.width, args.spp, lookfrom=tuple(args.lookfrom), focus_dist=args.focus))
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--spp", type=int, default=100)
    ap.add_argument("--width", type=int, default=1200)
    ap.add_argument("--half", type=int, default=11)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--census", action="store_true")
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--lookfrom", type=float, nargs=3, default=[13, 2, 3])
    ap.add_argument("--ground", type=float, default=1000.0)
    ap.add_argument("--focus", type=float, default=10.0)
    args = ap.parse_args()
    rs = [make(v, args) for v in args.variants]
    buf = torch.zeros(args.width * args.width * 3, dtype=torch.int64, device="cuda:0")
    times = [[] for _ in rs]
    rays = [0] * len(rs)
    ref = None
    for rnd in range(args.rounds + 1):
        for i, r in enumerate(rs):
            buf.zero_()
            r.render_accumulate(crt.CRT_WORLD_BVH, 0, args.spp, args.depth, buf.data_ptr(), 0)
            st = r.stats()
            if rnd:
                times[i].append(st.kernel_seconds * 1e3)
            rays[i] = st.rays
            if rnd == 0:
                chk = int(buf.sum())
                if ref is None:
                    ref = chk
                print(f"# {args.variants[i]!r}: rays {st.rays} checksum {chk} {'OK' if chk == ref else 'DIFFERS'}", flush=True)
    if args.census:
        import ctypes as C
        for i, r in enumerate(rs):
            r.set_count_stats(2)
            buf.zero_()
            r.render_accumulate(crt.CRT_WORLD_BVH, 0, args.spp, args.depth, buf.data_ptr(), 0)
            st = r.stats()
            c = (C.c_uint64 * 32)()
            crt.lib().crt_debug_counters(r._ctx, c)
            c = list(c)
            names = ["RUN", "LEAF", "SHADE"]
            msg = " ".join(f"{names[j]}: iters {c[4 + 2 * j]} lanes/iter {c[5 + 2 * j] / max(1, c[4 + 2 * j]):.1f}" for j in range(3))
            msg += f" RUN steps {c[11]} ({c[11] / st.rays:.3f}/ray) lanes per walking step {c[12] / max(1, c[11]):.1f};"
            msg += f" Philox loop iters {c[13]} ({c[13] / max(1, c[8]):.2f}/shade phase) lanes/iter {c[14] / max(1, c[13]):.1f};"
            msg += f" alive/iter {c[10] / max(1, c[4] + c[6] + c[8]):.1f}"
            if c[0]:  # SCHED=3 walkers: refill attempts, requests per attempt, polls of a wholly idle walker; watchdog
                msg += f" WALKER refills {c[0]} ({c[1] / c[0]:.2f} requests each) idle polls {c[2]} owner waits {c[3]} watchdog {c[29]};"
            r.set_count_stats(3)  # per-phase shader cycles only (no per-iteration atomics)
            buf.zero_()
            r.render_accumulate(crt.CRT_WORLD_BVH, 0, args.spp, args.depth, buf.data_ptr(), 0)
            st3 = r.stats()
            t = (C.c_uint64 * 32)()
            crt.lib().crt_debug_counters(r._ctx, t)
            tot = max(1, sum(t[16:22]))
            msg += f" cycles (kernel {st3.kernel_seconds * 1e3:.1f} ms): " + " ".join(
                f"{nm} {t[16 + j] / tot * 100:.1f}%" for j, nm in enumerate(["sched", "walk", "sphere", "shade1", "blocks", "shade3"]))
            if t[28]:
                nwv, t0 = t[28], (~t[24]) & (2**64 - 1)
                rel = lambda total: ((total - nwv * t0) % 2**64) / nwv / 100.0  # mean, microseconds after the first wave's start
                msg += (f" timeline us after first start: {nwv} waves, start mean {rel(t[27]):.0f}; queue empty mean {rel(t[22]):.0f} last {(t[23] - t0) / 100.0:.0f};"
                        f" exit mean {rel(t[25]):.0f} last {(t[26] - t0) / 100.0:.0f}")
                import numpy as np
                nwv = min(int(nwv), 8192)
                tl = (C.c_uint64 * (3 * nwv))()
                crt.lib().crt_debug_timeline(r._ctx, tl, nwv)
                tl = np.array(tl, dtype=np.uint64).reshape(nwv, 3).astype(np.int64)
                base = tl[:, 0].min()
                pct = lambda col: " ".join(f"{np.percentile((tl[:, col] - base) / 100.0, q):.0f}" for q in (0, 1, 10, 50, 90, 99, 100))
                msg += f" percentiles 0/1/10/50/90/99/100 us: queue empty {pct(1)}; exit {pct(2)}; exit - queue empty {' '.join(f'{np.percentile((tl[:, 2] - tl[:, 1]) / 100.0, q):.0f}' for q in (0, 10, 50, 90, 99, 100))}"
            print(f"# census {args.variants[i]!r}: kernel {st.kernel_seconds * 1e3:.2f} ms, I/ray {st.pair_visits / st.rays:.2f} S/ray {st.sphere_tests / st.rays:.2f}; {msg}")
            r.set_count_stats(False)
    for i, v in enumerate(args.variants):
        med, mn = statistics.median(times[i]), min(times[i])
        print(f"{v:48s} median {med:9.3f} ms  min {mn:9.3f} ms  {rays[i] / med / 1e3:9.1f} Mrays/s", flush=True)


if __name__ == "__main__":
    main()
