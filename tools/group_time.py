#!/usr/bin/env python3
"""crt_group_render with 1 / 2 / 4 / 8 members that share GPU 0 (C2, 500 spp), both splits: the overhead of the
device-list handle itself (zeroing, frame copies, add kernel) on one chip.  python tools/group_time.py"""
import os
import sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cuda_ray_tracer_amd as crt
sc = crt.scene_random(18, 11)
cam = crt.camera(1200, 500)
for devs in ([0], [0, 0], [0, 0, 0, 0], [0] * 8):
    for split in (crt.CRT_SPLIT_SAMPLES, crt.CRT_SPLIT_TILES):
        g = crt.Group(devs, 1)
        g.upload_scene(sc); g.build_bvh(17); g.set_camera(cam)
        g.render(crt.CRT_WORLD_BVH, 500, 0, 50, split)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); fb, st = g.render(crt.CRT_WORLD_BVH, 500, 0, 50, split); ts.append(time.perf_counter() - t0)
        print(f"members {len(devs)} split {'tiles' if split else 'samples'}: wall {min(ts)*1e3:.2f} ms, slowest member kernel {st.kernel_seconds*1e3:.2f} ms, {st.rays/min(ts)/1e6:.0f} Mrays/s", flush=True)
        g.close()
