#!/usr/bin/env python3
"""Where does context creation go?  Times crt_create (CRT_TRACE_INIT=1 prints its laps) and the first
launch (code object load) in a fresh process, with and without torch having initialised HIP first.
  python tools/ctx_timing.py [--no-torch-init]"""
import os
import sys
import time

os.environ["CRT_TRACE_INIT"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
t0 = time.perf_counter()
import torch  # noqa: E402
t_imp = time.perf_counter() - t0
if "--no-torch-init" not in sys.argv:
    t0 = time.perf_counter()
    torch.cuda.set_device(0)
    t_set = time.perf_counter() - t0
    t0 = time.perf_counter()
    x = torch.zeros(16, device="cuda")
    torch.cuda.synchronize()
    print(f"import torch {t_imp*1e3:.1f} ms, set_device {t_set*1e3:.1f} ms, first torch alloc+sync {(time.perf_counter()-t0)*1e3:.1f} ms", file=sys.stderr)
import cuda_ray_tracer_amd as crt  # noqa: E402
t0 = time.perf_counter()
crt.lib()
print(f"dlopen libcrt_hip.so {(time.perf_counter()-t0)*1e3:.1f} ms", file=sys.stderr)
t0 = time.perf_counter()
r = crt.Renderer(0, 1)
print(f"crt_create total {(time.perf_counter()-t0)*1e3:.1f} ms", file=sys.stderr)
t0 = time.perf_counter()
r2 = crt.Renderer(0, 1)
print(f"second crt_create {(time.perf_counter()-t0)*1e3:.1f} ms", file=sys.stderr)
sc = crt.scene_random(18, 11)
t0 = time.perf_counter()
r.upload_scene(sc)
print(f"upload_scene {(time.perf_counter()-t0)*1e3:.1f} ms", file=sys.stderr)
t0 = time.perf_counter()
r.build_bvh(17)
print(f"build_bvh (first: loads crt_bvh code object) {(time.perf_counter()-t0)*1e3:.1f} ms", file=sys.stderr)
t0 = time.perf_counter()
r.build_bvh(17)
print(f"build_bvh (second) {(time.perf_counter()-t0)*1e3:.1f} ms", file=sys.stderr)
r.set_camera(crt.camera(256, 4))
t0 = time.perf_counter()
r.render(crt.CRT_WORLD_BVH, 4)
print(f"first render 256x144x4 (loads render code object) {(time.perf_counter()-t0)*1e3:.1f} ms", file=sys.stderr)
t0 = time.perf_counter()
r.render(crt.CRT_WORLD_BVH, 4)
print(f"second render {(time.perf_counter()-t0)*1e3:.1f} ms", file=sys.stderr)
