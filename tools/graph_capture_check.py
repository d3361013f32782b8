#!/usr/bin/env python3
"""crt_render_static under HIP graph capture (the caller it exists for): capture one call into a graph on a torch stream,
replay the graph, compare the frame with crt_render's.  python tools/graph_capture_check.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import cuda_ray_tracer_amd as crt  # noqa: E402

r = crt.Renderer(0, 3)
r.upload_scene(crt.scene_random(18, 11))
r.build_bvh(17)
cam = crt.camera(640, 6)
r.set_camera(cam)
want, st0 = r.render(crt.CRT_WORLD_BVH, 6, 0, 50)
out = torch.zeros((cam.image_height, 640, 3), dtype=torch.float32, device="cuda:0")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    r.render_static(crt.CRT_WORLD_BVH, 6, 0, 50, out.data_ptr(), s.cuda_stream)  # warm-up: loads code objects, resolves the scratch
s.synchronize()
out.zero_()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    r.render_static(crt.CRT_WORLD_BVH, 6, 0, 50, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("captured; frame untouched during capture:", float(out.abs().sum()) == 0.0)
for _ in range(3):
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    same = np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))
    print("replay: frame equals crt_render's:", same)
    assert same
print("OK")
