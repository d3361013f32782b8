#!/usr/bin/env python3
"""Soak: 60 rounds of create / render / render_static / destroy of a context and of a 3-member group on one GPU; frames
must stay bit-identical and free device memory must not drift.  python tools/soak.py"""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
import cuda_ray_tracer_amd as crt
sc = crt.scene_random(18, 11)
cam = crt.camera(320, 4)
free0 = None
ref = None
for it in range(60):
    r = crt.Renderer(0, 1); r.upload_scene(sc); r.build_bvh(17); r.set_camera(cam)
    fb, st = r.render(crt.CRT_WORLD_BVH, 4, 0, 50)
    out = torch.empty((cam.image_height, 320, 3), dtype=torch.float32, device="cuda:0")
    r.render_static(crt.CRT_WORLD_BVH, 4, 0, 50, out.data_ptr(), 0)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint32), fb.view(np.uint32))
    r.close()
    g = crt.Group([0, 0, 0], 1); g.upload_scene(sc); g.build_bvh(17); g.set_camera(cam)
    fb2, _ = g.render(crt.CRT_WORLD_BVH, 4, 0, 50, it % 2)
    g.close()
    assert np.array_equal(fb2.view(np.uint32), fb.view(np.uint32))
    if ref is None: ref = fb
    assert np.array_equal(ref.view(np.uint32), fb.view(np.uint32))
    del out
    torch.cuda.empty_cache()
    free = torch.cuda.mem_get_info()[0]
    if it == 5: free0 = free
print("ok; free memory drift since iteration 5:", (free0 - free) / 1e6, "MB")
