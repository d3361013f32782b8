"""crt_render_static -- the entry point that allocates nothing (callers under HIP graph capture, tight serving loops,
harnesses that forbid hipMalloc): its frame against crt_render's, bit for bit, through the C ABI, on consecutive
calls of shrinking size; statistics; free device memory unchanged; the size limit.  Named to be collected last."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits_equal(a, b):
    return a.shape == b.shape and np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def test_static_entry_point_matches_crt_render(crt):
    import torch
    sc = crt.scene_random(18, 11)
    r = crt.Renderer(0, 3)
    r.upload_scene(sc)
    r.build_bvh(17)
    stream = torch.cuda.current_stream().cuda_stream
    # (the first launch of a kernel loads its code object, which the runtime allocates for: warm both worlds up first,
    # as a caller that must not allocate would)
    r.set_camera(crt.camera(64, 1))
    r.render(crt.CRT_WORLD_BVH, 1, 0, 5)
    r.render(crt.CRT_WORLD_LIST, 1, 0, 5)
    warm = torch.empty((36, 64, 3), dtype=torch.float32, device="cuda:0")
    r.render_static(crt.CRT_WORLD_BVH, 1, 0, 5, warm.data_ptr(), stream)  # first call on this GPU: loads the static array
    torch.cuda.synchronize()
    # first call: C2 geometry; second call: a smaller frame, fewer samples, the other world, another offset -- the
    # scratch still holds the first frame's sums and counters and must not leak into the second
    for W, spp, s0, depth, world in ((1200, 9, 0, 50, crt.CRT_WORLD_BVH), (200, 4, 7, 10, crt.CRT_WORLD_LIST),
                                     (640, 5, 2, 50, crt.CRT_WORLD_BVH)):
        cam = crt.camera(W, spp)
        r.set_camera(cam)
        H = cam.image_height
        out = torch.full((H, W, 3), -1.0, dtype=torch.float32, device="cuda:0")
        torch.cuda.synchronize()
        before = torch.cuda.mem_get_info()[0]  # free device memory: sees hipMalloc by anyone in the process
        r.render_static(world, spp, s0, depth, out.data_ptr(), stream)
        st = r.stats()
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == before
        want, st0 = r.render(world, spp, s0, depth)
        assert _bits_equal(out.cpu().numpy(), want)
        assert st.rays == st0.rays and st.samples == st0.samples == W * H * spp
    # the limit: frames beyond CRT_STATIC_MAX_PIXELS (2^21) are rejected with a clear error; 1920x1080 still fits
    r.set_camera(crt.camera(1920, 1))
    out = torch.empty((1080, 1920, 3), dtype=torch.float32, device="cuda:0")
    r.render_static(crt.CRT_WORLD_BVH, 1, 0, 50, out.data_ptr(), stream)
    want, _ = r.render(crt.CRT_WORLD_BVH, 1, 0, 50)
    torch.cuda.synchronize()
    assert _bits_equal(out.cpu().numpy(), want)
    r.set_camera(crt.camera(2048, 1))  # 2048 x 1152 = 2 359 296 pixels
    with pytest.raises(crt.CrtError, match="exceeds the static scratch"):
        r.render_static(crt.CRT_WORLD_BVH, 1, 0, 50, out.data_ptr(), stream)
    r.close()
    # a second context on the same GPU shares that GPU's array (calls ordered on one stream)
    r2 = crt.Renderer(0, 3)
    r2.upload_scene(sc)
    r2.build_bvh(17)
    cam = crt.camera(320, 3)
    r2.set_camera(cam)
    out = torch.empty((cam.image_height, 320, 3), dtype=torch.float32, device="cuda:0")
    r2.render_static(crt.CRT_WORLD_BVH, 3, 0, 50, out.data_ptr(), stream)
    want, _ = r2.render(crt.CRT_WORLD_BVH, 3, 0, 50)
    torch.cuda.synchronize()
    assert _bits_equal(out.cpu().numpy(), want)
    r2.close()


def test_static_entry_point_under_graph_capture(crt):
    """The caller the entry point exists for: one crt_render_static call captured into a HIP graph (stream capture
    forbids allocation), replayed three times; every replay must produce crt_render's frame.  (crt_render could not be
    captured: it sizes its buffers with hipMalloc and copies to the host synchronously.)"""
    import torch
    r = crt.Renderer(0, 3)
    r.upload_scene(crt.scene_random(18, 11))
    r.build_bvh(17)
    cam = crt.camera(640, 6)
    r.set_camera(cam)
    want, _ = r.render(crt.CRT_WORLD_BVH, 6, 0, 50)
    out = torch.zeros((cam.image_height, 640, 3), dtype=torch.float32, device="cuda:0")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):  # warm-up outside the capture: code objects, the scratch array's address
        r.render_static(crt.CRT_WORLD_BVH, 6, 0, 50, out.data_ptr(), s.cuda_stream)
    s.synchronize()
    out.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        r.render_static(crt.CRT_WORLD_BVH, 6, 0, 50, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert float(out.abs().sum()) == 0.0  # capture records, it does not run
    _, st_want = r.render(crt.CRT_WORLD_BVH, 6, 0, 50)
    for _ in range(3):
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert _bits_equal(out.cpu().numpy(), want)
    # the captured call recorded no timing events (they would have become graph nodes): statistics still work --
    # the counters of the last replay, kernel_seconds = 0 -- and so does the ring of kernel times
    r.render_static(crt.CRT_WORLD_BVH, 6, 0, 50, out.data_ptr(), s.cuda_stream)   # timed, un-captured
    s.synchronize()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=s):
        r.render_static(crt.CRT_WORLD_BVH, 6, 0, 50, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    g2.replay()
    torch.cuda.synchronize()
    st = r.stats()
    assert st.kernel_seconds == 0.0 and st.rays == st_want.rays and st.samples == st_want.samples
    kt = r.kernel_times(2)
    assert kt[0] > 0.0 and kt[1] == 0.0
    # a rejected call enqueues nothing: under capture it must add no node (here: unknown world kind)
    g3 = torch.cuda.CUDAGraph()
    out.zero_()
    with torch.cuda.graph(g3, stream=s):
        with pytest.raises(crt.CrtError):
            r.render_static(7, 6, 0, 50, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        out.add_(1.0)  # so that the graph is not empty
    g3.replay()
    torch.cuda.synchronize()
    assert float(out.min()) == 1.0 and float(out.max()) == 1.0
    del g, g2, g3
    r.close()
