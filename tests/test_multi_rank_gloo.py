"""The N>1 path on CPU: gloo ranks split the samples exactly as bench.py does on GPUs
(cuda_ray_tracer_amd.multi_gpu), with the oracle standing in for the device kernel; the reduced
fixed-point buffer must equal the single-rank buffer bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world_size, port, spp, out_path, split="samples"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    import oracle_lib as orc
    from cuda_ray_tracer_amd.multi_gpu import render_split
    sc = orc.scene_small()
    cam = orc.camera(48, spp)
    sums = torch.zeros(cam.image_height, cam.image_width, 3, dtype=torch.int64)

    def trace_range(first, count, tile_stride, tile_offset):
        part, _ = orc.render(sc, cam, 7, first, count, 10)
        # the tile share of crt_render_accumulate_tiles: 8x8 tiles t = tile_offset mod tile_stride, row-major
        H, W = part.shape[:2]
        ty, tx = np.meshgrid(np.arange(H) // 8, np.arange(W) // 8, indexing="ij")
        mine = (ty * ((W + 7) // 8) + tx) % tile_stride == tile_offset
        sums.add_(torch.from_numpy(part * mine[:, :, None]))

    render_split(trace_range, sums, spp, world_size, rank, split=split)
    if rank == 0:
        np.save(out_path, sums.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world_size,spp,split", [(2, 9, "samples"), (3, 8, "samples"), (2, 1, "samples"), (3, 4, "tiles")])
def test_sample_split_reduce(tmp_path, world_size, spp, split):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as orc
    from cuda_ray_tracer_amd.multi_gpu import sample_range
    ranges = [sample_range(spp, world_size, r) for r in range(world_size)]
    assert ranges[0][0] == 0 and ranges[-1][1] == spp
    assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world_size - 1))
    out = str(tmp_path / "img.npy")
    port = 29500 + os.getpid() % 2000 + world_size
    mp.spawn(_worker, args=(world_size, port, spp, out, split), nprocs=world_size, join=True)
    got = np.load(out)
    sc, cam = orc.scene_small(), orc.camera(48, spp)
    whole, _ = orc.render(sc, cam, 7, 0, spp, 10)
    # integer sums: the reduced buffer equals the single-rank buffer exactly, for any rank count
    assert got.dtype == np.int64 and np.array_equal(got, whole)
    import torch
    from cuda_ray_tracer_amd.multi_gpu import resolve
    assert np.array_equal(resolve(torch.from_numpy(got), spp).numpy().view(np.uint32), orc.resolve(whole, spp).view(np.uint32))
