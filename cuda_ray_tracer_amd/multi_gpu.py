"""Sample-split rendering over the ranks of one node (BASELINE config 3).

The scene and BVH are replicated; rank g traces samples [g*spp/G, (g+1)*spp/G) of EVERY pixel --
the Philox counter carries the global sample index, so the set of paths does not depend on G --
into its own un-normalised sum buffer (int64, 2^-40 fixed point: crt_render_accumulate); one
reduce(SUM) to rank 0 (RCCL over xGMI on GPUs, gloo in the CPU tests) combines them and rank 0
resolves to fp32.  Integer sums are associative: the image is bit-identical for every G and every
reduction order.  There is no other collective.
"""
import torch
import torch.distributed as dist


def sample_range(spp, world_size, rank):
    """Half-open range of global sample indices of `rank`; ranges tile [0, spp) exactly."""
    return rank * spp // world_size, (rank + 1) * spp // world_size


def share_of_rank(spp, world_size, rank, split="samples"):
    """What `rank` traces of one image: split "samples" = its sample range of every tile (north_star),
    split "tiles" = every sample of the 8x8 tiles t = rank mod world_size (BASELINE.json configs[4])."""
    if split == "tiles":
        return {"first": 0, "nsamples": spp, "tile_stride": world_size, "tile_offset": rank}
    lo, hi = sample_range(spp, world_size, rank)
    return {"first": lo, "nsamples": hi - lo, "tile_stride": 1, "tile_offset": 0}


def resolve(sums, spp):
    """fp32 frame of fixed-point sums, the formula of crt_resolve: float(double(sum) * (2^-40 / spp))."""
    return (sums.to(torch.float64) * (2.0 ** -40 / spp)).to(torch.float32)


def reduce_sums(sums, host_staged=False):
    """reduce(SUM) of the int64 frame to rank 0.  host_staged: through a host copy -- the one-GPU rehearsal where several
    ranks share a device (RCCL refuses two ranks on one GPU) and the process group is gloo."""
    if not host_staged:
        dist.reduce(sums, dst=0, op=dist.ReduceOp.SUM)
        return
    host = sums.cpu()
    dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
    if dist.get_rank() == 0:
        sums.copy_(host)


def render_split(trace_range, sums, spp, world_size, rank, always_reduce=False, split="samples", host_staged=False):
    """One image: zero `sums`, trace this rank's share into it with `trace_range(first, n, tile_stride, tile_offset)`
    (the arguments of crt_render_accumulate_tiles; the callback adds to `sums`), reduce to rank 0.  `sums` is a (H, W, 3) int64 tensor on the rank's device; after the call rank
    0 holds the sums of all `spp` samples (resolve them with crt_resolve / resolve()).
    `always_reduce`: issue the collective in a one-rank group too (the one-GPU rehearsal of the
    N > 1 path: tests/test_gpu_rccl.py)."""
    sh = share_of_rank(spp, world_size, rank, split)
    sums.zero_()
    if sh["nsamples"] > 0:
        trace_range(sh["first"], sh["nsamples"], sh["tile_stride"], sh["tile_offset"])
    if world_size > 1 or always_reduce:
        reduce_sums(sums, host_staged)
    return sums
