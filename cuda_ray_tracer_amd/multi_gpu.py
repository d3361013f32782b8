"""Sample-split rendering over the ranks of one node (BASELINE config 3).

The scene and BVH are replicated; rank g traces samples [g*spp/G, (g+1)*spp/G) of EVERY pixel --
the Philox counter carries the global sample index, so the set of paths does not depend on G --
into its own un-normalised fp32 sum buffer; one reduce(SUM) to rank 0 (RCCL over xGMI on GPUs,
gloo in the CPU tests) combines them and rank 0 scales by 1/spp.  There is no other collective.
"""
import torch
import torch.distributed as dist


def sample_range(spp, world_size, rank):
    """Half-open range of global sample indices of `rank`; ranges tile [0, spp) exactly."""
    return rank * spp // world_size, (rank + 1) * spp // world_size


def render_split(trace_range, sums, spp, world_size, rank, scale=True):
    """One image: zero `sums`, trace this rank's samples into it with `trace_range(lo, n, sums)`,
    reduce to rank 0, normalise there.  `sums` is a (H, W, 3) fp32 tensor on the rank's device."""
    lo, hi = sample_range(spp, world_size, rank)
    sums.zero_()
    if hi > lo:
        trace_range(lo, hi - lo, sums)
    if world_size > 1:
        dist.reduce(sums, dst=0, op=dist.ReduceOp.SUM)
    if rank == 0 and scale:
        sums.mul_(float(torch.tensor(1.0 / spp, dtype=torch.float64).to(torch.float32)))
    return sums
