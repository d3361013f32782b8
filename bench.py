#!/usr/bin/env python3
"""bench.py -- the reference's headline workload on N MI355X of one node.

Workload (BASELINE.json configs[1]): the RTOW random-spheres scene (helper.h:37-102, ~486 spheres,
1023-node flat BVH) at 1200x675, 500 spp, depth 50.  One STEP = one complete image: every rank
traces its share of the 500 samples of every pixel (sample split, scene replicated), the
un-normalised fixed-point (int64) sums are combined with one RCCL reduce to rank 0, rank 0 resolves
them to the fp32 frame.
Total work per step is fixed, so scaling over N is "strong".

Metric: Mrays/s, whole job; one ray = one closest-hit query of the bounce loop
(render_with_cuda.h:41), counted on the device.

  python bench.py --gpus N --steps K --warmup W          # N > 1: starts its own N ranks
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import contextlib
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VALU_FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 vector (= the f32 MFMA dense peak)
HBM_PEAK_GBS = 8000.0
L1_RECORD_RATE_PEAK_GS = 290.0  # tools/ubench/gather_quad.hip: dependent 64-byte record reads, chip-wide, L1-resident table


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1200)
    ap.add_argument("--spp", type=int, default=500)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--scene-seed", type=int, default=18)
    ap.add_argument("--cpu-seconds", type=float, default=10.0,
                    help="target CPU time of the cpu_baseline's three samples together (0 = skip)")
    ap.add_argument("--config", choices=["c1", "c2", "c4", "c5"], default="c2",
                    help="c2 = BASELINE configs[1] (the metric's config, default); c1 = configs[0], the 5-sphere scene through "
                         "the brute-force list world (400x225, 10 spp, depth 10); c4 = configs[3], the synthetic 1M-sphere "
                         "HBM stress (1920x1080, 64 spp); c5 = configs[4], the random-spheres scene at 3840x2160, 1024 spp, "
                         "rendered progressively in --passes launches per image -- c1/c4/c5 are DESIGN.md measurements, not "
                         "the driver's line")
    ap.add_argument("--passes", type=int, default=0,
                    help="progressive accumulation: every rank cuts its share of an image into this many launches into the "
                         "same sums buffer (default 1; 8 for --config c5); one reduce + resolve per image")
    ap.add_argument("--split", choices=["samples", "tiles"], default="samples",
                    help="N > 1: rank g traces samples [g spp/N, (g+1) spp/N) of every pixel (default, north_star), or "
                         "all samples of the 8x8 tiles t = g mod N")
    ap.add_argument("--arith", choices=["exact", "fast"], default="exact",
                    help="arithmetic of the TIMED steps: exact (default; every fp32 operation rounded separately, frames "
                         "bit-identical to the CPU oracle) or fast (fma slab test + contraction; within the tolerance of "
                         "tests/test_gpu_round2.py::test_fast_arithmetic_within_stated_tolerance).  The other mode is measured "
                         "beside it, untimed by the step clock (HIP events)")
    ap.add_argument("--pipeline", choices=["auto", "on", "off"], default="auto",
                    help="pipelined steps: three contexts / sums buffers / stream pairs per rank, so that the reduce + "
                         "resolve of image k run behind the trace of image k+1 and the next launch's workgroups start in the "
                         "drain tail of the previous one (DESIGN.md 5).  auto = on for N > 1 (and the one-GPU rehearsals of "
                         "the N-rank path), off for N = 1: the driver's single-GPU line renders its steps strictly one after "
                         "the other.  time_to_image_ms is always the un-pipelined figure")
    ap.add_argument("--verify-steps", action="store_true",
                    help="read every timed step's frame back and report its SHA-256 (step_frame_sha256; tests/test_gpu_rccl.py) "
                         "-- the read-backs synchronise, so the timings of such a run mean nothing")
    ap.add_argument("--force-dist", action="store_true",
                    help="with --gpus 1: still create the RCCL process group and issue every collective of the "
                         "N > 1 path (one-GPU rehearsal, tests/test_gpu_rccl.py)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="one-GPU rehearsal of the N-rank HIP path (tests/test_gpu_rccl.py): every rank renders its share on "
                         "GPU 0 with the real kernels, the int64 frames are reduced through host copies over gloo (RCCL "
                         "refuses two ranks on one device); a correctness rehearsal, its timings mean nothing")
    ap.add_argument("--rehearse-cpu", action="store_true",
                    help="control-flow rehearsal without a GPU (tests/test_bench_launcher.py): gloo ranks, CPU tensors and "
                         "a stand-in tracer that adds 2^40 per sample; prints a line marked \"data\": \"rehearsal\"")
    return ap.parse_args(argv)


# ---- N > 1 from a plain `python bench.py --gpus N`: start the ranks as fresh child processes ----------

def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_launch_command(n, argv, port):
    """The child command of `python bench.py --gpus n ...` (argv = the parent's own arguments)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), *argv]


def launch_ranks(n, argv):
    """Runs the N ranks as children and relays their output; the parent never touches the GPU (a process that
    has initialised HIP must not exec or fork workers).  Returns the launcher's exit code (non-zero when any
    rank failed)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    proc = subprocess.Popen(rank_launch_command(n, argv, free_port()), env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


# ---- CPU baseline -----------------------------------------------------------------------------------

def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, seed):
    """The oracle (a scalar port of the reference loop) on this box's host cores, on bounded samples of the
    workload: THREE samples of about --cpu-seconds / 3 each on the box's CPU share (OMP_NUM_THREADS, 16 per GPU on
    the pool) -- `value` is their MEDIAN, min and max are reported beside it (one 10-second sample moved by 40 %
    between runs of the same box) -- plus one sample on every hardware thread the box reports and one on one thread."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as orc
    host_cores = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = host_cores
    share = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, host_cores)
    share = max(1, min(share, host_cores))
    all_threads = max(1, min(affinity, host_cores, 512))
    sc = orc.scene_random(args.scene_seed, 11)
    nodes, _ = orc.bvh_build(sc, 17)
    cam = orc.camera(args.width, args.spp)
    per_sample = args.cpu_seconds / 3.0

    def sample(threads, first, spp=None):
        if spp is None:  # size the sample from one 1-spp pass
            t0 = time.perf_counter()
            orc.render(sc, cam, seed, 0, 1, args.depth, nodes, nthreads=threads)
            t1 = time.perf_counter() - t0
            spp = int(max(1, min(128, per_sample / max(t1, 1e-3))))
        t0 = time.perf_counter()
        _, st = orc.render(sc, cam, seed, first, spp, args.depth, nodes, nthreads=threads)
        dt = time.perf_counter() - t0
        return {"threads": threads, "mrays_s": st[0] / dt / 1e6, "spp": spp, "first_sample": first, "rays": st[0], "seconds": dt}

    runs = [sample(share, 1)]
    for i in (1, 2):  # different sample ranges of the same frame, same size
        runs.append(sample(share, 1 + i * runs[0]["spp"], runs[0]["spp"]))
    rates = sorted(r["mrays_s"] for r in runs)
    median = rates[1]
    mid = next(r for r in runs if r["mrays_s"] == median)
    wide = sample(all_threads, 1) if all_threads != share else None
    t0 = time.perf_counter()
    y0 = max(0, min(300, cam.image_height - 64))  # a 64-row band (rows 300..363 of the 675 at the default size)
    _, st1 = orc.render(sc, cam, seed, 0, 1, args.depth, nodes, region=(0, y0, cam.image_width, min(cam.image_height, y0 + 64)), nthreads=1)
    dt1 = time.perf_counter() - t0
    out = {"value": median, "unit": "Mrays/s", "cores": share, "kind": "port",
           "min": rates[0], "median": median, "max": rates[2],
           "sample": f"3 samples of the full {cam.image_width}x{cam.image_height} frame, {mid['spp']} of {args.spp} spp each "
                     f"(different sample ranges), depth {args.depth}, BVH world, on {share} threads; median sample: {mid['rays']} rays "
                     f"in {mid['seconds']:.2f} s (oracle/rt_oracle.cpp, g++ -O2); value = median",
           "host_cores": host_cores, "affinity_cores": affinity, "cpu_model": cpu_model(),
           "runs": runs, "single_thread_mrays_s": st1[0] / dt1 / 1e6}
    if wide:
        out["all_hardware_threads"] = wide
    return out


# ---- one rank ---------------------------------------------------------------------------------------

class GpuRank:
    """This rank's GPU: context, scene, BVH, camera, buffers (torch is plumbing: device memory, stream, RCCL)."""

    def __init__(self, args, local_rank, scene_kw, cam_kw):
        import torch
        import cuda_ray_tracer_amd as crt
        self.torch, self.crt = torch, crt
        n_dev = torch.cuda.device_count()  # does not initialise the GPU on this image
        if args.share_gpu:
            local_rank = 0
        if local_rank >= n_dev:
            raise SystemExit(f"bench.py: {args.gpus} devices requested, {n_dev} visible (rank {local_rank} has no GPU)")
        torch.cuda.set_device(local_rank)
        self.dev = torch.device("cuda", local_rank)
        self.world = crt.CRT_WORLD_LIST if args.config == "c1" else crt.CRT_WORLD_BVH
        t0 = time.perf_counter()
        # first HIP work of the process: device discovery, queue creation, the runtime's own blit kernels -- HIP
        # runtime start-up, not library code (tools/ctx_timing.py: crt_create itself takes 0.1 ms once HIP is up)
        torch.zeros(1, device=self.dev)
        torch.cuda.synchronize()
        self.t_hip = time.perf_counter() - t0
        self.r = crt.Renderer(local_rank, args.seed)
        self.r.set_arith(crt.CRT_ARITH_FAST if args.arith == "fast" else crt.CRT_ARITH_EXACT)
        self.t_ctx = time.perf_counter() - t0 - self.t_hip   # crt_create: library state on an initialised runtime
        self.scene = crt.scene_small() if args.config == "c1" else crt.scene_random(args.scene_seed, **scene_kw)
        self.t_scene = time.perf_counter() - t0 - self.t_ctx - self.t_hip
        self.r.upload_scene(self.scene)
        t1 = time.perf_counter()
        self.num_nodes = self.r.build_bvh(17)   # GPU builder (crt_bvh.hip); synchronous
        self.t_bvh = time.perf_counter() - t1
        self.cam = crt.camera(args.width, args.spp, **cam_kw)
        self.r.set_camera(self.cam)
        torch.cuda.synchronize()
        self.t_setup = time.perf_counter() - t0
        self.W, self.H = self.cam.image_width, self.cam.image_height
        self.sums = torch.zeros(self.H, self.W, 3, dtype=torch.int64, device=self.dev)   # fixed-point (2^-40) RGB sums
        self.frame = torch.empty(self.H, self.W, 3, dtype=torch.float32, device=self.dev)
        self.stream = torch.cuda.current_stream().cuda_stream
        self.depth = args.depth
        self.backend = "gloo" if args.share_gpu else "nccl"
        self.args = args
        self.slots = []

    def trace(self, first, count, tile_stride=1, tile_offset=0):
        self.r.render_accumulate(self.world, first, count, self.depth, self.sums.data_ptr(), self.stream,
                                 tile_stride=tile_stride, tile_offset=tile_offset)

    def resolve(self, spp):
        self.r.resolve(self.sums.data_ptr(), spp, self.frame.data_ptr(), self.stream)

    def make_slots(self, count=3):
        """`count` (context, sums, frame, trace stream, combine stream) sets for pipelined steps: launches that overlap
        on the GPU need a context each (include/crt_hip.h), and each set works on streams of its own."""
        torch, crt = self.torch, self.crt
        for i in range(count):
            slot = GpuSlot()
            if i == 0:
                slot.r, slot.sums, slot.frame = self.r, self.sums, self.frame
            else:
                slot.r = crt.Renderer(self.dev.index, self.args.seed)
                slot.r.set_arith(crt.CRT_ARITH_FAST if self.args.arith == "fast" else crt.CRT_ARITH_EXACT)
                slot.r.upload_scene(self.scene)
                slot.r.build_bvh(17)
                slot.r.set_camera(self.cam)
                slot.sums = torch.zeros_like(self.sums)
                slot.frame = torch.empty_like(self.frame)
            slot.tstream = torch.cuda.Stream(device=self.dev)   # zeroing + trace
            slot.cstream = torch.cuda.Stream(device=self.dev)   # reduce + resolve
            slot.stream = slot.tstream.cuda_stream
            slot.traced, slot.free = torch.cuda.Event(), None
            slot.world, slot.depth = self.world, self.depth
            self.slots.append(slot)
        torch.cuda.synchronize()
        return self.slots

    def sync(self):
        self.torch.cuda.synchronize()

    def close(self):
        for slot in self.slots[1:]:
            slot.r.close()
        self.r.close()


class GpuSlot:
    def trace(self, first, count, tile_stride=1, tile_offset=0):
        self.r.render_accumulate(self.world, first, count, self.depth, self.sums.data_ptr(), self.stream,
                                 tile_stride=tile_stride, tile_offset=tile_offset)

    def resolve(self, spp):  # on the slot's combine stream
        self.r.resolve(self.sums.data_ptr(), spp, self.frame.data_ptr(), self.cstream.cuda_stream)


class RehearsalRank:
    """No GPU: the same control flow on CPU tensors with a stand-in tracer (every sample adds 2^40 to every
    component of its pixel, so a complete image resolves to exactly 1.0 whatever the split)."""

    def __init__(self, args, local_rank, scene_kw, cam_kw):
        import torch
        self.torch = torch
        self.dev = torch.device("cpu")
        self.W, self.H = args.width, max(1, int(args.width / (16.0 / 9.0)))
        self.sums = torch.zeros(self.H, self.W, 3, dtype=torch.int64)
        self.frame = torch.empty(self.H, self.W, 3, dtype=torch.float32)
        self.t_hip = self.t_ctx = self.t_scene = self.t_bvh = self.t_setup = 0.0
        self.num_nodes, self.backend = 0, "gloo"
        self.last = (0, 0)

    def trace(self, first, count, tile_stride=1, tile_offset=0):
        tiles_x = (self.W + 7) // 8
        ty, tx = self.torch.meshgrid(self.torch.arange(self.H) // 8, self.torch.arange(self.W) // 8, indexing="ij")
        mine = ((ty * tiles_x + tx) % tile_stride) == tile_offset
        self.sums += (mine.to(self.torch.int64) * (count << 40))[:, :, None]
        self.last = (int(mine.sum()) * count * 2, int(mine.sum()) * count)  # "rays" = 2 per sample

    def resolve(self, spp):
        self.frame.copy_((self.sums.to(self.torch.float64) * (2.0 ** -40 / spp)).to(self.torch.float32))

    def make_slots(self, count=3):
        """CPU stand-in of GpuRank.make_slots: `count` sets of buffers, no streams."""
        import copy
        self.tstream = None
        self.slots = [self]
        for _ in range(count - 1):
            other = copy.copy(self)
            other.sums, other.frame = self.torch.zeros_like(self.sums), self.torch.empty_like(self.frame)
            self.slots.append(other)
        return self.slots

    def sync(self):
        pass

    def close(self):
        pass


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the driver's plain `python bench.py --gpus N`: no torch.cuda / HIP call has happened in this process
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    import numpy as np
    import torch
    import torch.distributed as dist
    from cuda_ray_tracer_amd.multi_gpu import reduce_sums, render_split, share_of_rank

    scene_kw, cam_kw = dict(half=11, ground_radius=1000.0), {}
    if args.config == "c4":  # the same generator on a 1000x1000 grid, camera pulled back 40x
        scene_kw = dict(half=500, ground_radius=1.0e5)
        cam_kw = dict(lookfrom=(520.0, 80.0, 120.0), focus_dist=540.0)
        if args.width == 1200 and args.spp == 500:
            args.width, args.spp = 1920, 64
    if args.config == "c1" and args.width == 1200 and args.spp == 500 and args.depth == 50:
        args.width, args.spp, args.depth = 400, 10, 10
    if args.config == "c5" and args.width == 1200 and args.spp == 500:
        args.width, args.spp = 3840, 1024
    passes = args.passes if args.passes > 0 else (8 if args.config == "c5" else 1)
    args.passes = passes

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_size}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    g = (RehearsalRank if args.rehearse_cpu else GpuRank)(args, local_rank, scene_kw, cam_kw)
    use_dist = world_size > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        kw = {} if (args.rehearse_cpu or args.share_gpu) else {"device_id": g.dev}
        dist.init_process_group(g.backend, rank=rank, world_size=world_size, **kw)
    W, H = g.W, g.H

    # this rank's share of one image: a sample range of every pixel, or every sample of every N-th tile
    share = share_of_rank(args.spp, world_size, rank, args.split)
    has_work = share["nsamples"] > 0

    rays_of_passes = []

    def trace_in_passes(first, count, tile_stride, tile_offset, count_rays=False):
        # progressive accumulation (crt_render_accumulate ADDS): the share's samples in `passes` launches, same buffer
        for q in range(passes):
            lo, hi = first + q * count // passes, first + (q + 1) * count // passes
            if hi > lo:
                g.trace(lo, hi - lo, tile_stride, tile_offset)
                if count_rays:
                    st_q = g.r.stats()
                    rays_of_passes.append((st_q.rays, st_q.samples))

    def step():  # zero, trace this rank's share, RCCL reduce(SUM) to rank 0, resolve to fp32 there.  EVERY rank calls it.
        render_split(trace_in_passes if passes > 1 else g.trace, g.sums, args.spp, world_size, rank, always_reduce=use_dist,
                     split=args.split, host_staged=args.share_gpu)
        if rank == 0:
            g.resolve(args.spp)

    # Pipelined steps (N > 1): image k on slot k mod 3 -- a context, sums, frame, trace stream and combine stream of its
    # own.  Trace stream: zero the sums, trace this rank's share.  Combine stream, behind an event of the trace: the
    # reduce (torch orders the collective behind the CURRENT stream and the current stream behind the collective) and,
    # on rank 0, the resolve.  Image k+1 traces on ANOTHER slot's stream, so its workgroups start in the drain tail of
    # image k while image k is reduced and resolved; with three slots the trace of image k+2 waits for nothing of
    # image k either (with two it would queue behind k's resolve, which needs room on a chip that trace k+1 fills).
    # A slot is reused three images later, behind its own previous combine.  Collectives are issued in step order on
    # every rank.  The frames are the un-pipelined frames bit for bit (integer sums).
    pipelined = args.pipeline == "on" or (args.pipeline == "auto" and use_dist)
    if passes > 1:
        pipelined = False
    slots = g.make_slots(3) if pipelined else []
    step_hashes = []

    def step_pipelined(k):
        sl = slots[k % len(slots)]
        streams = getattr(sl, "tstream", None) is not None
        with (torch.cuda.stream(sl.tstream) if streams else contextlib.nullcontext()):
            if streams and sl.free is not None:
                sl.tstream.wait_event(sl.free)  # this slot's previous reduce + resolve have read its sums
            sl.sums.zero_()
            if has_work:
                sl.trace(share["first"], share["nsamples"], share["tile_stride"], share["tile_offset"])
            if streams:
                sl.traced.record(sl.tstream)
        with (torch.cuda.stream(sl.cstream) if streams else contextlib.nullcontext()):
            if streams:
                sl.cstream.wait_event(sl.traced)
            if use_dist:
                reduce_sums(sl.sums, args.share_gpu)
            if rank == 0:
                sl.resolve(args.spp)
                if args.verify_steps:
                    step_hashes.append(hashlib.sha256(sl.frame.cpu().numpy().tobytes()).hexdigest())
            if streams:
                sl.free = torch.cuda.Event()
                sl.free.record(sl.cstream)

    def sync():
        if use_dist:
            dist.barrier()
        g.sync()

    for k in range(args.warmup):
        step_pipelined(k) if pipelined else step()
    sync()
    step_hashes.clear()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step_pipelined(args.warmup + k) if pipelined else step()
        if args.verify_steps and not pipelined and rank == 0:
            step_hashes.append(hashlib.sha256(g.frame.cpu().numpy().tobytes()).hexdigest())
    g.sync()
    if use_dist:
        dist.barrier()
        g.sync()
    elapsed = time.perf_counter() - t0
    kernel_s = []
    if args.rehearse_cpu:
        rays_rank, samples_rank = (float(v) for v in max(sl.last for sl in (slots or [g])))
    else:
        st = g.r.stats() if has_work else None  # last launch: rays of one step on this rank (identical every step)
        rays_rank = float(st.rays) if st else 0.0
        samples_rank = float(st.samples) if st else 0.0
        # per-launch kernel time of the TIMED steps: HIP events the library records on the launch
        # stream around every launch (it keeps the last 64)
        if has_work and args.steps > 0 and pipelined:  # the timed launches are spread over the two contexts
            for i, sl in enumerate(slots):
                mine = len([k for k in range(args.warmup, args.warmup + args.steps) if k % len(slots) == i])
                if mine:
                    kernel_s += sl.r.kernel_times(min(mine, 64))
        elif has_work and args.steps > 0:
            kernel_s = g.r.kernel_times(min(args.steps * passes, 64))
        if has_work and passes > 1:  # rays of one image = the sum over its passes: one more, untimed, image with a read-back per pass
            g.sums.zero_()
            trace_in_passes(share["first"], share["nsamples"], share["tile_stride"], share["tile_offset"], count_rays=True)
            rays_rank = float(sum(v[0] for v in rays_of_passes))
            samples_rank = float(sum(v[1] for v in rays_of_passes))

    t = torch.tensor([elapsed, rays_rank, samples_rank], dtype=torch.float64, device="cpu" if args.share_gpu else g.dev)
    if use_dist:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
    rays_step, samples_step = float(t[1]), float(t[2])

    # time-to-image: set-up + one step + read-back of the finished frame.  The step holds a collective, so
    # every rank runs it; only rank 0 reads the frame back.
    sync()
    t1 = time.perf_counter()
    step()
    img = g.frame.cpu() if rank == 0 else None
    g.sync()
    t_image = g.t_setup + (time.perf_counter() - t1)
    sync()

    if rank == 0:
        ms_per_step = elapsed / max(args.steps, 1) * 1e3
        value = rays_step * args.steps / elapsed / 1e6
        out = {
            "metric": "Mrays/s", "value": value, "unit": "Mrays/s", "n_gpus": world_size, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "rehearsal" if args.rehearse_cpu else "synthetic",
            "config": {"workload": f"RTOW random-spheres scene (helper.h:37-102, Philox scene seed {args.scene_seed}), "
                                   f"{W}x{H}, {args.spp} spp, depth {args.depth}, flat BVH",
                       "width": W, "height": H, "spp": args.spp, "max_depth": args.depth,
                       "parallelism": (f"spp split x{world_size}" if args.split == "samples" else f"8x8-tile split x{world_size}")
                                      + " + RCCL reduce(sum) to rank 0"},
            "msamples_per_s": samples_step * args.steps / elapsed / 1e6,
            "rays_per_step": rays_step, "segments_per_sample": rays_step / max(samples_step, 1.0),
            "time_to_image_ms": t_image * 1e3, "setup_ms": g.t_setup * 1e3,
            "hip_runtime_init_ms": g.t_hip * 1e3, "context_ms": g.t_ctx * 1e3, "scene_generation_ms": g.t_scene * 1e3, "bvh_build_ms": g.t_bvh * 1e3,
            "image_mean": float(img.mean()),
            "frame_sha256": hashlib.sha256(img.numpy().tobytes()).hexdigest(),
            "pipelined": bool(pipelined),
        }
        if args.verify_steps:
            out["step_frame_sha256"] = step_hashes
        if not args.rehearse_cpu:
            gpu_report(args, g, out, share, kernel_s, world_size, np)
        print(json.dumps(out), flush=True)
    g.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def gpu_report(args, g, out, share, kernel_s, world_size, np):
    """roofline + cpu_baseline of the line (rank 0 only; no collective in here)."""
    crt, r = g.crt, g.r
    W, H = g.W, g.H
    build = crt.build_info()
    build["arith"] = args.arith
    out["arith"] = args.arith
    out["library_build_id"] = build["build_id"]
    out["config"].update({"spheres": g.scene.n, "bvh_nodes": g.num_nodes})
    kernel = r.kernel_name()  # of the last timed-path launch (the time-to-image step)
    # exact traversal work of this rank's share, from the counting kernel variant (untimed)
    r.set_count_stats(True)
    g.trace(share["first"], share["nsamples"], share["tile_stride"], share["tile_offset"])
    cst = r.stats()
    r.set_count_stats(False)
    I = cst.pair_visits / cst.rays
    S = cst.sphere_tests / cst.rays
    flops_per_ray = 50.0 * I + 45.0 * S + 60.0
    bytes_per_ray = 64.0 * I + 32.0 * S
    passes = args.passes
    kavg = float(np.mean(kernel_s)) * passes  # kernel time of one image's launches on this rank
    overlapped = bool(out.get("pipelined"))
    if overlapped:
        # pipelined steps: consecutive launches overlap on the chip (the next one starts in the drain of the previous
        # one), so the events around ONE launch also cover its wait for room and part of its neighbours; the time the
        # chip spends per image is the step time
        kavg = out["ms_per_step"] * 1e-3
    achieved_tflops = flops_per_ray * cst.rays / kavg / 1e12
    roofline = {
        "bound": "valu", "achieved": achieved_tflops, "peak": VALU_FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": achieved_tflops / VALU_FP32_PEAK_TFLOPS, "traffic": None,
        "kernel": kernel, "kernel_ms": kavg * 1e3, "kernel_ms_is_step_time": overlapped,
        "rays_per_launch": int(cst.rays), "pair_visits_per_ray": I, "sphere_tests_per_ray": S,
        "flops_per_ray": flops_per_ray, "bytes_per_ray": bytes_per_ray,
        "scene_fetch_GBs": bytes_per_ray * cst.rays / kavg / 1e9,
        "note": "not a dense contraction (no MFMA); the BVH (50 KB of sign-sorted box planes + leaf tags) is LDS-resident and the "
                "15 KB of spheres stay in L1/L2 (HBM traffic = the frame only): the bound is fp32 VALU issue under wave64 "
                "divergence; peak = 157.3 TFLOP/s fp32 vector (= the f32-input MFMA dense peak); flops_per_ray is the "
                "algorithmic count of SURVEY.md 8(d) (50 I + 45 S + 60), scene_fetch_GBs the algorithmic scene fetch rate "
                "(64 I + 32 S bytes per ray)",
    }
    if build["arith"] == "exact":
        # the exact arithmetic contract (-ffp-contract=off: the reference's separately rounded multiplies and adds)
        # rules out FMA, which is half of the 157.3 TFLOP/s: against the add/mul issue peak the fraction doubles
        roofline.update({"peak_without_fma": VALU_FP32_PEAK_TFLOPS / 2, "frac_without_fma": achieved_tflops / (VALU_FP32_PEAK_TFLOPS / 2)})
    # HBM traffic per launch: PMC counters cannot be read from inside this process; the figure is the rocprofv3
    # FETCH_SIZE + WRITE_SIZE measurement of the same launch committed under profiles/ (FETCH_SIZE doubled: on
    # gfx950 it tallies 128-byte requests at 64 bytes, MI355X_MICROARCH.md 'HBM').  The file records the library
    # build (hash of the kernel sources) and the kernel it was taken at: any other build or kernel reports null.
    try:
        pmc_all = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        pmc = pmc_all[args.config + ("_fast" if args.arith == "fast" else "")]
        same = (pmc.get("build_id") == build["build_id"] and pmc.get("kernel") == kernel and pmc.get("width") == W
                and pmc.get("spp") == args.spp and world_size == 1)
        if same:
            roofline["traffic"] = pmc["fetch_bytes_per_launch"] + pmc["write_bytes_per_launch"]
            roofline["traffic_source"] = f"committed profile {pmc.get('file')} (build {pmc.get('build_id')}, commit {pmc.get('commit')})"
            if "valu" in pmc:  # PMC view of the same kernel: what the SIMDs actually issue
                roofline["pmc"] = dict(pmc["valu"])
                for k in ("l1_accesses_per_clk_per_cu", "l1_hit", "l2_hit", "wave_cycles_waiting_on_memory", "fetch_bytes_raw"):
                    if k in pmc:  # C4: the memory-side counters of the same launch (profiles/<round>_pmc_summary.txt)
                        roofline["pmc"][k] = pmc[k]
        else:
            roofline["traffic_source"] = (f"null: profiles/pmc_traffic.json was taken at build {pmc.get('build_id')} / "
                                          f"{pmc.get('kernel')}, this run is build {build['build_id']} / {kernel}")
    except (OSError, KeyError, ValueError):
        pass
    out["roofline"] = roofline
    if args.config == "c1":
        out["config"]["workload"] = (f"RTOW small scene (helper.h:7-35, 5 spheres), {W}x{H}, {args.spp} spp, depth {args.depth}, "
                                     "brute-force hittable_list world")
    if passes > 1:
        out["config"]["passes"] = passes
        out["config"]["workload"] += f", progressive in {passes} launches per image"
    if args.config == "c4":
        out["config"]["workload"] = (f"synthetic {g.scene.n}-sphere scene (helper.h:37-102 generator on a 1000x1000 grid, "
                                     f"ground r=1e5), {W}x{H}, {args.spp} spp, depth {args.depth}, flat BVH in HBM")
        hbm_gbs = (roofline["traffic"] / kavg / 1e9) if roofline["traffic"] else None
        rec_gs = I * cst.rays / kavg / 1e9
        roofline.update({"bound": "l1-request-rate", "achieved": rec_gs, "peak": L1_RECORD_RATE_PEAK_GS, "unit": "Grecords/s",
                         "frac": rec_gs / L1_RECORD_RATE_PEAK_GS,
                         "fp32_TFLOPs": achieved_tflops, "frac_of_fp32_vector_peak": achieved_tflops / VALU_FP32_PEAK_TFLOPS,
                         "hbm_GBs_measured": hbm_gbs, "hbm_peak_GBs": HBM_PEAK_GBS,
                         "frac_of_hbm_peak_measured": (hbm_gbs / HBM_PEAK_GBS) if hbm_gbs else None,
                         "note": "scene (67 MB child-pair records + 32 MB spheres) is read by random access and fits the "
                                 "256 MB Infinity Cache: measured memory-side traffic (PMC, hbm_GBs_measured: fabric requests of "
                                 "the L2s, Infinity-Cache hits included, FETCH_SIZE doubled) is about a tenth of the algorithmic "
                                 "bytes (64 I + 32 S per ray = scene_fetch_GBs, which therefore exceeds the HBM peak): the "
                                 "instance is bound by the L1's access rate (PMC of this build, roofline.pmc: ~0.9 L1 accesses per "
                                 "clock per CU, L1 hit 76%, L2 hit 89%, 64% of wave cycles in s_waitcnt); achieved = I x rays / time (child-pair "
                                 "records per second, the upper ten levels come from LDS), peak = what "
                                 "tools/ubench/gather_quad.hip reaches with dependent 64-byte record reads from an "
                                 "L1-resident table and nothing else to do (250-290 G records/s chip-wide)",
                         "record_fetch_Gs": rec_gs, "record_rate_limit_Gs": [250, 290]})
    if world_size == 1:
        # the other arithmetic mode on the same workload, beside the timed one (3 launches, HIP events; not `value`)
        other = "fast" if args.arith == "exact" else "exact"
        r.set_arith(crt.CRT_ARITH_FAST if other == "fast" else crt.CRT_ARITH_EXACT)
        for _ in range(4):
            g.trace(share["first"], share["nsamples"], share["tile_stride"], share["tile_offset"])
        ost = r.stats()
        oms = float(np.mean(r.kernel_times(3))) * 1e3
        r.set_arith(crt.CRT_ARITH_FAST if args.arith == "fast" else crt.CRT_ARITH_EXACT)
        out["other_arith"] = {"arith": other, "kernel": r.kernel_name(), "kernel_ms": oms, "rays_per_launch": int(ost.rays),
                              "mrays_s": ost.rays / oms / 1e3,
                              "achieved_TFLOPs": flops_per_ray * ost.rays / (oms * 1e-3) / 1e12,  # same algorithmic flops per ray
                              "roofline_frac": flops_per_ray * ost.rays / (oms * 1e-3) / 1e12 / VALU_FP32_PEAK_TFLOPS,
                              "contract": "exact: bit-identical to the oracle; fast: >= 99 % of paths keep their segment count, "
                                          "500-spp frames agree to PSNR >= 45 dB (measured 64 dB), tests/test_gpu_round2.py"}
    if world_size == 1 and args.cpu_seconds > 0 and args.config == "c2":
        out["cpu_baseline"] = cpu_baseline(args, args.seed)
        out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]  # against the MEDIAN of the three CPU samples


if __name__ == "__main__":
    main()
