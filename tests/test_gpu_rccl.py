"""One-GPU rehearsal of the N > 1 path on the real backend: bench.py with --force-dist creates the
RCCL ("nccl") process group with one rank and issues every collective of the multi-GPU step
(reduce(SUM) of the int64 sums to rank 0, barrier, the MAX / SUM all-reduces of the timing), so the
API use, dtypes and stream ordering of that path are exercised on an MI355X; the N-rank arithmetic
itself is covered by the gloo tests (test_multi_rank_gloo.py)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_step_through_rccl_group_of_one():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29600 + os.getpid() % 300))
    common = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
              "--width", "160", "--spp", "8", "--cpu-seconds", "0"]
    plain = subprocess.run(common, env=env, capture_output=True, text=True, timeout=300)
    assert plain.returncode == 0, plain.stderr[-2000:]
    forced = subprocess.run(common + ["--force-dist"], env=env, capture_output=True, text=True, timeout=300)
    assert forced.returncode == 0, forced.stderr[-2000:]
    a = json.loads([l for l in plain.stdout.splitlines() if l.startswith("{")][-1])
    b = json.loads([l for l in forced.stdout.splitlines() if l.startswith("{")][-1])
    # same rays, and the frame that went through the RCCL reduce is the frame rendered without it
    assert a["rays_per_step"] == b["rays_per_step"] > 0
    assert a["image_mean"] == b["image_mean"]
    assert b["n_gpus"] == 1 and b["value"] > 0


@pytest.mark.gpu
def test_hip_path_under_n_ranks_on_one_gpu():
    """The N-rank path with the REAL kernels in every rank (BASELINE.json configs[2], as far as one GPU allows):
    `python bench.py --gpus N --share-gpu` starts its own N ranks, all on GPU 0; each traces its share through the C
    ABI, the int64 frames are reduced over gloo through host copies (RCCL refuses two ranks on one device).  The frame
    must be the single-rank frame bit for bit (SHA-256 of the fp32 frame) and the rays must add up, for both splits."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    common = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--width", "320", "--spp", "9",
              "--cpu-seconds", "0"]

    def run(extra):
        r = subprocess.run(common + extra, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])

    one = run(["--gpus", "1"])
    two = run(["--gpus", "2", "--share-gpu"])
    three = run(["--gpus", "3", "--share-gpu", "--split", "tiles"])
    assert two["n_gpus"] == 2 and three["n_gpus"] == 3
    assert not one["pipelined"] and two["pipelined"] and three["pipelined"]  # N > 1: double-buffered steps by default
    for other in (two, three):
        assert other["frame_sha256"] == one["frame_sha256"]
        assert other["rays_per_step"] == one["rays_per_step"] > 0
        assert other["msamples_per_s"] > 0


@pytest.mark.gpu
def test_pipelined_steps_give_the_single_rank_frame_every_step():
    """Double-buffered steps (bench.py --pipeline, the default for N > 1): image k on context / sums / stream k % 2, so the
    reduce and resolve of image k run behind the trace of image k+1.  Every one of 5 timed steps, at 2 and 3 ranks
    (both splits) and on one rank, must produce the frame of the plain single-rank step, bit for bit."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    common = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--width", "320", "--spp", "9",
              "--cpu-seconds", "0", "--verify-steps"]

    def run(extra):
        r = subprocess.run(common + extra, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])

    plain = run(["--gpus", "1", "--pipeline", "off"])
    want = plain["frame_sha256"]
    assert plain["step_frame_sha256"] == [want] * 5 and not plain["pipelined"]
    for extra in (["--gpus", "1", "--pipeline", "on"], ["--gpus", "2", "--share-gpu"],
                  ["--gpus", "3", "--share-gpu", "--split", "tiles"], ["--gpus", "1", "--force-dist"]):
        out = run(extra)
        assert out["pipelined"], extra
        assert out["step_frame_sha256"] == [want] * 5, extra
        assert out["frame_sha256"] == want and out["rays_per_step"] == plain["rays_per_step"]
