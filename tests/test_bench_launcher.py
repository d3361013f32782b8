"""bench.py's own N-rank launcher and its control flow, without a GPU: `python bench.py --gpus N` (the driver's
command) must start N ranks by itself, every collective must be issued by every rank (a rank-asymmetric collective
hangs the rehearsal and the timeout names it), and a missing GPU must be reported by the child ranks."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def test_rank_launch_command():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.rank_launch_command(4, ["--gpus", "4", "--steps", "3", "--warmup", "1"], 29999)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    i = cmd.index(BENCH)
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert 1024 < bench.free_port() < 65536


def _run(args, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, BENCH, *args], env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("gpus,split", [(2, "samples"), (3, "tiles"), (1, "samples")])
def test_control_flow_rehearsal(gpus, split):
    r = _run(["--gpus", str(gpus), "--steps", "2", "--warmup", "1", "--width", "72", "--spp", "7", "--rehearse-cpu",
              "--split", split])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == gpus and out["data"] == "rehearsal" and out["steps"] == 2
    # the stand-in tracer adds 2^40 per sample: a complete image resolves to exactly 1.0 whatever the split,
    # and "rays" = 2 per sample summed over ALL ranks
    assert out["image_mean"] == 1.0
    assert out["rays_per_step"] == 2.0 * 72 * 40 * 7
    assert out["value"] > 0 and out["time_to_image_ms"] > 0
    assert out["pipelined"] == (gpus > 1)  # N > 1: double-buffered steps


def test_pipelined_control_flow_rehearsal():
    """The double-buffered step loop on CPU tensors, 2 ranks: every timed step's frame is the complete image."""
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--width", "72", "--spp", "7", "--rehearse-cpu", "--verify-steps"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["pipelined"] and len(out["step_frame_sha256"]) == 3
    assert set(out["step_frame_sha256"]) == {out["frame_sha256"]}


def test_missing_gpus_are_reported_by_the_ranks():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has the GPUs")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--cpu-seconds", "0"])
    assert r.returncode != 0
    assert "2 devices requested" in r.stderr and "visible" in r.stderr
    assert "WORLD_SIZE=" not in r.stderr  # not the old "launch with torch.distributed.run" refusal
