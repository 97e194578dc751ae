"""`-m gpu`: bench.py's multi-rank code path with its exchange legs EXECUTED, on a one-GPU box.

The driver launches `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` on an 8-GPU node; this test runs
the same launch with N = 2 and DTSIM_BENCH_ONE_GPU=1 (both ranks on device 0, gloo for the collectives: the frame / observation
exchange is staged through pinned host memory instead of RCCL over xGMI) at a small batch, and checks the JSON line: whole-job
value over both ranks, and every `gather` leg present, finite and checksum-clean.  What it cannot cover is the RCCL transport itself
(one GPU): that stays "unmeasured on hardware" in DESIGN.md 6.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(extra_env):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, DTSIM_BENCH_ONE_GPU="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--envs", "64", "--steps", "3", "--warmup", "1",
           "--windows", "1", "--cpu-steps", "0", "--gather-timeout", "240"]
    return subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)


def test_bench_two_ranks_one_gpu_runs_the_gather_legs():
    r = _launch({})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines                         # ONE json line: the guard process stood down
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert abs(d["value"] - 2 * 64 * 3 / (d["ms_per_step"] * 3e-3)) <= 1e-6 * d["value"]      # whole-job: both ranks' env-steps
    g = d["gather"]
    assert g and "error" not in g, g
    assert g["value"] > 0 and g["checksum_ok"] is True and g["bytes_per_rank_per_step"] == 64 * 480 * 640 * 3
    for leg in ("to_root_overlapped", "observations", "observations_to_root_overlapped"):
        assert leg in g and "error" not in g[leg] and g[leg]["value"] > 0, (leg, g.get(leg))


def test_bench_line_survives_a_crash_inside_the_exchange():
    """The frame exchange between GPUs has never run on hardware: if rank 0 dies inside it (here: os._exit through a test hook), the
    guard process bench.py armed before the exchange prints the line measured so far -- the driver's scaling run still gets its value."""
    r = _launch({"DTSIM_BENCH_DIE_IN_GATHER": "1"})
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["ms_per_step"] > 0
    assert "error" in d["gather"] and "frame exchange" in d["gather"]["error"]
