"""`python bench.py --gpus N` must start N ranks itself (the driver's N = 1 call has no launcher in front of it, and the
N > 1 contract is `torch.distributed.run`): rehearsed here on CPU with --dry-launch, which joins a gloo group and never
touches the GPU."""
import json
import os
import subprocess
import sys

from .conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_gpus_2_launches_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--dry-launch"], env=_clean_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["world_size_env"] == "2" and d["ranks_seen"] == [0, 1]


def test_gpus_1_runs_in_process():
    r = subprocess.run([sys.executable, BENCH, "--dry-launch"], env=_clean_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["world_size_env"] is None


def test_gpus_must_match_the_launcher():
    env = dict(_clean_env(), WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--dry-launch"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
