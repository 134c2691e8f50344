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
    assert d["n_gpus"] == 2 and d["launched_ranks"] == 2 and d["world_size_env"] == "2" and d["ranks_seen"] == [0, 1]
    assert d["rccl_ranks"] == 0                    # gloo rehearsal: no RCCL rank exists


def test_gpus_1_runs_in_process():
    r = subprocess.run([sys.executable, BENCH, "--dry-launch"], env=_clean_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["world_size_env"] is None


def test_gpus_must_match_the_launcher():
    env = dict(_clean_env(), WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--dry-launch"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])          # the refusal is a JSON line with "error", not only a message
    assert "WORLD_SIZE=2" in d["error"] and d["value"] is None and d["n_gpus"] == 4


def test_missing_gpus_fail_loudly_with_a_json_error_line():
    """`bench.py --gpus N` with fewer than N visible GPUs must not hang, fall back or print a bare message: exit code
    != 0 and ONE JSON line carrying "error" (64 GPUs exist on no box of the pool, so this runs the same on the CPU container and a GPU box)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "64", "--steps", "1", "--warmup", "0"], env=_clean_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert "error" in d and "64" in d["error"] and d["value"] is None and d["n_gpus"] == 64 and d["visible_gpus"] < 64


def _stub_expectation(n_gpus, batch, n_job, n_batches):
    """What the stub engine must produce for the job: document g of the job belongs to rank g % world, is that rank's local document
    j = g // world, i.e. position j % batch of step j // batch, which runs resident batch (j // batch) % n_batches (round 6: the steps cycle
    through DISTINCT batches, batch i of rank r drawn with seed 1234 + 1000 r + 7919 i); exit = sum(input_ids) % 6, logits one-hot at
    sum % 16, confidence 0.5 (bench._StubEngine)."""
    import importlib
    import numpy as np
    pkg = importlib.import_module("multi-modal-early-exit_amd")
    cfg = pkg.ModelConfig.base(EE_config=dict(exits=[2, 4, 6, 8, 10], encoder_layer_strategy="ramp"))
    sums = [[pkg.synth.make_documents(cfg, batch, seed=1234 + 1000 * r + 7919 * i, text_len=512)["input_ids"].sum(1) for i in range(n_batches)]
            for r in range(n_gpus)]
    pick = lambda g: sums[g % n_gpus][((g // n_gpus) // batch) % n_batches][(g // n_gpus) % batch]
    ex = np.array([int(pick(g) % 6) for g in range(n_job)])
    checksum = float(n_job * 1.0 + ex.sum() + 0.5 * n_job)          # one-hot logits + exit index + confidence
    layer = np.array([2, 4, 6, 8, 10, 12])[ex]
    return ex, layer, checksum


def _run_stub(extra):
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--stub-engine", "--batch", "8"] + extra, env=_clean_env(),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_rank_body_weak_scaling_on_two_gloo_ranks():
    """bench.py's own rank body (threshold broadcast -> pinned plan broadcast -> sharded steps -> ONE all-gather -> max over ranks ->
    per-rank statistics -> line) executed by two real ranks over gloo with a stand-in engine: what the driver's N > 1 runs execute, minus the GPU."""
    d = _run_stub(["--steps", "3", "--warmup", "1"])
    ex, layer, checksum = _stub_expectation(2, 8, 2 * 8 * 3, 3)        # one resident batch per step
    assert d["stub_engine"] and d["n_gpus"] == 2 and d["rccl_ranks"] == 0 and d["collective_backend"] == "gloo"
    assert d["scaling"] == "weak" and d["steps"] == 3 and d["config"]["total_docs"] == 48
    assert abs(d["gathered_checksum"] - checksum) < 1e-6
    assert abs(d["mean_exit_layer"] - layer.mean()) < 1e-9
    assert d["per_rank"]["docs"] == [24, 24]
    assert abs(d["per_rank"]["mean_exit_layer"][0] - layer[0::2].mean()) < 1e-3 and abs(d["per_rank"]["mean_exit_layer"][1] - layer[1::2].mean()) < 1e-3
    assert d["value"] > 0 and len(d["per_rank"]["compute_ms"]) == 2
    assert d["distinct_documents"]["resident_batches"] == 3 and d["distinct_documents"]["every_timed_document_distinct"]


def test_rank_body_strong_scaling_uneven_shards():
    """--total-docs (BASELINE configs[3]'s mode): 37 documents over 2 ranks = shards of 19 and 18, batches of 8 -> 3 steps with a partial
    last batch; the gathered array must hold every document once, in order."""
    d = _run_stub(["--warmup", "0", "--total-docs", "37"])
    ex, layer, checksum = _stub_expectation(2, 8, 37, 3)                # ceil(37 / (2 x 8)) = 3 resident batches
    assert d["scaling"] == "strong" and d["steps"] == 3 and d["config"]["total_docs"] == 37
    assert d["per_rank"]["docs"] == [19, 18]
    assert abs(d["gathered_checksum"] - checksum) < 1e-6
    assert abs(d["mean_exit_layer"] - layer.mean()) < 1e-9
