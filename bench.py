#!/usr/bin/env python3
"""bench.py — docs/sec (+ mean exit layer) of the MI355X early-exit document-classification path.

    python bench.py --gpus N --steps K --warmup W

N > 1 runs one rank per GPU over RCCL.  Started bare (`python bench.py --gpus 8`), this process launches the ranks itself as a
child `python -m torch.distributed.run --nproc-per-node N bench.py ...` before anything touches the GPU and passes the
child's JSON line and exit code through; started by a launcher (RANK / WORLD_SIZE in the environment) it is one of the ranks,
and --gpus must equal WORLD_SIZE.

Workload (BASELINE.json configs[1]): LayoutLMv3-base, exit head every 2 layers (2,4,6,8,10) + final classifier, ramp
policy (max-confidence thresholds, strict '>'), synthetic RVL-CDIP-shaped documents (512 text tokens padded + 197
visual tokens), random-init weights.  One *step* = one pass of the hot path (ee_forward: embeddings ->
encoder layers with on-device exit + compaction -> (logits, exit_layer, confidence)) over one batch that is already
resident in HBM.  N > 1: documents shard data-parallel over the ranks (weak scaling, no data-path collective) and one
RCCL all-gather of the per-document results closes the timed region.

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events around the dominant kernel's launches
(ee_profile); `cpu_baseline` times the CPU oracle (full depth, every exit, simulated policy — the reference's own
semantics) on a bounded sample and doubles as a parity check of the same documents.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md, Matrix cores: v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md, Matrix cores: BF16/F16 ~2.5 PF dense
SPLIT_TERMS = 3                   # f16 MFMA terms per algorithmic MAC in the split-precision GEMM (hi*hi + hi*lo + lo*hi)
EXIT_LAYERS = [2, 4, 6, 8, 10]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=512, help="documents per step per GPU")
    ap.add_argument("--precision", default="auto", choices=["auto", "fp32", "split"],
                    help="GEMM back end: fp32 = v_mfma_f32_32x32x2_f32; split = f32 operands as two f16 planes, three f16 MFMA terms, "
                         "f32 accumulate (same parity bar); auto = split where the layer shapes allow it")
    ap.add_argument("--workload", default="config2", choices=["config2", "config3", "config5"],
                    help="config2 (default, BASELINE configs[1]): base, ramp exits every 2 layers.  config3 (BASELINE configs[2]): "
                         "LayoutLMv3-large, gate exit at every layer, per-exit temperatures.  config5 (BASELINE configs[4]): "
                         "image-only DiT-base (BEiT), ramp exit head at every layer")
    ap.add_argument("--dense-rows", action="store_true", help="keep pad rows (A/B switch of the ragged layout)")
    ap.add_argument("--whole-layers", action="store_true", help="run exit layers whole before deciding (A/B switch of probe-first)")
    ap.add_argument("--probe-always", action="store_true", help="probe first at every exit layer (default: chosen per layer)")
    ap.add_argument("--release", type=float, default=0.2, help="fraction of arriving documents each exit releases")
    ap.add_argument("--cpu-docs", type=int, default=-1, help="documents of the CPU baseline sample (-1 = auto, 0 = skip)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--thresholds", default="", help="comma list of per-exit thresholds: skip the calibration pass (used for "
                    "rocprofv3 runs so that every forward in the process is an identical step)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc child passes (HBM traffic)")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--stream-docs", type=int, default=40000,
                    help="N = 1 only: after the resident-batch measurement, push this many DISTINCT raw documents (uint8 pages + ragged "
                         "token ids) through feed.DeviceFeeder -> early_exit and report feed_inclusive_docs_per_sec (0 = skip)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="rehearse the launch only: every rank joins a gloo group, reports its RANK / WORLD_SIZE and exits "
                         "without touching the GPU (CPU test of the --gpus N path)")
    return ap.parse_args()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(a):
    """`python bench.py --gpus N` with no rank environment: start N ranks (one per GPU) as a CHILD
    `python -m torch.distributed.run` and pass its output and exit code through.  This process never imports torch and
    never touches the GPU (a process that has initialised HIP must not exec / be replaced, and the children need the
    devices to themselves)."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver supports dmabuf IPC only (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def dry_launch(a, world, rank):
    """Launch rehearsal: no GPU call.  Rank 0 prints which ranks joined."""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
        seen = torch.zeros(world, dtype=torch.int64)
        mine = torch.tensor([rank], dtype=torch.int64)
        dist.all_gather_into_tensor(seen, mine)
        ranks = seen.tolist()
        dist.destroy_process_group()
    else:
        ranks = [0]
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": world, "rccl_ranks": world, "ranks_seen": ranks,
                          "world_size_env": os.environ.get("WORLD_SIZE"), "gpus_arg": a.gpus}))


def calibrate_thresholds(conf, release):
    """Per-exit thresholds (the interface of Policy.accuracy_calibration_heuristic, EE/policy.py:77-92) from one
    dump-all pass: every exit releases the fraction ``release`` of the documents that reach it.  With random-init
    weights the CLS states of different documents are strongly correlated, so ONE global threshold degenerates into
    "an exit fires for everybody or nobody"; per-exit quantiles give the documented exit mix instead.  Each threshold
    sits in the middle of a gap between neighbouring confidences so that the strict '>' test is well-posed."""
    E1, n = conf.shape
    active = np.ones(n, dtype=bool)
    thr = np.full(E1, 2.0)
    for e in range(E1 - 1):
        c = np.sort(conf[e, active])
        if len(c) < 2:
            break
        k = int(round((1.0 - release) * len(c)))
        k = min(max(k, 1), len(c) - 1)
        # widest gap in a small window around the quantile
        lo, hi = max(1, k - 3), min(len(c) - 1, k + 3)
        j = lo + int(np.argmax(c[lo:hi + 1] - c[lo - 1:hi]))
        thr[e] = 0.5 * (c[j - 1] + c[j])
        active &= ~(conf[e] > thr[e])
    return thr


def measure_hbm_traffic(thr, batch, kernel_substr, precision, timeout=240):
    """HBM bytes per launch of the dominant kernel from the PMC counters, as MI355X_MICROARCH.md (HBM section)
    prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (TCC slots), both in KiB; on gfx950
    FETCH_SIZE reports half of the bytes of wide coalesced streaming reads, so it is doubled; WRITE_SIZE is exact for
    16-byte-per-lane stores.  Each pass is a child process running one identical step of this script.  Returns
    (bytes_per_launch, detail) or (None, reason)."""
    import csv, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not found"
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="mmee_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "t", "--", sys.executable,
               os.path.abspath(__file__), "--steps", "1", "--warmup", "0", "--cpu-docs", "0", "--no-profile", "--no-traffic",
               "--stream-docs", "0",
               "--batch", str(batch), "--precision", precision, "--thresholds", ",".join(repr(float(t)) for t in thr[:-1])]
        env = dict(os.environ, TMPDIR="/tmp")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout, check=True)
            f = [os.path.join(dp, x) for dp, _, fs in os.walk(d) for x in fs if x.endswith("counter_collection.csv")]
            tot, n = 0.0, 0
            for row in csv.DictReader(open(f[0])):
                if row["Counter_Name"] == ctr and kernel_substr in row["Kernel_Name"]:
                    tot += float(row["Counter_Value"]); n += 1
            if not n:
                return None, f"{ctr}: kernel not found in the counter file"
            vals[ctr] = tot / n
        except Exception as e:  # noqa: BLE001
            return None, f"{ctr} pass failed: {type(e).__name__}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    b = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
    return b, {"FETCH_SIZE_KiB_raw": vals["FETCH_SIZE"], "WRITE_SIZE_KiB": vals["WRITE_SIZE"],
               "correction": "gfx950: FETCH_SIZE x2 (wide coalesced reads), WRITE_SIZE exact"}


def main():
    a = parse()
    if a.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if a.gpus > 1:                       # parent of the ranks: nothing below this line runs in it
            sys.exit(launch_ranks(a))
    elif int(os.environ["WORLD_SIZE"]) != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks; "
                 f"pass --gpus {os.environ['WORLD_SIZE']} (or run `python bench.py --gpus N` and let it launch the ranks)")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.dry_launch:
        return dry_launch(a, world, rank)
    import torch
    import torch.distributed as dist
    # one process per GPU over RCCL ("nccl"); MMEE_DIST_BACKEND=gloo lets two ranks share one GPU to rehearse the flow on a
    # one-GPU box (collectives then travel through host memory)
    backend = os.environ.get("MMEE_DIST_BACKEND", "nccl")
    ndev = max(1, torch.cuda.device_count())
    if backend == "nccl" and world > ndev:
        sys.exit(f"bench.py: {world} RCCL ranks but only {ndev} GPU(s) visible (MMEE_DIST_BACKEND=gloo rehearses on fewer)")
    local_dev = local % ndev if backend != "nccl" else local
    dev = torch.device(f"cuda:{local_dev}")
    torch.cuda.set_device(dev)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    pkg = importlib.import_module("multi-modal-early-exit_amd")

    global EXIT_LAYERS
    temps = None
    if a.workload == "config3":
        EXIT_LAYERS = list(range(1, 24))
        ee = dict(exits=list(EXIT_LAYERS), encoder_layer_strategy="gate", inference_strategy="max_confidence")
        cfg = pkg.ModelConfig.large(EE_config=ee)
        temps = np.random.default_rng(a.seed).uniform(0.5, 3.0, len(EXIT_LAYERS) + 1)
    elif a.workload == "config5":
        EXIT_LAYERS = list(range(1, 12))
        ee = dict(exits=list(EXIT_LAYERS), encoder_layer_strategy="ramp", inference_strategy="max_confidence")
        cfg = pkg.ModelConfig.dit_base(EE_config=ee)
    else:
        ee = dict(exits=list(EXIT_LAYERS), encoder_layer_strategy="ramp", inference_strategy="max_confidence")
        cfg = pkg.ModelConfig.base(EE_config=ee)
    beit = cfg.arch == "beit"
    W = (pkg.synth.make_weights_beit if beit else pkg.synth.make_weights)(cfg, seed=a.seed, head_gain=6.0)
    B, T = a.batch, 512
    eng = pkg.EarlyExitEngine(cfg, max_docs=B, max_text_len=T, precision=a.precision, device=dev)
    eng.load_weights(W)
    docs = pkg.synth.make_documents(cfg, B, seed=a.seed + 1000 * rank, text_len=T if not beit else 8)
    d_px = torch.from_numpy(docs["pixel_values"]).to(dev)
    if beit:
        d_ids = d_am = d_bb = None
    else:
        d_ids = torch.from_numpy(docs["input_ids"]).to(dev)
        d_am = torch.from_numpy(docs["attention_mask"]).to(dev)
        d_bb = torch.from_numpy(docs["bbox"]).to(dev)

    # ---- threshold calibration on the resident batch (untimed): dump-all pass -> confidences -> global threshold ----
    if a.thresholds:
        thr = np.array([float(x) for x in a.thresholds.split(",")] + [2.0])[:len(EXIT_LAYERS) + 1]
    else:
        out = eng.forward(d_ids, d_am, d_bb, d_px, dump_all=True, want_all=True, dense_rows=a.dense_rows, temperatures=temps)
        conf = out.all_crit.cpu().numpy().astype(np.float64)
        thr = calibrate_thresholds(conf, a.release)
    if world > 1:                       # every rank uses rank 0's thresholds
        thr = pkg.dist.broadcast_array(thr, 0, device=dev)

    def step():
        return eng.forward(d_ids, d_am, d_bb, d_px, thresholds=thr, dense_rows=a.dense_rows, temperatures=temps, whole_layers=a.whole_layers, probe_always=a.probe_always)

    def run_local(idx):
        # this rank's shard of the job's documents (global document g = rank + world * i, i = step * B + position): K steps of
        # the hot path over the resident batch, one (logits | exit_layer | confidence) row per document
        rows = []
        for _ in range(a.steps):
            o = step()
            rows.append(pkg.dist.pack_results(o.logits, o.exit_layer, o.confidence))
        run_local.last = o
        r = torch.cat(rows, dim=0)
        assert r.shape[0] == len(idx)
        return r

    for _ in range(a.warmup):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()

    # no data-path collective; the one all-gather of the per-document results closes the timed region (RCCL over xGMI)
    gathered = pkg.dist.run_sharded(run_local, world * B * a.steps, rank, world)
    out = run_local.last
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        dt = pkg.dist.max_over_ranks(dt, device=dev)

    n_docs = gathered.shape[0]
    exits = gathered[:, cfg.num_labels].cpu().numpy().astype(np.int64)
    layer_of_exit = np.array(list(EXIT_LAYERS) + [cfg.num_hidden_layers])
    counts = eng.stage_counts()
    fl = eng.flops()

    line = {
        "metric": "docs_per_sec", "value": n_docs / dt, "unit": "docs/s", "n_gpus": world, "rccl_ranks": world,
        "collective_backend": ("none (single rank)" if world == 1 else "RCCL all_gather_into_tensor" if backend == "nccl" else backend),
        "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32" if eng.precision in ("fp32", "f32") else "f32 (operands split into 2 x f16, 3 f16 MFMA terms per MAC, f32 accumulate)",
        "data": "synthetic",
        "config": {"workload": ("BASELINE configs[1]: LayoutLMv3-base, exits at layers 2/4/6/8/10 + final, ramp, "
                                "per-exit max-confidence thresholds, S=512+197, synthetic RVL-CDIP-shaped docs, random-init weights")
                   if a.workload == "config2" else
                   ("BASELINE configs[2]: LayoutLMv3-large, gate exit at every layer (policy sees classifier(gate input)), "
                    "per-exit temperatures U(0.5,3), per-exit thresholds, S=512+197, synthetic docs, random-init weights")
                   if a.workload == "config3" else
                   ("BASELINE configs[4]: image-only DiT-base (BEiT, S=197), ramp exit head at every layer (extrapolation: the "
                    "reference defines no DiT exits), per-exit thresholds, synthetic pages, random-init weights"),
                   "docs_per_step_per_gpu": B, "text_len": T, "rows_layout": "dense" if a.dense_rows else "ragged",
                   "exit_layers": "whole" if a.whole_layers else "probe first" if a.probe_always else "probe first where it pays",
                   "parallelism": f"dp{world}", "thresholds": [round(float(t), 6) for t in thr[:-1]],
                   "release_fraction_per_exit": a.release},
        "mean_exit_layer": float(layer_of_exit[exits].mean()), "mean_exit_index": float(exits.mean()),
        "exit_distribution": {str(int(layer_of_exit[e])): float((exits == e).mean()) for e in range(len(layer_of_exit))},
        "stage_docs_last_step_rank0": counts["docs"], "executed_tflop_per_step_rank0": fl["total"] / 1e12,
        "executed_tflops_rank0": fl["total"] / (dt / a.steps) / 1e12,
    }

    if rank == 0 and not a.no_profile:
        # ---- roofline of the dominant kernel, live: HIP events around every launch of one more (untimed) step ------
        eng.profile(True)
        step()
        prof = eng.profile_read()
        eng.profile(False)
        c = eng.stage_counts()
        H, I = cfg.hidden_size, cfg.intermediate_size
        # rows each layer's FFN-up launch ran on.  Exit layers decide first (CLS probe) and run their bulk on the rows that stay;
        # the last layer is the probe alone.  The probes' own small launches are a separate role (cls_probe), not in this figure.
        plan = eng.layer_plan()
        rows = [r for r in plan["rows_main"] if r > 0]
        up_flops = sum(2.0 * r * H * I for r in rows)                  # algorithmic FLOPs of the FFN-up launches
        line["layer_plan_last_step_rank0"] = {k: plan[k] for k in ("rows_qkv", "rows_main", "docs_probe")}
        up = prof["gemm_ffn_up"]
        gemm_ms = sum(prof[k]["ms"] for k in ("gemm_qkv", "gemm_attn_out", "gemm_ffn_up", "gemm_ffn_down", "gemm_patch"))
        ach = up_flops / (up["ms"] * 1e-3) / 1e12
        split = eng.precision not in ("fp32", "f32")
        # split mode: every algorithmic MAC costs three f16 MFMA MACs, so the matrix-pipe ceiling for ALGORITHMIC flops
        # is the f16 dense peak / 3 (= 5.3 x the f32 MFMA peak); both ratios are reported
        peak = PEAK_F16_MFMA_TFLOPS / SPLIT_TERMS if split else PEAK_F32_MFMA_TFLOPS
        line["roofline"] = {"bound": "mfma",
                            "kernel": ("gemm_split_kernel<GELU, split out> (FFN up + GELU)" if split else
                                       "gemm_f32_dma_kernel<GELU> (FFN up + GELU)"),
                            "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                            "peak_basis": ("f16 dense MFMA peak 2500 TFLOP/s / 3 MFMA terms per algorithmic MAC" if split else
                                           "v_mfma_f32_32x32x2_f32 dense peak"),
                            "vs_f32_mfma_peak": ach / PEAK_F32_MFMA_TFLOPS,
                            "traffic": None, "launches": up["launches"], "avg_launch_ms": up["ms"] / max(1, up["launches"]),
                            "flops_per_launch_avg": up_flops / max(1, up["launches"])}
        tot = sum(v["ms"] for v in prof.values())
        line["kernel_time_share"] = {k: round(v["ms"] / tot, 4) for k, v in prof.items() if v["launches"]}
        fl2 = eng.flops()
        if world == 1 and not a.no_traffic:
            # kernel-name fragments of the FFN-up launches as rocprofv3 prints them
            ksub = "16>, 1, true, false, 0>" if split else "gemm_f32_dma_kernel<1, 0"   # CfgC, EPI_GELU, split output
            tb, detail = measure_hbm_traffic(thr, B, ksub, eng.precision)
            line["roofline"]["traffic"] = tb
            line["roofline"]["traffic_detail"] = detail
            if tb:
                # algorithmic HBM bytes of the same launches: read A (rows x H) + W once, write rows x I
                alg = sum(4.0 * (r * H + r * I) for r in rows) / max(1, up["launches"]) + 4.0 * H * I
                line["roofline"]["algorithmic_hbm_bytes_per_launch"] = alg
        line["gemm_class_tflops"] = fl2["gemm"] / (gemm_ms * 1e-3) / 1e12 if gemm_ms else None
        line["attention_tflops"] = fl2["attention"] / (prof["attention"]["ms"] * 1e-3) / 1e12 if prof["attention"]["ms"] else None

    if rank == 0 and world == 1 and a.stream_docs > 0 and not beit:
        # ---- streaming run: every document distinct, host packing + PCIe + device preprocessing inside the clock (the reference's
        # loop being replaced: EE/utils.py:93-98, 169-173).  The headline `value` stays the resident-batch rate. ---------------------
        stream = pkg.synth.RawDocumentStream(cfg, a.stream_docs, seed=a.seed + 77, text_len=T)
        feeder = pkg.feed.DeviceFeeder(stream, batch_size=B, size=cfg.input_size, max_length=T, device=dev)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        srows = []
        for batch in feeder:
            o = eng.forward(batch["input_ids"], batch["attention_mask"], batch["bbox"], batch["pixel_values"], thresholds=thr,
                            dense_rows=a.dense_rows, temperatures=temps, whole_layers=a.whole_layers, probe_always=a.probe_always)
            srows.append(pkg.dist.pack_results(o.logits, o.exit_layer, o.confidence))
        srows = torch.cat(srows, dim=0)
        torch.cuda.synchronize()
        sdt = time.perf_counter() - t1
        sex = srows[:, cfg.num_labels].cpu().numpy().astype(np.int64)
        line["feed_inclusive_docs_per_sec"] = len(stream) / sdt
        line["feed_inclusive"] = {"docs": len(stream), "distinct_documents": True, "seconds": sdt,
                                  "h2d_bytes_per_doc": feeder.bytes_h2d / max(1, len(stream)),
                                  "mean_exit_layer": float(layer_of_exit[sex].mean()),
                                  "what": "RawDocumentStream (1000x762 uint8 pages, ragged ids/boxes) -> DeviceFeeder (pinned double "
                                          "buffer, one async H2D copy per batch, resize/normalise/pad on a side stream) -> ee_forward; "
                                          "host packing, PCIe and preprocessing are inside the clock, page synthesis is not"}

    if rank == 0 and world == 1 and a.cpu_docs != 0:
        # ---- CPU baseline (N = 1 runs only): the oracle = the reference's semantics (every layer, every exit, simulated policy), B=1 ---
        oracle = importlib.import_module("oracle.ee_oracle")
        otorch = importlib.import_module("oracle.ee_oracle_torch")
        cores = min(16, os.cpu_count() or 1)        # the box's CPU share for one GPU
        torch.set_num_threads(cores)
        strat = ee["encoder_layer_strategy"]
        if beit:
            class _B:      # numpy restatement (S = 197 is small enough for numpy)
                def forward_all(self, b, exits, strategy="ramp"):
                    return oracle.forward_all_beit(cfg, W, b["pixel_values"], exits, strategy=strategy)
            tor = _B()
        else:
            tor = otorch.TorchOracle(cfg, W)
        one = {k: v[:1] for k, v in docs.items()}
        tor.forward_all(one, ee["exits"], strategy=strat)                  # warm the thread pool / allocator
        t1 = time.perf_counter()
        r0 = tor.forward_all(one, ee["exits"], strategy=strat)
        per_doc = time.perf_counter() - t1
        n = a.cpu_docs if a.cpu_docs > 0 else int(min(32, max(2, round(15.0 / max(per_doc, 1e-3)))))
        t1 = time.perf_counter()
        stores = [r0["logits_store"]]
        for i in range(1, n):
            stores.append(tor.forward_all({k: v[i:i + 1] for k, v in docs.items()}, ee["exits"], strategy=strat)["logits_store"])
        store = np.concatenate(stores, axis=1)
        if temps is not None:
            store = oracle.temperature_scale(store, temps)
        ex_cpu, pred_cpu, _ = oracle.policy_scan(store, thr)
        cpu_dt = per_doc + (time.perf_counter() - t1)
        line["cpu_baseline"] = {"value": n / cpu_dt, "unit": "docs/s", "cores": cores, "kind": "port",
                                "sample": f"{n} documents of the same batch, B=1 per forward (reference default "
                                          f"eval_batch_size=1), full depth + all exits + simulated policy, "
                                          f"{'numpy' if beit else 'torch-CPU'} float32 restatement on {cores} host threads"}
        g_ex = out.exit_layer.cpu().numpy()[:n] if world == 1 else exits[:n]
        g_lg = gathered[:n, :cfg.num_labels].cpu().numpy() if world > 1 else out.logits.cpu().numpy()[:n]
        line["parity_vs_cpu_sample"] = {"docs": n, "exit_index_equal": bool(np.array_equal(g_ex, ex_cpu)),
                                        "max_abs_dlogit": float(np.abs(g_lg - pred_cpu).max())}
    if rank == 0:
        print(json.dumps(line))
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
