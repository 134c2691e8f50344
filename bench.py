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
resident in HBM.  N > 1: documents shard data-parallel over the ranks (no data-path collective) and one RCCL all-gather of
the per-document results closes the timed region.  Default: weak scaling (every rank runs K steps of B documents).
`--total-docs D` (BASELINE configs[3]: 400 000 documents over 8 GPUs): strong scaling, the D documents are dealt round-robin
to the ranks (dist.shard_indices) and every rank runs ceil(shard / B) steps, the last one on a partial batch.

Prints ONE JSON line (rank 0).  `roofline` (the dominant MFMA-bound kernel) and `roofline_hbm` (every HBM-bound kernel role) are
measured live with HIP events around the launches (ee_profile) on a schedule pinned after the warm-up (ee_set_probe_mask), so that
the timed steps, the profiled step and the rocprofv3 child passes all run the same launch sequence; `cpu_baseline` times the CPU
oracle (full depth, every exit, simulated policy — the reference's own semantics) on a bounded sample, at B = 1 (the reference's
default) and at the best batch size, and doubles as a parity check of the same documents.

`--workload sweep` is the other measured path: the device-side threshold sweep of EE/large_scale.py at the
reference's scale (7 exits x 40 000 documents x 1 500 000 threshold vectors), with its own roofline and a numpy baseline.
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
PEAK_HBM_GBPS = 8000.0            # MI355X_MICROARCH.md, Chip-level parameters: HBM3E 8.0 TB/s (6.3 TB/s is what a copy achieves)
PEAK_CLOCK_GHZ = 2.4
N_SIMD = 1024                     # 256 CUs x 4
SPLIT_TERMS = 3                   # f16 MFMA terms per algorithmic MAC in the split-precision GEMM (hi*hi + hi*lo + lo*hi)
EXIT_LAYERS = [2, 4, 6, 8, 10]
NESTED_ROLES = ("pair_index", "patch_split", "head_out")      # timed inside prep / gemm_patch / exit_head (ee_profile_read)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=2048, help="documents per step per GPU.  Round 5: 2048 = two micro-batches of 1024 (the batch of rounds 3-4 per "
                                                            "handle: 22 GB of workspace each); the latency-bound parts of a step -- CLS probes, exit heads, queue tails -- "
                                                            "amortise over more documents: +1 % of the matrix-pipe ceiling against 2 x 512 on the same box "
                                                            "(profiles/r05_micro_batches_ab.txt)")
    ap.add_argument("--total-docs", type=int, default=0,
                    help="strong scaling (BASELINE configs[3]): this many documents in total, dealt round-robin to the ranks; "
                         "--steps is then derived (ceil(shard / batch))")
    ap.add_argument("--precision", default="auto", choices=["auto", "fp32", "split"],
                    help="GEMM back end: fp32 = v_mfma_f32_32x32x2_f32; split = f32 operands as two f16 planes, three f16 MFMA terms, "
                         "f32 accumulate (same parity bar); auto = split where the layer shapes allow it")
    ap.add_argument("--workload", default="config2", choices=["config2", "config3", "config5", "sweep"],
                    help="config2 (default, BASELINE configs[1]): base, ramp exits every 2 layers.  config3 (BASELINE configs[2]): "
                         "LayoutLMv3-large, gate exit at every layer, per-exit temperatures.  config5 (BASELINE configs[4]): "
                         "image-only DiT-base (BEiT), ramp exit head at every layer.  sweep: the threshold sweep of "
                         "EE/large_scale.py on the device (N3)")
    ap.add_argument("--dense-rows", action="store_true", help="keep pad rows (A/B switch of the ragged layout)")
    ap.add_argument("--whole-layers", action="store_true", help="run exit layers whole before deciding (A/B switch of probe-first)")
    ap.add_argument("--probe-always", action="store_true", help="probe first at every exit layer, whatever the cost model suggests (default: the plan ee_suggest_probe_mask prices from a warm-up forward, pinned)")
    ap.add_argument("--xprobe", dest="xprobe", action="store_true", default=True,
                    help="probe-first layers take the CLS context in X space: no Q | K | V for documents that leave (default; "
                         "MMEE_FLAG_XPROBE, built for LayoutLMv3-base shapes, other models run the K | V probe)")
    ap.add_argument("--no-xprobe", dest="xprobe", action="store_false",
                    help="probe with the layer's own K | V rows: an exit's row is then BIT-identical to the dump-all row (A/B switch)")
    ap.add_argument("--probe-layers", default=None,
                    help="comma list of 0-based layers to probe first ('' = none): pins the schedule instead of deriving it from the "
                         "warm-up (the rocprofv3 child passes get the parent's plan this way)")
    ap.add_argument("--release", type=float, default=0.2, help="fraction of arriving documents each exit releases")
    ap.add_argument("--micro-batches", type=int, default=2,
                    help="slices of a step's batch run on that many handles and HIP streams (MicroBatchedEngine: every kernel's tail overlaps "
                         "the other slice's kernels; bit-identical results); 1 = one handle, one stream")
    ap.add_argument("--serial-slices", action="store_true",
                    help="run the micro-batches one after the other on ONE stream (the profiling configuration: under rocprofv3 --kernel-trace two "
                         "streams make a small kernel's start-to-end time include its wait for the other stream's kernel, so per-kernel durations "
                         "stop adding up; the roofline's HIP-event step of a normal run is taken the same way)")
    ap.add_argument("--distinct-batches", type=int, default=-1,
                    help="DIFFERENT resident batches the timed steps cycle through (round 6: -1 = one per step, so `value` is measured over steps x "
                         "batch distinct documents -- 40 960 at the driver's --steps 20; all staged in HBM before the clock starts, 1.28 GB per 2048 "
                         "documents; 1 = rounds 1-5: every step re-runs one batch)")
    ap.add_argument("--no-small-batch", action="store_true",
                    help="skip the small_batch block (the reference's own operating points: B = 1 / 8 / 64 per forward, eager launches against the "
                         "captured graph, and BASELINE configs[0]'s 64-document job)")
    ap.add_argument("--no-extra-rates", action="store_true",
                    help="skip the fixed-work rates beside the headline (full depth with MMEE_FLAG_NO_EXIT, release fractions 0.1 / 0.3)")
    ap.add_argument("--calib-seed-offset", type=int, default=500000,
                    help="the thresholds are calibrated on a DIFFERENT synthetic batch (seed + this) than the timed one; 0 = on the timed batch itself")
    ap.add_argument("--cpu-docs", type=int, default=-1, help="documents of the CPU baseline sample (-1 = auto, 0 = skip)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--thresholds", default="", help="comma list of per-exit thresholds: skip the calibration pass (used for "
                    "rocprofv3 runs so that every forward in the process is an identical step)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 --pmc child passes (HBM traffic, clock, MFMA busy)")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--stream-docs", type=int, default=40000,
                    help="N = 1 only: after the resident-batch measurement, push this many DISTINCT raw documents (uint8 pages + ragged "
                         "token ids) through feed.DeviceFeeder -> early_exit and report feed_inclusive_docs_per_sec (0 = skip)")
    ap.add_argument("--sweep-vectors", type=int, default=1500000, help="--workload sweep: threshold vectors (EE/large_scale.py:179-180)")
    ap.add_argument("--sweep-docs", type=int, default=40000, help="--workload sweep: documents of the confidence table")
    ap.add_argument("--dry-launch", action="store_true",
                    help="rehearse the launch only: every rank joins a gloo group, reports its RANK / WORLD_SIZE and exits "
                         "without touching the GPU (CPU test of the --gpus N path)")
    ap.add_argument("--stub-engine", action="store_true",
                    help="CPU rehearsal of the whole rank body (threshold broadcast -> sharded steps -> all-gather -> max over ranks -> "
                         "line) with a stand-in engine on gloo: no GPU, the line says so and is not a measurement")
    return ap.parse_args(argv)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def fail_line(a, msg, **extra):
    """One JSON line with "error" on stdout and a non-zero exit code: a bench that cannot run the configuration it was asked for says so in
    the format its reader parses, instead of a bare message on stderr."""
    print(json.dumps({"metric": "docs_per_sec", "value": None, "unit": "docs/s", "n_gpus": a.gpus, "error": msg, **extra}))
    sys.stdout.flush()
    sys.exit(2)


def visible_gpus():
    """Devices the ranks could use.  torch.cuda.device_count() does not initialise HIP on this image, so the launching parent may ask."""
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception:  # noqa: BLE001
        return 0


def launch_ranks(a):
    """`python bench.py --gpus N` with no rank environment: start N ranks (one per GPU) as a CHILD
    `python -m torch.distributed.run` and pass its output and exit code through.  This process never touches the GPU (counting
    devices does not initialise HIP; a process that has initialised HIP must not exec / be replaced, and the children need the
    devices to themselves)."""
    import subprocess
    if not (a.dry_launch or a.stub_engine) and os.environ.get("MMEE_DIST_BACKEND", "nccl") == "nccl":
        n = visible_gpus()
        if n < a.gpus:
            fail_line(a, f"--gpus {a.gpus} needs {a.gpus} visible GPUs for its RCCL ranks, {n} visible "
                         "(MMEE_DIST_BACKEND=gloo rehearses the flow on fewer)", visible_gpus=n)
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
        print(json.dumps({"dry_launch": True, "n_gpus": world, "rccl_ranks": 0, "launched_ranks": world, "ranks_seen": ranks,
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


def pmc_pass(counters, child_args, timeout=300):
    """One `rocprofv3 --pmc <counters>` child pass over one identical step of this script (MI355X_MICROARCH.md, rocprofv3 PMC slots:
    counters that do not fit one pass go into separate calls).  Returns {kernel_name: {counter: [sum, launches], "_ns": [durations]}}
    or a string with the reason it failed."""
    import csv, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return "rocprofv3 not found"
    d = tempfile.mkdtemp(prefix="mmee_pmc_", dir="/tmp")
    cmd = [exe, "--pmc"] + list(counters) + ["--output-format", "csv", "-d", d, "-o", "t", "--", sys.executable,
                                              os.path.abspath(__file__)] + child_args
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = {}
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout, text=True)
        if r.returncode != 0:
            return f"{'+'.join(counters)} pass failed: rocprofv3 exit code {r.returncode}: {(r.stderr or '')[-300:].strip()}"
        f = [os.path.join(dp, x) for dp, _, fs in os.walk(d) for x in fs if x.endswith("counter_collection.csv")]
        if not f:
            return f"{'+'.join(counters)} pass failed: rocprofv3 wrote no counter_collection.csv"
        seen = set()
        for row in csv.DictReader(open(f[0])):
            k = out.setdefault(row["Kernel_Name"], {"_ns": []})
            c = k.setdefault(row["Counter_Name"], [0.0, 0])
            c[0] += float(row["Counter_Value"]); c[1] += 1
            did = row.get("Dispatch_Id")
            if did not in seen and row.get("Start_Timestamp") and row.get("End_Timestamp"):
                seen.add(did)
                k["_ns"].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    except Exception as e:  # noqa: BLE001
        return f"{'+'.join(counters)} pass failed: {type(e).__name__}: {e}"
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return out


def _pick(pass_out, substr):
    """Sum the kernels of a pmc pass whose name contains ``substr``."""
    acc = {"_ns": []}
    for name, d in pass_out.items():
        if substr in name:
            for c, v in d.items():
                if c == "_ns":
                    acc["_ns"] += v
                else:
                    a = acc.setdefault(c, [0.0, 0])
                    a[0] += v[0]; a[1] += v[1]
    return acc


class _StubEngine:
    """Stand-in for EarlyExitEngine in --stub-engine runs (CPU rehearsal of the rank body; nothing here is measured or shipped):
    the exit index is a hash of the document's ids, logits are one-hot on another, so the gathered result can be checked."""
    precision = "stub"

    def __init__(self, E, K):
        self.E, self.K = E, K
        self._last = None

    def forward(self, ids, am, bb, px, thresholds=None, **kw):
        import torch
        h = ids.sum(1)
        ex = (h % (self.E + 1)).to(torch.int32)
        lg = torch.nn.functional.one_hot((h % self.K).long(), self.K).float()
        self._last = ex
        from types import SimpleNamespace
        return SimpleNamespace(logits=lg, exit_layer=ex, confidence=torch.full((ids.shape[0],), 0.5),
                               all_crit=torch.rand((self.E + 1, ids.shape[0]), generator=torch.Generator().manual_seed(0)))

    def stage_counts(self):
        n = int(self._last.shape[0])
        return {"docs": [int((self._last >= e).sum()) for e in range(self.E + 1)], "rows": [n] * (self.E + 1)}

    def flops(self):
        return {"gemm": 0.0, "attention": 0.0, "probe": 0.0, "total": 0.0}

    def pin_schedule(self, layers=None):
        return []

    def close(self):
        pass


def sweep_workload(a):
    """N3 at the reference's scale (EE/large_scale.py:46-84, 179-180): exit(v, n) = first exit whose confidence reaches threshold
    vector v, accuracy and mean exit per vector.  One 'step' = the whole sweep."""
    import torch
    pkg = importlib.import_module("multi-modal-early-exit_amd")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    E1, N, V = 7, a.sweep_docs, a.sweep_vectors
    rng = np.random.default_rng(a.seed)
    store = rng.standard_normal((E1, N, 16)) * np.linspace(1.0, 3.0, E1)[:, None, None]
    refs = rng.integers(0, 16, N)
    store[:, np.arange(N), refs] += np.linspace(0.5, 3.0, E1)[:, None]
    thr = rng.uniform(0.0, 1.0, (V, E1))
    thr[:, -1] = 0.0                               # EE/large_scale.py:50-52: the last row's threshold stays 0 (everybody exits there at the latest)
    conf_d, corr_d = pkg.sweep.msp_table(store, refs, device=dev)
    thr_d = torch.from_numpy(thr).to(dev)
    res = None
    for _ in range(max(1, a.warmup)):
        res = pkg.sweep.threshold_sweep(conf_d, corr_d, thr_d)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(a.steps):
        res = pkg.sweep.threshold_sweep(conf_d, corr_d, thr_d)
    e1.record()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    kms = e0.elapsed_time(e1) / a.steps
    # algorithmic work of the main kernel (exit_ops.hip, sweep_main_kernel): per (vector, document) 2 E1 + 4 integer vector operations
    # (compare + select per exit, payload decode + two adds); algorithmic bytes: the rank table once per 256 vectors from
    # L2, the thresholds once, 16 bytes of results per vector
    ops_per_pair = 2 * E1 + 4
    alg_bytes = E1 * N * 9.0 + V * E1 * 8.0 + V * 16.0
    cmp_per_s = V * N * ops_per_pair / (kms * 1e-3)
    # VALU bound: 1024 SIMDs x 32 lanes per clock (a wave64 instruction issues in 2 cycles, MI355X_MICROARCH.md cycle constants)
    valu_peak = N_SIMD * 32 * PEAK_CLOCK_GHZ * 1e9
    line = {"metric": "threshold_vectors_per_sec", "value": V / dt, "unit": "vectors/s", "n_gpus": 1, "rccl_ranks": 0, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": 1e3 * dt, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 ranks (exact order transform of the f64 table)",
            "data": "synthetic",
            "config": {"workload": f"SURVEY 8f N3: threshold sweep of EE/large_scale.py:46-84 on the device, {E1} exits x {N} documents x {V} "
                                   "threshold vectors (reference: num_mixtures = 1 500 000, EE/large_scale.py:179-180)",
                       "exits": E1, "docs": N, "vectors": V},
            "roofline": {"bound": "valu (integer compare + select on ranks; the rank table is read once per 256 vectors from L2)",
                         "kernel": "sweep_rank_kernel + sweep_thr_kernel + sweep_main_kernel<7>", "achieved": cmp_per_s / 1e12, "peak": valu_peak / 1e12,
                         "unit": "T lane-op/s", "frac": cmp_per_s / valu_peak, "avg_launch_ms": kms, "lane_ops_per_vector_document": ops_per_pair,
                         "algorithmic_hbm_bytes_per_launch": alg_bytes, "hbm_GBps_algorithmic": alg_bytes / (kms * 1e-3) / 1e9,
                         "traffic": None}}
    # numpy baseline on a bounded sample of the vectors: the reference's own expression (EE/large_scale.py:46-84)
    if a.cpu_docs != 0:
        conf = conf_d.cpu().numpy()
        corr = corr_d.cpu().numpy()
        nv = 0
        t1 = time.perf_counter()
        acc_ref = []
        while time.perf_counter() - t1 < 10.0 and nv < V:
            t = thr[nv]
            ex = (conf >= t[:, None]).argmax(0)
            acc_ref.append((corr[ex, np.arange(N)].mean(), ex.mean()))
            nv += 1
        cdt = time.perf_counter() - t1
        acc_g, me_g = res[0][:nv].cpu().numpy(), res[1][:nv].cpu().numpy()
        ref = np.array(acc_ref)
        line["cpu_baseline"] = {"value": nv / cdt, "unit": "vectors/s", "cores": 1, "kind": "port",
                                "sample": f"{nv} of the {V} threshold vectors, numpy `(CSF >= thr[:, None]).argmax(0)` + accuracy / mean exit "
                                          "per vector as EE/large_scale.py:46-96 computes them, one host thread"}
        line["parity_vs_cpu_sample"] = {"vectors": nv, "accuracy_equal": bool(np.array_equal(acc_g, ref[:, 0])),
                                        "mean_exit_equal": bool(np.array_equal(me_g, ref[:, 1]))}
    print(json.dumps(line))


def small_batch_block(pkg, a, cfg, W, ee, batch0, B, T, thr, temps, dev, beit):
    """The reference's own operating points on the MI355X: it evaluates at eval_batch_size = 1 (EE/configs.py:36; loop
    EE/utils.py:169-193) and BASELINE configs[0] is a 64-document job.  One handle sized for 64 documents runs forwards of 1 / 8 / 64 DIFFERENT
    documents (slices of resident batch 0, the headline's thresholds, default schedule: every decision layer probed first, X-space probe)
    three ways: eager launches enqueued back to back (`pipelined`), eager with a synchronisation behind every forward (`latency`: what a caller
    that reads each result before the next call sees), and the captured graph (ee_graph_capture / ee_graph_launch: one graph launch per
    forward, inputs copied device-to-device into the graph's static buffers inside the clock).  Then configs[0]'s job itself: base, exit
    head at layer 6 + final, 64 documents."""
    import torch
    sync = torch.cuda.synchronize
    out = {"what": "docs/s and ms per forward of small batches on ONE handle / stream; eager = ~185 kernel launches per forward, graph = one "
                   "hipGraphLaunch; every forward a different slice of resident batch 0; same thresholds / temperatures as the headline",
           "by_batch": {}}

    def run_points(eng, thr_, temps_, sizes, src):
        res = {}
        nb = src[3].shape[0]
        sl = lambda j, n: tuple(None if t is None else t[(j * n) % (nb - n + 1):(j * n) % (nb - n + 1) + n] for t in src)
        for n in sizes:
            n_fw = max(8, min(200, 1600 // n))
            fw = lambda j: eng.forward(*sl(j, n), thresholds=thr_, temperatures=temps_, xprobe=bool(a.xprobe))
            for j in range(3):
                fw(j)
            sync()
            t0 = time.perf_counter()
            outs = [fw(j) for j in range(n_fw)]
            sync()
            dt_p = time.perf_counter() - t0
            ex_e = torch.cat([o.exit_layer for o in outs]).cpu().numpy()
            # whole layers (what the reference does, and what the cost model picks when nobody can be saved much): 7 launches per layer
            # instead of a 13-launch probe in front of every decision
            fww = lambda j: eng.forward(*sl(j, n), thresholds=thr_, temperatures=temps_, whole_layers=True)
            fww(0); sync()
            wl = []
            for j in range(n_fw):                       # (the same documents as the pipelined loop: a forward's time depends on where its documents leave)
                t0 = time.perf_counter()
                fww(j)
                sync()
                wl.append(time.perf_counter() - t0)
            lat = []
            for j in range(n_fw):
                t0 = time.perf_counter()
                fw(j)
                sync()
                lat.append(time.perf_counter() - t0)
            first = sl(0, n)
            keys = ("input_ids", "attention_mask", "bbox", "pixel_values")
            cap = eng.capture(**{k: v.clone() for k, v in zip(keys, first) if v is not None}, thresholds=thr_, temperatures=temps_,
                              xprobe=bool(a.xprobe))

            def gl(j):
                for k, v in zip(keys, sl(j, n)):
                    if v is not None:
                        cap.inputs[k].copy_(v)
                return cap.launch(thresholds=thr_, temperatures=temps_)
            for j in range(3):
                gl(j)
            sync()
            t0 = time.perf_counter()
            exg = []
            for j in range(n_fw):
                o = gl(j)
                exg.append(o.exit_layer.clone())          # the outputs are the graph's static tensors
            sync()
            dt_g = time.perf_counter() - t0
            ex_g = torch.cat(exg).cpu().numpy()
            glat = []
            for j in range(n_fw):
                t0 = time.perf_counter()
                gl(j)
                sync()
                glat.append(time.perf_counter() - t0)
            cap.close()
            res[str(n)] = {"forwards": n_fw,
                           "eager": {"docs_per_sec": n * n_fw / dt_p, "ms_per_forward_pipelined": 1e3 * dt_p / n_fw,
                                     "ms_per_forward_latency_mean": 1e3 * float(np.mean(lat)), "ms_per_forward_latency_median": 1e3 * float(np.median(lat)),
                                     "ms_per_forward_latency_mean_whole_layers": 1e3 * float(np.mean(wl)),
                                     "ms_per_forward_latency_median_whole_layers": 1e3 * float(np.median(wl))},
                           "graph": {"docs_per_sec": n * n_fw / dt_g, "ms_per_forward_pipelined": 1e3 * dt_g / n_fw,
                                     "ms_per_forward_latency_mean": 1e3 * float(np.mean(glat)), "ms_per_forward_latency_median": 1e3 * float(np.median(glat))},
                           "graph_over_eager": (dt_p / dt_g), "exit_indices_equal": bool(np.array_equal(ex_e, ex_g))}
        return res

    eng = pkg.EarlyExitEngine(cfg, max_docs=64, max_text_len=T, precision=a.precision, device=dev)
    eng.load_weights(W)
    out["by_batch"] = run_points(eng, thr, temps, (1, 8, 64), batch0)
    eng.close()
    if a.workload == "config2" and not beit:
        # BASELINE configs[0]: "LayoutLMv3-base, 2 exit heads (layers 6/12), 64-doc RVL-CDIP subset, ramp policy" -- the exit head of layer 6 is
        # config 2's third head (same weights, renumbered), the threshold is calibrated on 64 OTHER documents of batch 0 at the same release
        ee1 = dict(exits=[6], encoder_layer_strategy="ramp", inference_strategy="max_confidence")
        cfg1 = pkg.ModelConfig.base(EE_config=ee1)
        k6 = EXIT_LAYERS.index(6)
        W1 = {k: v for k, v in W.items() if ".early_exits." not in k}
        for k, v in W.items():
            tag = f".early_exits.{k6}."
            if tag in k:
                W1[k.replace(tag, ".early_exits.0.")] = v
        e1 = pkg.EarlyExitEngine(cfg1, max_docs=64, max_text_len=T, precision=a.precision, device=dev)
        e1.load_weights(W1)
        cal = tuple(t[64:128] for t in batch0)
        conf = e1.forward(*cal, dump_all=True, want_all=True).all_crit.cpu().numpy().astype(np.float64)
        thr1 = calibrate_thresholds(conf, a.release)
        job = tuple(t[:64] for t in batch0)
        pts = run_points(e1, thr1, None, (1, 64), job)
        o = e1.forward(*job, thresholds=thr1)
        ex1 = o.exit_layer.cpu().numpy().astype(np.int64)
        out["config1_64_documents"] = {"workload": "BASELINE configs[0]: LayoutLMv3-base, exit head at layer 6 + final classifier, ramp, 64 documents",
                                       "threshold": float(thr1[0]), "mean_exit_layer": float(np.array([6, 12])[ex1].mean()),
                                       "as_one_forward_of_64": pts["64"], "as_the_references_loop_of_64_forwards_of_1": pts["1"]}
        e1.close()
    return out


def main(argv=None):
    a = parse(argv)
    if a.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if (a.gpus == 1 and "WORLD_SIZE" not in os.environ and not (a.dry_launch or a.stub_engine) and visible_gpus() < 1):
        fail_line(a, "no GPU visible: the HIP path has no CPU fallback (the CPU oracle is test infrastructure, never the thing measured)", visible_gpus=0)
    if "WORLD_SIZE" not in os.environ:
        if a.gpus > 1:                       # parent of the ranks: nothing below this line runs in it
            sys.exit(launch_ranks(a))
    elif int(os.environ["WORLD_SIZE"]) != a.gpus:
        msg = (f"--gpus {a.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks; "
               f"pass --gpus {os.environ['WORLD_SIZE']} (or run `python bench.py --gpus N` and let it launch the ranks)")
        if int(os.environ.get("RANK", "0")) == 0:
            fail_line(a, msg)
        sys.exit(2)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.dry_launch:
        return dry_launch(a, world, rank)
    if a.workload == "sweep":
        if world > 1:
            sys.exit("bench.py: --workload sweep is a one-GPU measurement")
        return sweep_workload(a)
    import torch
    import torch.distributed as dist
    stub = a.stub_engine
    # one process per GPU over RCCL ("nccl"); MMEE_DIST_BACKEND=gloo lets two ranks share one GPU to rehearse the flow on a
    # one-GPU box (collectives then travel through host memory); --stub-engine runs the rank body on the CPU over gloo
    backend = "gloo" if stub else os.environ.get("MMEE_DIST_BACKEND", "nccl")
    if stub:
        dev = torch.device("cpu")
        sync = lambda: None
    else:
        ndev = max(1, torch.cuda.device_count())
        if backend == "nccl" and world > torch.cuda.device_count():
            if rank == 0:
                fail_line(a, f"{world} RCCL ranks but only {torch.cuda.device_count()} GPU(s) visible (MMEE_DIST_BACKEND=gloo rehearses on fewer)",
                          visible_gpus=torch.cuda.device_count())
            sys.exit(2)
        local_dev = local % ndev if backend != "nccl" else local
        dev = torch.device(f"cuda:{local_dev}")
        torch.cuda.set_device(dev)
        sync = torch.cuda.synchronize
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
    B, T = a.batch, 512
    if stub:
        W = None
        eng = _StubEngine(len(EXIT_LAYERS), cfg.num_labels)
    else:
        W = (pkg.synth.make_weights_beit if beit else pkg.synth.make_weights)(cfg, seed=a.seed, head_gain=6.0)
        if a.micro_batches > 1:
            eng = pkg.MicroBatchedEngine(cfg, max_docs=B, max_text_len=T, precision=a.precision, device=dev, micro_batches=a.micro_batches)
        else:
            eng = pkg.EarlyExitEngine(cfg, max_docs=B, max_text_len=T, precision=a.precision, device=dev)
        eng.load_weights(W)
    # ---- the resident batches (round 6): BASELINE configs[1] is a 40 000-document job, so the timed steps cycle through
    # DIFFERENT batches of the same generator (batch i: seed + 1000 * rank + 7919 * i; batch 0 is the batch of rounds 1-5), all staged in HBM
    # before the clock starts (PCIe stays outside, as the contract says); 20 x 2048 documents x 0.63 MB = 25.6 GB of the 288 GB ----------------
    strong_ = a.total_docs > 0
    n_batches = a.distinct_batches if a.distinct_batches > 0 else (a.steps if not strong_ else min(20, max(1, -(-a.total_docs // (world * B)))))
    n_batches = max(1, min(n_batches, 64))
    seeds = [a.seed + 1000 * rank + 7919 * i for i in range(n_batches)]
    gen = lambda sd: pkg.synth.make_documents(cfg, B, seed=sd, text_len=T if not beit else 8)

    def to_dev(dd):
        px = torch.from_numpy(dd["pixel_values"]).to(dev)
        if beit:
            return (None, None, None, px)
        return tuple(torch.from_numpy(dd[k]).to(dev) for k in ("input_ids", "attention_mask", "bbox")) + (px,)

    t_gen = time.perf_counter()
    docs = gen(seeds[0])                              # batch 0 stays on the host too: the CPU baseline's documents
    batches = [to_dev(docs)]
    if n_batches > 1:
        # numpy's generators release the GIL: the other batches are drawn on a few threads, copied to the device as they arrive and dropped
        from concurrent.futures import ThreadPoolExecutor
        nthr = min(n_batches - 1, max(1, min(8, (os.cpu_count() or 2) // 2)))
        with ThreadPoolExecutor(nthr) as ex:
            pend = []
            it = iter(seeds[1:])
            for sd in it:
                pend.append(ex.submit(gen, sd))
                if len(pend) >= nthr:
                    batches.append(to_dev(pend.pop(0).result()))
            for f in pend:
                batches.append(to_dev(f.result()))
    t_gen = time.perf_counter() - t_gen
    d_ids, d_am, d_bb, d_px = batches[0]

    # ---- threshold calibration (untimed): dump-all pass over a DIFFERENT synthetic batch (round 5, VERDICT r04 item 2: rounds 1-4 calibrated on
    # the timed batch itself) -> confidences -> per-exit thresholds; the timed batch then leaves through them as any unseen batch would ------
    calib_conf = None
    if a.thresholds:
        thr = np.array([float(x) for x in a.thresholds.split(",")] + [2.0])[:len(EXIT_LAYERS) + 1]
    else:
        if a.calib_seed_offset and not stub:
            cdocs = pkg.synth.make_documents(cfg, B, seed=a.seed + a.calib_seed_offset, text_len=T if not beit else 8)
            c_px = torch.from_numpy(cdocs["pixel_values"]).to(dev)
            c_in = (None, None, None) if beit else tuple(torch.from_numpy(cdocs[k]).to(dev) for k in ("input_ids", "attention_mask", "bbox"))
            out = eng.forward(c_in[0], c_in[1], c_in[2], c_px, dump_all=True, want_all=True, dense_rows=a.dense_rows, temperatures=temps)
            del cdocs, c_px, c_in
        else:
            out = eng.forward(d_ids, d_am, d_bb, d_px, dump_all=True, want_all=True, dense_rows=a.dense_rows, temperatures=temps)
        calib_conf = out.all_crit.cpu().numpy().astype(np.float64)
        thr = calibrate_thresholds(calib_conf, a.release)
    if world > 1:                       # every rank uses rank 0's thresholds
        thr = pkg.dist.broadcast_array(thr, 0, device=dev)

    def step(n=None, xprobe=None, i=0, **kw):
        sl = (lambda t: t if (t is None or n is None or n == B) else t[:n])
        xp = a.xprobe if xprobe is None else xprobe
        if a.serial_slices and getattr(eng, "n", 1) > 1:
            kw.setdefault("serial", True)
        b_ids, b_am, b_bb, b_px = batches[i % len(batches)]          # step i of a leg runs resident batch i (mod the number staged)
        return eng.forward(sl(b_ids), sl(b_am), sl(b_bb), sl(b_px), thresholds=thr, dense_rows=a.dense_rows, temperatures=temps,
                           whole_layers=a.whole_layers, probe_always=a.probe_always, xprobe=bool(xp), **kw)

    # ---- the job: weak scaling = K full batches per rank; strong scaling = --total-docs dealt round-robin ----------
    strong = a.total_docs > 0
    n_job = a.total_docs if strong else world * B * a.steps
    n_mine = pkg.dist.shard_size(n_job, rank, world)
    sizes = [B] * (n_mine // B) + ([n_mine % B] if n_mine % B else [])
    steps = max(len([B] * (pkg.dist.shard_size(n_job, 0, world) // B)) + (1 if pkg.dist.shard_size(n_job, 0, world) % B else 0), 1) if strong else a.steps

    # ---- warm-up, then pin the exit-layer schedule.  The library's default (probe first at every exit layer) never changes by itself; the
    # bench asks the cost model once (ee_suggest_probe_mask on the warm-up's stage populations) and pins its answer for everything that follows ----
    for _ in range(max(1, a.warmup) if not (a.probe_layers is not None or a.whole_layers or a.probe_always or stub) else a.warmup):
        step()
    sync()
    if a.probe_layers is not None:
        plan_layers = eng.pin_schedule([int(x) for x in a.probe_layers.split(",") if x != ""])
    elif a.whole_layers or a.probe_always or stub:
        plan_layers = None
    else:
        plan_layers = eng.pin_schedule(None, xprobe=bool(a.xprobe))
    if world > 1 and plan_layers is not None:       # same launches on every rank: rank 0's plan
        pl = np.full(64, -1, dtype=np.int64)
        pl[:len(plan_layers)] = plan_layers
        pl = pkg.dist.broadcast_array(pl, 0, device=dev)
        plan_layers = eng.pin_schedule([int(x) for x in pl if x >= 0])

    t_local = [0.0]

    def run_local(idx):
        # this rank's shard of the job's documents (global document g = rank + world * i): steps of the hot path over the resident
        # batch, one (logits | exit_layer | confidence) row per document
        rows = []
        o = None
        for i, n in enumerate(sizes):
            o = step(n, i=i)
            rows.append(pkg.dist.pack_results(o.logits, o.exit_layer, o.confidence))
        run_local.last = o
        sync()
        t_local[0] = time.perf_counter() - t0
        r = torch.cat(rows, dim=0) if rows else torch.zeros((0, cfg.num_labels + 2), dtype=torch.int32, device=dev)
        assert r.shape[0] == len(idx)
        return r

    sync()
    if world > 1:
        dist.barrier()
    sync()
    # shader-clock stamps around the timed region (s_memtime / s_memrealtime per XCD, one-wave kernels on the launch stream): the clock the
    # chip HELD over these steps, which differs by +-5 % between the boxes of a pool under the same load
    stamp0 = eng.clock_stamp() if not stub else None
    t0 = time.perf_counter()

    # no data-path collective; the one all-gather of the per-document results closes the timed region (RCCL over xGMI)
    gathered = pkg.dist.run_sharded(run_local, n_job, rank, world)
    out = run_local.last
    stamp1 = eng.clock_stamp() if not stub else None
    sync()
    if world > 1:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    clock_ghz, clock_per_xcd = eng.clock_ghz(stamp0, stamp1) if not stub else (None, [])
    layer_of_exit = np.array(list(EXIT_LAYERS) + [cfg.num_hidden_layers])
    # work and stage populations of the HEADLINE schedule's last step (rounds 1-3 read them behind the K | V-probe A/B below, whose forward
    # projects Q | K | V for every row: executed_tflops was 3.5 % high)
    counts = eng.stage_counts()
    fl = eng.flops()
    # executed work of the WHOLE job, not of its last step: the batches differ (+-3 % flops), so every batch is run once more, untimed, and
    # its executed flops are read back (ee_last_flops synchronises: it cannot be asked inside the clock)
    fl_job = None
    if len(batches) > 1 and not strong and not stub:
        fl_job = 0.0
        for i in range(steps):
            step(i=i)
            fl_job += eng.flops()["total"]
    out0 = out if (len(batches) == 1 or stub) else step(i=0)      # batch 0's results: the documents the CPU baseline re-computes
    # rounds 1-5's headline, kept beside the new one: every step re-runs ONE resident batch (batch 0)
    resident_rate = None
    if world == 1 and len(batches) > 1 and not stub and not strong and not a.thresholds:
        step(i=0); sync()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            step(i=0)
        sync()
        resident_rate = a.steps * B / (time.perf_counter() - t1)
    # A/B beside the headline (N = 1): the same steps with the K | V probe (exit rows bit-identical to the dump-all rows)
    kv_probe_rate = None
    if world == 1 and a.xprobe and not stub and not a.whole_layers and not strong and not a.thresholds:   # (not in the pinned profile / PMC child runs)
        step(xprobe=False); sync()
        t1 = time.perf_counter()
        for i in range(a.steps):
            o_kv = step(xprobe=False, i=i)          # the same batches in the same order: the last outputs belong to the same documents
        sync()
        kv_probe_rate = a.steps * B / (time.perf_counter() - t1)
        kv_same_exits = bool(torch.equal(o_kv.exit_layer, out.exit_layer))
        kv_dlogit = float((o_kv.logits - out.logits).abs().max())
    # per-rank view: compute time before the all-gather, documents, mean exit layer (exit depth varies per document, so an uneven
    # deal is the one thing that can bend the scaling curve)
    g_logits, g_exit, g_conf = pkg.dist.unpack_results(gathered)      # (f32 logits, i32 exit_layer, f32 confidence): the north-star contract
    my_ex = g_exit[rank::world].cpu().numpy().astype(np.int64) if world > 1 else None
    if world > 1:
        dt = pkg.dist.max_over_ranks(dt, device=dev)
        mine = np.array([t_local[0] * 1e3, float(n_mine), float(layer_of_exit[my_ex].mean()) if len(my_ex) else 0.0])
        per_rank = np.stack([pkg.dist.broadcast_array(mine, r, device=dev) for r in range(world)])
    else:
        per_rank = None

    n_docs = gathered.shape[0]
    exits = g_exit.cpu().numpy().astype(np.int64)

    # Reported beside the headline, NEVER as `value` (SURVEY 8d config 2 / BASELINE.md section 4: "bf16 throughput mode reports its measured
    # deviation separately"): the same steps with ONE f16 MFMA term per MAC in the layer GEMMs and the attention (MMEE_FLAG_ONE_TERM)
    lowprec = None
    if (world == 1 and not stub and not beit and eng.precision == "split" and not a.whole_layers and not strong and not a.thresholds
            and a.workload == "config2"):
        step(one_term=True); sync()
        t1 = time.perf_counter()
        for i in range(a.steps):
            o_lp = step(one_term=True, i=i)
        sync()
        lp_dt = time.perf_counter() - t1
        flips = (o_lp.exit_layer != out.exit_layer)
        same = ~flips
        lowprec = {"docs_per_sec": a.steps * B / lp_dt,
                   "exit_flip_rate": float(flips.float().mean()),
                   "max_abs_dlogit_where_exit_equal": float((o_lp.logits - out.logits)[same].abs().max()) if bool(same.any()) else None,
                   "mean_exit_layer": float(layer_of_exit[o_lp.exit_layer.cpu().numpy().astype(np.int64)].mean()),
                   "what": "MMEE_FLAG_ONE_TERM: hi planes only (plain f16 operands, f32 accumulate) in the four layer GEMMs and the attention; "
                           "CLS probes and exit heads keep three terms; same thresholds as the headline run.  Outside the 1e-4 parity bar by "
                           "construction: a measured deviation, not a result"}

    # ---- fixed-work rates beside the headline (N = 1): the exit mix is a free parameter that moves docs/s far more than
    # any kernel change, and the boxes of a pool hold different clocks.  (i) the same batch at FULL depth (MMEE_FLAG_NO_EXIT: every layer,
    # every exit head, nobody leaves -- fixed work, mix-independent); (ii) the same batch under thresholds calibrated for release fractions
    # 0.1 and 0.3 (same calibration batch); each with its own clock stamps, so that docs/s per GHz can be compared between leases ----------
    extra_rates = None
    if world == 1 and not stub and not strong and not a.thresholds and not a.no_extra_rates and calib_conf is not None:
        ceiling = PEAK_F32_MFMA_TFLOPS if eng.precision in ("fp32", "f32") else PEAK_F16_MFMA_TFLOPS / SPLIT_TERMS

        def timed(fn, nsteps):
            fn(); sync()
            sa = eng.clock_stamp()
            t1 = time.perf_counter()
            o_ = None
            for _ in range(nsteps):
                o_ = fn()
            sb = eng.clock_stamp()
            sync()
            d_ = time.perf_counter() - t1
            return nsteps * B / d_, eng.clock_ghz(sa, sb)[0], o_, d_ / nsteps

        k2 = max(2, min(a.steps, 4))
        fwd = lambda **kw: eng.forward(d_ids, d_am, d_bb, d_px, dense_rows=a.dense_rows, temperatures=temps, **kw)
        r_, g_, _, spt = timed(lambda: fwd(dump_all=True), k2)
        fl_ne = eng.flops()
        extra_rates = {"steps_each": k2,
                       "no_exit": {"docs_per_sec": r_, "clock_ghz": g_, "docs_per_sec_per_ghz": (r_ / g_) if g_ else None,
                                   "executed_tflop_per_step": fl_ne["total"] / 1e12, "step_frac_of_ceiling": fl_ne["total"] / spt / 1e12 / ceiling,
                                   "what": "MMEE_FLAG_NO_EXIT: all layers and every exit head for every document of the same batch (whole layers; "
                                           "the reference's own evaluation mode, EE/utils.py:63-71): fixed work, independent of thresholds"}}
        for rel in (0.1, 0.3):
            thr_r = calibrate_thresholds(calib_conf, rel)
            f_ = lambda: fwd(thresholds=thr_r, whole_layers=a.whole_layers, probe_always=a.probe_always, xprobe=bool(a.xprobe))
            if plan_layers is not None:      # this mix's own plan, priced by the same cost model from one forward under the default schedule
                eng.pin_schedule(False)
                f_(); sync()
                pl_r = eng.pin_schedule(None, xprobe=bool(a.xprobe))
            else:
                pl_r = None
            r_, g_, o_, spt = timed(f_, k2)
            fl_r = eng.flops()
            ex_r = o_.exit_layer.cpu().numpy().astype(np.int64)
            extra_rates[f"release_{rel}"] = {"docs_per_sec": r_, "clock_ghz": g_, "docs_per_sec_per_ghz": (r_ / g_) if g_ else None,
                                             "mean_exit_layer": float(layer_of_exit[ex_r].mean()), "probe_layers": pl_r,
                                             "step_frac_of_ceiling": fl_r["total"] / spt / 1e12 / ceiling,
                                             "thresholds": [round(float(t), 6) for t in thr_r[:-1]]}
        if plan_layers is not None:
            eng.pin_schedule(plan_layers)
        else:
            eng.pin_schedule(False)

    line = {
        "metric": "docs_per_sec", "value": n_docs / dt, "unit": "docs/s", "n_gpus": world,
        "rccl_ranks": world if (backend == "nccl" and world > 1) else 0,
        "collective_backend": ("none (single rank)" if world == 1 else "RCCL all_gather_into_tensor" if backend == "nccl" else backend),
        "steps": steps,
        "warmup": a.warmup, "ms_per_step": 1e3 * dt / steps, "higher_is_better": True, "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "stub" if stub else "f32" if eng.precision in ("fp32", "f32") else
                 "f32 (operands split into 2 x f16, 3 f16 MFMA terms per MAC, f32 accumulate)",
        "data": "synthetic",
        "config": {"workload": ("BASELINE configs[1]: LayoutLMv3-base, exits at layers 2/4/6/8/10 + final, ramp, "
                                "per-exit max-confidence thresholds, S=512+197, synthetic RVL-CDIP-shaped docs, random-init weights")
                   if a.workload == "config2" else
                   ("BASELINE configs[2]: LayoutLMv3-large, gate exit at every layer (policy sees classifier(gate input)), "
                    "per-exit temperatures U(0.5,3), per-exit thresholds, S=512+197, synthetic docs, random-init weights")
                   if a.workload == "config3" else
                   ("BASELINE configs[4]: image-only DiT-base (BEiT, S=197), ramp exit head at every layer (extrapolation: the "
                    "reference defines no DiT exits), per-exit thresholds, synthetic pages, random-init weights"),
                   "docs_per_step_per_gpu": B, "total_docs": n_job, "text_len": T, "rows_layout": "dense" if a.dense_rows else "ragged",
                   "exit_layers": ("whole" if a.whole_layers else "probe first" if a.probe_always else
                                   f"probe first at layers {plan_layers} (0-based; plan of a warm-up forward, pinned)") +
                                  (", CLS context in X space (xprobe)" if a.xprobe else ""),
                   "parallelism": f"dp{world}", "thresholds": [round(float(t), 6) for t in thr[:-1]],
                   "release_fraction_per_exit": a.release,
                   "thresholds_calibrated_on": ("--thresholds" if a.thresholds else "the timed batch itself" if not a.calib_seed_offset or stub else
                                                f"a different synthetic batch (seed {a.seed + a.calib_seed_offset}; the timed batch is seed {a.seed})"),
                   "micro_batches": getattr(eng, "n", 1), **({"serial_slices": True} if a.serial_slices else {})},
        **({"kv_probe": {"docs_per_sec": kv_probe_rate, "what": "--no-xprobe: probe-first layers read the layer's K | V rows (bit-identical "
                         "to whole layers) instead of the X-space CLS context", "exit_index_equal_to_headline_run": kv_same_exits,
                         "max_abs_dlogit_vs_headline_run": kv_dlogit}} if kv_probe_rate is not None else {}),
        **({"lowprec": lowprec} if lowprec is not None else {}),
        # the shader clock held over the timed region (rank 0; s_memtime / s_memrealtime stamps, mean over the XCDs) and the rate per GHz:
        # two leases of the pool that hold different clocks should agree on the latter
        "clock_ghz_timed_region": clock_ghz, "clock_ghz_per_xcd": [round(x, 4) for x in clock_per_xcd],
        "docs_per_sec_per_ghz": (n_docs / dt / clock_ghz) if clock_ghz else None,
        **({"docs_per_sec_no_exit": extra_rates["no_exit"]["docs_per_sec"], "fixed_work_rates": extra_rates} if extra_rates else {}),
        "mean_exit_layer": float(layer_of_exit[exits].mean()), "mean_exit_index": float(exits.mean()),
        "exit_distribution": {str(int(layer_of_exit[e])): float((exits == e).mean()) for e in range(len(layer_of_exit))},
        "stage_docs_last_step_rank0": counts["docs"],
        "executed_tflop_per_step_rank0": (fl_job / steps if fl_job is not None else fl["total"]) / 1e12,
        # (strong scaling: the LAST step is a partial batch, so its flops over the mean step time would mean nothing)
        "executed_tflops_rank0": None if strong else (fl_job / dt if fl_job is not None else fl["total"] / (dt / steps)) / 1e12,
        # the WHOLE step against the matrix-pipe ceiling of its precision (split: f16 dense peak / 3 terms; fp32: the f32 MFMA peak): every
        # kernel of the step is in the numerator's time, only executed GEMM / attention / probe flops in its work
        "step_frac_of_ceiling": None if strong else ((fl_job / dt if fl_job is not None else fl["total"] / (dt / steps)) / 1e12) /
                                (PEAK_F32_MFMA_TFLOPS if eng.precision in ("fp32", "f32") else PEAK_F16_MFMA_TFLOPS / SPLIT_TERMS),
        # round 6: what the timed steps ran over
        "distinct_documents": {"resident_batches": len(batches), "documents_per_rank": len(batches) * B,
                               "every_timed_document_distinct": bool(not strong and len(batches) >= steps),
                               "seeds_rank0": [a.seed + 7919 * i for i in range(len(batches))], "staging_seconds_untimed": round(t_gen, 1),
                               "what": "the timed steps cycle through this many DIFFERENT batches of the same generator, staged in HBM before the clock "
                                       "starts; thresholds calibrated on yet another batch; exit_distribution / mean_exit_layer are over every timed document"},
        **({"value_resident_batch": resident_rate} if resident_rate is not None else {}),
    }
    if per_rank is not None:
        line["per_rank"] = {"compute_ms": [round(float(x), 3) for x in per_rank[:, 0]], "docs": [int(x) for x in per_rank[:, 1]],
                            "mean_exit_layer": [round(float(x), 4) for x in per_rank[:, 2]]}
    if stub:
        line["stub_engine"] = True
        line["metric"] = "docs_per_sec (STUB ENGINE on CPU over gloo: rehearsal of the rank body, not a measurement)"
        line["gathered_checksum"] = float((g_logits.double().sum() + g_exit.double().sum() + g_conf.double().sum()).item())

    if rank == 0 and not a.no_profile and not stub:
        # ---- rooflines, live: HIP events around every launch of one more (untimed) step of the SAME pinned schedule --------------
        eng.profile(True)
        step(**({"serial": True} if getattr(eng, "n", 1) > 1 else {}))      # micro-batches one after the other: overlapping launches would be timed twice
        prof = eng.profile_read()
        eng.profile(False)
        c = eng.stage_counts()
        H, I, K = cfg.hidden_size, cfg.intermediate_size, cfg.num_labels
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
        tot = sum(v["ms"] for k, v in prof.items() if k not in NESTED_ROLES)
        line["kernel_time_share"] = {k: round(v["ms"] / tot, 4) for k, v in prof.items() if v["launches"] and k not in NESTED_ROLES}
        fl2 = eng.flops()
        line["gemm_class_tflops"] = fl2["gemm"] / (gemm_ms * 1e-3) / 1e12 if gemm_ms else None
        line["attention_tflops"] = fl2["attention"] / (prof["attention"]["ms"] * 1e-3) / 1e12 if prof["attention"]["ms"] else None

        # ---- HBM-side roofline of the memory-bound kernel roles: SURVEY 8d algorithmic bytes / event time / 8 TB/s ----------------
        docs_e = c["docs"]                                   # documents arriving at each exit (last = final classifier)
        rows_e = c["rows"]
        n_exits = len(docs_e)
        Pv = (cfg.input_size // cfg.patch_size) ** 2 + 1
        if beit:
            n_text = 0
            len_sq = B * float(Pv) ** 2
        else:
            am = docs["attention_mask"]
            n_text = int(am.sum()) if not a.dense_rows else B * T
            lens = (am.sum(1) if not a.dense_rows else np.full(B, T)) + Pv
            len_sq = float((lens.astype(np.float64) ** 2).sum())
        pooled = any(isinstance(e, str) for e in ee["exits"])
        alg = {
            "layernorm": sum(2 * r for r in rows) * (H * 4.0 + H * 4.0),                                  # f32 sum in, split planes out
            "embed_text": (B * T if pooled else n_text) * (3 * H * 4.0 + H * 4.0),                       # word + position + 6 spatial slices in, row out
            "embed_visual": B * Pv * (H * 4.0 + H * 4.0) + Pv * H * 4.0,                                  # projected patch in, row out; pos_embed once
            # round 6: two bytes (4 bx, 4 by) per (query, key) pair + the 32-bit words of query block 0 for the X-space probe (T <= 512); before: one word per pair
            "pair_index": (len_sq * 2.0 + 128.0 * float(lens.sum())) if (not beit and T <= 512) else len_sq * 4.0,
            "patch_split": B * cfg.num_channels * cfg.input_size ** 2 * 8.0,                              # pixel in, split pixel out
            "head_out": sum(docs_e) * (H * 4.0 + K * 4.0) + n_exits * K * H * 4.0,
            "exit_decide": sum(docs_e) * (K * 4.0 + 72.0),
            "compact": sum(rows_e[1:]) * (4.0 + 16.0 + 16.0),
            "gather_cls": sum(docs_e) * 2 * H * 4.0,
        }
        hb = {}
        for role, nbytes in alg.items():
            p_ = prof.get(role)
            if not p_ or not p_["launches"] or p_["ms"] <= 0:
                continue
            gbps = nbytes / (p_["ms"] * 1e-3) / 1e9
            hb[role] = {"launches": p_["launches"], "ms": round(p_["ms"], 4), "algorithmic_bytes": nbytes, "achieved_GBps": round(gbps, 1),
                        "frac_of_8TBps": round(gbps / PEAK_HBM_GBPS, 4)}
        line["roofline_hbm"] = {"peak_GBps": PEAK_HBM_GBPS, "what": "SURVEY 8d algorithmic bytes of every launch of the role in one step / HIP-event "
                                "time of those launches; kernels of a few microseconds are launch-latency-bound, not bandwidth-bound",
                                "roles": hb}

        if world == 1 and not a.no_traffic:
            child = ["--steps", "1", "--warmup", "0", "--cpu-docs", "0", "--no-profile", "--no-traffic", "--stream-docs", "0", "--no-extra-rates", "--serial-slices",
                     "--micro-batches", str(a.micro_batches), "--batch", str(B), "--precision", eng.precision if eng.precision != "split" else "split", "--workload", a.workload,
                     "--release", str(a.release), "--thresholds", ",".join(repr(float(t)) for t in thr[:-1]),
                     "--probe-layers", ",".join(str(x) for x in (plan_layers or []))]
            if a.dense_rows:
                child.append("--dense-rows")
            if a.whole_layers:
                child.append("--whole-layers")
            if a.probe_always:
                child.append("--probe-always")
            if not a.xprobe:
                child.append("--no-xprobe")
            # kernel-name fragments as rocprofv3 prints them
            ksub = "16>, 1, true, false, 0, 3>" if split else "gemm_f32_dma_kernel<1, 0"   # CfgC, EPI_GELU, split output, three terms
            passes = {}
            for name, ctrs in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]),
                               ("sq", ["GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY"])):
                passes[name] = pmc_pass(ctrs, child)
            detail = {}
            if all(isinstance(v, dict) for v in passes.values()):
                f, w, q = (_pick(passes[k], ksub) for k in ("fetch", "write", "sq"))
                if "FETCH_SIZE" in f and "WRITE_SIZE" in w:
                    nl = max(1, f["FETCH_SIZE"][1])
                    # MI355X_MICROARCH.md, HBM: both counters in KiB; gfx950 FETCH_SIZE reports half of the bytes of wide coalesced
                    # streaming reads (doubled), WRITE_SIZE is exact for 16-byte-per-lane stores
                    tb = (2.0 * f["FETCH_SIZE"][0] / nl + w["WRITE_SIZE"][0] / max(1, w["WRITE_SIZE"][1])) * 1024.0
                    line["roofline"]["traffic"] = tb
                    detail = {"FETCH_SIZE_KiB_raw_per_launch": f["FETCH_SIZE"][0] / nl, "WRITE_SIZE_KiB_per_launch": w["WRITE_SIZE"][0] / max(1, w["WRITE_SIZE"][1]),
                              "correction": "gfx950: FETCH_SIZE x2 (wide coalesced reads), WRITE_SIZE exact"}
                    # algorithmic HBM bytes of the same launches: read A (rows x H) + W once, write rows x I
                    line["roofline"]["algorithmic_hbm_bytes_per_launch"] = sum(4.0 * (r * H + r * I) for r in rows) / max(1, up["launches"]) + 4.0 * H * I
                if "GRBM_GUI_ACTIVE" in q and q["_ns"]:
                    # MI355X_MICROARCH.md, DVFS give-back: clock = GRBM_GUI_ACTIVE / 8 XCDs / wall time of the dispatch (profiled passes
                    # run a little below the un-profiled clock); MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles of the dispatch)
                    cyc = q["GRBM_GUI_ACTIVE"][0] / 8.0
                    ghz = cyc / sum(q["_ns"])
                    busy = q["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (N_SIMD * cyc)
                    line["roofline"].update({"clock_ghz": round(ghz, 3), "mfma_busy": round(busy, 4),
                                             "frac_decomposed": {"mfma_busy_x_clock_over_2.4": round(busy * ghz / PEAK_CLOCK_GHZ, 4),
                                                                 "wait_share_of_wave_cycles": round(q["SQ_WAIT_ANY"][0] / max(1.0, q["SQ_WAVE_CYCLES"][0]), 4),
                                                                 "note": "frac = algorithmic flops / time / peak; busy x clock / 2.4 GHz counts every issued MFMA "
                                                                         "(the three terms of a MAC and padded tile rows included), measured under the profiler"}})
                # HBM traffic of the LayerNorm kernel, the largest memory-bound role
                fl_, wl_ = _pick(passes["fetch"], "ln_rows_kernel"), _pick(passes["write"], "ln_rows_kernel")
                if "FETCH_SIZE" in fl_ and "WRITE_SIZE" in wl_ and "layernorm" in hb:
                    hb["layernorm"]["traffic_bytes"] = (2.0 * fl_["FETCH_SIZE"][0] + wl_["WRITE_SIZE"][0]) * 1024.0
            else:
                detail = {"failed": [v for v in passes.values() if isinstance(v, str)]}
            if line["roofline"].get("traffic") is None:
                # a failed counter pass is part of the line, not a silently missing field
                line["roofline"]["traffic_error"] = "; ".join(v for v in passes.values() if isinstance(v, str)) or "counter rows for the FFN-up kernel not found in the pmc output"
            line["roofline"]["traffic_detail"] = detail

    if rank == 0 and world == 1 and not stub and not strong and not a.no_small_batch and not a.thresholds and not a.no_extra_rates:
        line["small_batch"] = small_batch_block(pkg, a, cfg, W, ee, batches[0], B, T, thr, temps, dev, beit)

    if rank == 0 and world == 1 and a.stream_docs > 0 and not beit and not stub and not strong:
        # ---- streaming run: every document distinct, host packing + PCIe + device preprocessing inside the clock (the reference's
        # loop being replaced: EE/utils.py:93-98, 169-173).  The headline `value` stays the resident-batch rate. ---------------------
        # Round 6: the streamed pages are scans (strokes + sensor noise through the device resize), the resident generator
        # draws independent pixels -- other visual statistics, so the resident thresholds released another mix on them (mean exit layer 6.79
        # against 7.29 in round 5: "not comparable").  The stream now gets thresholds calibrated the same way as the resident ones -- one
        # dump-all pass over B documents of ANOTHER stream (seed + 78), the same release fraction -- so both runs do the same nominal work.
        thr_s = thr
        if not a.thresholds:
            cstream = pkg.synth.RawDocumentStream(cfg, B, seed=a.seed + 78, text_len=T)
            cb = next(iter(pkg.feed.DeviceFeeder(cstream, batch_size=B, size=cfg.input_size, max_length=T, device=dev)))
            co = eng.forward(cb["input_ids"], cb["attention_mask"], cb["bbox"], cb["pixel_values"], dump_all=True, want_all=True,
                             dense_rows=a.dense_rows, temperatures=temps)
            thr_s = calibrate_thresholds(co.all_crit.cpu().numpy().astype(np.float64), a.release)
            del cstream, cb, co
        stream = pkg.synth.RawDocumentStream(cfg, a.stream_docs, seed=a.seed + 77, text_len=T)
        feeder = pkg.feed.DeviceFeeder(stream, batch_size=B, size=cfg.input_size, max_length=T, device=dev)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        srows = []
        for batch in feeder:
            o = eng.forward(batch["input_ids"], batch["attention_mask"], batch["bbox"], batch["pixel_values"], thresholds=thr_s,
                            dense_rows=a.dense_rows, temperatures=temps, whole_layers=a.whole_layers, probe_always=a.probe_always,
                            xprobe=bool(a.xprobe))
            srows.append(pkg.dist.pack_results(o.logits, o.exit_layer, o.confidence))
        srows = torch.cat(srows, dim=0)
        torch.cuda.synchronize()
        sdt = time.perf_counter() - t1
        eng.check()                                  # every forward of the stream has been looked at (errors are kept per forward)
        sex = pkg.dist.unpack_results(srows)[1].cpu().numpy().astype(np.int64)
        mel_s, mel_v = float(layer_of_exit[sex].mean()), float(layer_of_exit[exits].mean())
        line["feed_inclusive_docs_per_sec"] = len(stream) / sdt
        line["feed_inclusive"] = {"docs": len(stream), "distinct_documents": True, "seconds": sdt,
                                  "h2d_bytes_per_doc": feeder.bytes_h2d / max(1, len(stream)),
                                  "mean_exit_layer": mel_s,
                                  "exit_distribution": {str(int(layer_of_exit[e])): float((sex == e).mean()) for e in range(len(layer_of_exit))},
                                  "thresholds": [round(float(t), 6) for t in thr_s[:-1]],
                                  "thresholds_calibrated_on": ("--thresholds" if a.thresholds else
                                                               f"{B} documents of another RawDocumentStream (seed {a.seed + 78}), release {a.release} per exit: "
                                                               "the procedure of the resident run applied to the stream's own page statistics"),
                                  "comparable_with_value": bool(abs(mel_s - mel_v) <= 0.1),
                                  "mean_exit_layer_of_value": mel_v,
                                  "why": "both runs release the same fraction per exit under thresholds calibrated on an unseen batch of their OWN generator; "
                                         "comparable_with_value says whether the two mean exit layers ended within 0.1 of each other (equal work per document "
                                         "to ~1 %), the difference between the rates is then host packing + PCIe + device preprocessing",
                                  "what": "RawDocumentStream (1000x762 uint8 pages, ragged ids/boxes) -> DeviceFeeder (pinned double "
                                          "buffer, one async H2D copy per batch, resize/normalise/pad on a side stream) -> ee_forward; "
                                          "host packing, PCIe and preprocessing are inside the clock, page synthesis is not"}

    if rank == 0 and world == 1 and a.cpu_docs != 0 and not stub:
        # ---- CPU baseline (N = 1 runs only): the oracle = the reference's semantics (every layer, every exit, simulated policy) ------
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
        n = a.cpu_docs if a.cpu_docs > 0 else int(min(32, max(2, round(12.0 / max(per_doc, 1e-3)))))
        n = min(n, B)
        t1 = time.perf_counter()
        stores = [r0["logits_store"]]
        for i in range(1, n):
            stores.append(tor.forward_all({k: v[i:i + 1] for k, v in docs.items()}, ee["exits"], strategy=strat)["logits_store"])
        store = np.concatenate(stores, axis=1)
        if temps is not None:
            store = oracle.temperature_scale(store, temps)
        ex_cpu, pred_cpu, _ = oracle.policy_scan(store, thr)
        cpu_dt = per_doc + (time.perf_counter() - t1)
        # best batch size: the same restatement on batches of 4 and 16 of the same documents, bounded to a few seconds each
        best = {"B": 1, "docs_per_sec": n / cpu_dt}
        tried = {1: n / cpu_dt}
        if not beit:
            for bb_ in (4, 16):
                if bb_ > min(n, B) or per_doc * bb_ > 12.0:
                    continue
                t1 = time.perf_counter()
                tor.forward_all({k: v[:bb_] for k, v in docs.items()}, ee["exits"], strategy=strat)
                r_ = bb_ / (time.perf_counter() - t1)
                tried[bb_] = r_
                if r_ > best["docs_per_sec"]:
                    best = {"B": bb_, "docs_per_sec": r_}
        # the build's own C / OpenMP restatement, on two documents of the same batch
        c_port = None
        if not beit:
            ocm = importlib.import_module("oracle.ee_oracle_c")
            if ocm.available():
                os.environ["OMP_NUM_THREADS"] = str(cores)
                co = ocm.COracle(cfg, W)
                nc = min(2, n)
                t1 = time.perf_counter()
                rc_ = [co.forward_all({k: v[i:i + 1] for k, v in docs.items()}, ee["exits"], strategy=strat)["logits_store"] for i in range(nc)]
                c_dt = time.perf_counter() - t1
                c_store = np.concatenate(rc_, axis=1)
                if temps is not None:
                    c_store = oracle.temperature_scale(c_store, temps)
                c_port = {"value": nc / c_dt, "unit": "docs/s", "cores": cores, "docs": nc,
                          "what": "oracle/ee_oracle_c.c: plain C + OpenMP float32 restatement (row-blocked dot products, no BLAS), B=1",
                          "max_abs_dlogit_vs_torch_port": float(np.abs(c_store - store[:, :nc]).max())}
        line["cpu_baseline"] = {"value": n / cpu_dt, "unit": "docs/s", "cores": cores, "kind": "port",
                                "sample": f"{n} documents of the same batch, B=1 per forward (reference default "
                                          f"eval_batch_size=1), full depth + all exits + simulated policy, "
                                          f"{'numpy' if beit else 'torch-CPU'} float32 restatement on {cores} host threads",
                                "best_B": best, "docs_per_sec_by_batch_size": {str(k): round(v, 3) for k, v in tried.items()},
                                "c_openmp_port": c_port}
        g_ex = out0.exit_layer.cpu().numpy()[:n]
        g_lg = out0.logits.cpu().numpy()[:n]
        if g_ex.shape[0] == n:
            line["parity_vs_cpu_sample"] = {"docs": n, "exit_index_equal": bool(np.array_equal(g_ex, ex_cpu)),
                                            "max_abs_dlogit": float(np.abs(g_lg - pred_cpu).max())}
    if rank == 0:
        print(json.dumps(line))
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
