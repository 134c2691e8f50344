"""Where does a pipelined B = 1 forward's time go (GPU box)?  Host enqueue time per forward, GPU time per forward (events), gaps between forwards,
for eager launches and the captured graph, back to back and with a synchronisation behind each."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("multi-modal-early-exit_amd")
B = int(os.environ.get("B", "1"))
N = int(os.environ.get("N", "200"))
cfg = pkg.ModelConfig.base(EE_config=dict(exits=[2, 4, 6, 8, 10], encoder_layer_strategy="ramp"))
eng = pkg.EarlyExitEngine(cfg, max_docs=64, max_text_len=512)
eng.load_weights(pkg.synth.make_weights(cfg, seed=1234, head_gain=6.0))
d = pkg.synth.make_documents(cfg, 256, seed=900, text_len=512)
t = {k: torch.from_numpy(d[k]).cuda() for k in ("input_ids", "attention_mask", "bbox", "pixel_values")}
sl = lambda j: {k: v[(j * B) % (256 - B):(j * B) % (256 - B) + B] for k, v in t.items()}
thr = np.array([0.524, 0.542, 0.507, 0.431, 0.884, 2.0])
sync = torch.cuda.synchronize
for whole in (False, True):
    kw = dict(whole_layers=True) if whole else {}
    for j in range(5):
        eng.forward(**sl(j), thresholds=thr, **kw)
    sync()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(N)]
    host = []
    t0 = time.perf_counter()
    for j in range(N):
        h0 = time.perf_counter()
        ev[j][0].record()
        eng.forward(**sl(j), thresholds=thr, **kw)
        ev[j][1].record()
        host.append(time.perf_counter() - h0)
    t_enq = time.perf_counter() - t0
    sync()
    tot = time.perf_counter() - t0
    gpu = np.array([a.elapsed_time(b) for a, b in ev])
    gap = np.array([ev[j][1].elapsed_time(ev[j + 1][0]) for j in range(N - 1)])
    print(f"whole_layers={whole} pipelined: total {1e3 * tot / N:.3f} ms/fwd, host enqueue {1e3 * np.median(host):.3f} (all enqueued after {1e3 * t_enq / N:.3f}), "
          f"GPU per forward median {np.median(gpu):.3f} p10 {np.percentile(gpu, 10):.3f} p90 {np.percentile(gpu, 90):.3f}, gap between forwards {np.median(gap):.3f}")
    lat = []
    for j in range(40):
        h0 = time.perf_counter()
        eng.forward(**sl(j), thresholds=thr, **kw)
        h1 = time.perf_counter()
        sync()
        lat.append((h1 - h0, time.perf_counter() - h0))
    lat = np.array(lat)
    print(f"   synchronised: enqueue {1e3 * np.median(lat[:, 0]):.3f} ms, to completion {1e3 * np.median(lat[:, 1]):.3f} ms")
