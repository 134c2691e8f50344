"""Soak of what bench.py runs by default (GPU box only): N forwards of 2 x 1024 base-size documents through MicroBatchedEngine (two handles, two HIP
streams, default schedule, X-space probe), with a second batch of another size and a dump-all pass interleaved and no synchronisation between the calls
except every 16th: every repeat must reproduce the first result BIT FOR BIT and no forward may raise a device error flag.    python tools/soak_mb.py [n=120]"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

pkg = importlib.import_module("multi-modal-early-exit_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
B = 2048
cfg = pkg.ModelConfig.base(EE_config=dict(exits=[2, 4, 6, 8, 10], encoder_layer_strategy="ramp"))
eng = pkg.MicroBatchedEngine(cfg, max_docs=B, max_text_len=512)
eng.load_weights(pkg.synth.make_weights(cfg, seed=1234, head_gain=6.0))
dev = eng.device
mk = lambda n, seed: tuple(torch.from_numpy(pkg.synth.make_documents(cfg, n, seed=seed, text_len=512)[k]).to(dev)
                           for k in ("input_ids", "attention_mask", "bbox", "pixel_values"))
a, b = mk(B, 1234), mk(777, 99)
thr = [0.524019, 0.542159, 0.507206, 0.430526, 0.883812, 2.0]
first = eng.forward(*a, thresholds=thr)
ref = tuple(t.clone() for t in (first.logits, first.exit_layer, first.confidence))
torch.cuda.synchronize()
t0 = time.time()
bad = 0
for i in range(N):
    if i % 3 == 1:
        eng.forward(*b, thresholds=[0.4] * 5 + [2.0])
    if i % 11 == 5:
        eng.forward(*b, dump_all=True)
    o = eng.forward(*a, thresholds=thr)
    if i % 16 == 0:
        torch.cuda.synchronize()
    if not (torch.equal(o.logits, ref[0]) and torch.equal(o.exit_layer, ref[1]) and torch.equal(o.confidence, ref[2])):
        bad += 1
        print(f"forward {i}: differs from the first run", flush=True)
torch.cuda.synchronize()
eng.check()
ex = ref[1].cpu().numpy()
print(f"{N} forwards of {B} documents (+ {N // 3} of 777 under other thresholds, {N // 11} dumps) in {time.time() - t0:.0f} s; exits left at "
      f"{np.bincount(ex, minlength=6).tolist()}; mismatches: {bad}")
sys.exit(1 if bad else 0)
