"""Do two half-batches on two streams beat one batch on one stream?  (tail filling / kernel overlap; GPU box only)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

pkg = importlib.import_module("multi-modal-early-exit_amd")
ee = dict(exits=[2, 4, 6, 8, 10], encoder_layer_strategy="ramp")
cfg = pkg.ModelConfig.base(EE_config=ee)
W = pkg.synth.make_weights(cfg, seed=1234, head_gain=6.0)
B = 512
docs = pkg.synth.make_documents(cfg, B, seed=1234, text_len=512)
dev = torch.device("cuda:0")
T = {k: torch.as_tensor(v).to(dev) for k, v in docs.items()}
thr = [0.520424, 0.550076, 0.50362, 0.428325, 0.883209, 2.0]


def mk(n):
    e = pkg.EarlyExitEngine(cfg, max_docs=n, max_text_len=512)
    e.load_weights(W)
    return e


def run(engines, streams, parts, steps=4):
    def step():
        for e, s, (lo, hi) in zip(engines, streams, parts):
            with torch.cuda.stream(s):
                e.forward(T["input_ids"][lo:hi], T["attention_mask"][lo:hi], T["bbox"][lo:hi], T["pixel_values"][lo:hi], thresholds=thr)
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return B * steps / (time.perf_counter() - t0)


one = mk(B)
print("one stream, B=512:", round(run([one], [torch.cuda.Stream()], [(0, B)]), 1), "docs/s")
one.close()
a, b = mk(B // 2), mk(B // 2)
print("one stream, 2 x 256 back to back:", round(run([a, b], [torch.cuda.current_stream()] * 2, [(0, B // 2), (B // 2, B)]), 1), "docs/s")
print("two streams, 2 x 256:", round(run([a, b], [torch.cuda.Stream(), torch.cuda.Stream()], [(0, B // 2), (B // 2, B)]), 1), "docs/s")
a.close(); b.close()
q = [mk(B // 4) for _ in range(4)]
print("four streams, 4 x 128:", round(run(q, [torch.cuda.Stream() for _ in range(4)], [(i * B // 4, (i + 1) * B // 4) for i in range(4)]), 1), "docs/s")
