"""Timing variants of the head-pair attention kernel (MMEE_ATTN_DBG bits: 1 no bias gathers, 2 no softmax VALU, 4 no LDS-DMA,
8 no P V) — wrong results, timing only.  Prints attention ms per forward from ee_profile for the variant in the environment."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _diag  # noqa: F401  (diagnostic library)
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("multi-modal-early-exit_amd")
ee = dict(exits=[2, 4, 6, 8, 10], encoder_layer_strategy="ramp")
cfg = pkg.ModelConfig.base(EE_config=ee)
B = int(os.environ.get("B", "256"))
W = pkg.synth.make_weights(cfg, seed=1234, head_gain=6.0)
eng = pkg.EarlyExitEngine(cfg, max_docs=B, max_text_len=512)
eng.load_weights(W)
# WORDS=<n>: every document has exactly n words (n + 2 + 197 rows): how the rate depends on the length of an item's key loop
_w = os.environ.get("WORDS")
d = pkg.synth.make_documents(cfg, B, seed=5, text_len=512, **({"min_words": int(_w), "max_words": int(_w)} if _w else {}))
args = (d["input_ids"], d["attention_mask"], d["bbox"], d["pixel_values"])
for _ in range(2):
    eng.forward(*args, dump_all=True)
torch.cuda.synchronize()
eng.profile(True)
eng.forward(*args, dump_all=True)
p = eng.profile_read()
fl = eng.flops()
ms = p["attention"]["ms"]
sw = " ".join(f"{k[10:]}={v}" for k, v in sorted(os.environ.items()) if k.startswith("MMEE_ATTN_")) + (f" WORDS={_w}" if _w else "")
print(f"[{sw}] attention {ms:.2f} ms / forward  ({fl['attention'] / ms / 1e9:.1f} TFLOP/s algorithmic)  B={B}")
