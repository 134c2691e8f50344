"""Phase shares of the head-pair attention kernel (diagnostic build with in-kernel stamps): run as
`MMEE_ATTN_STAMPS=1 python tools/attn_stamps.py` on the GPU box.  Shares, not times (the stamps fence the phases apart)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _diag  # noqa: F401  (diagnostic library)
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("multi-modal-early-exit_amd")
ee = dict(exits=[2, 4, 6, 8, 10], encoder_layer_strategy="ramp")
cfg = pkg.ModelConfig.base(EE_config=ee)
B = int(os.environ.get("B", "256"))
W = pkg.synth.make_weights(cfg, seed=1234, head_gain=6.0)
eng = pkg.EarlyExitEngine(cfg, max_docs=B, max_text_len=512)
eng.load_weights(W)
_w = os.environ.get("WORDS")
d = pkg.synth.make_documents(cfg, B, seed=5, text_len=512, **({"min_words": int(_w), "max_words": int(_w)} if _w else {}))
args = (d["input_ids"], d["attention_mask"], d["bbox"], d["pixel_values"])
eng.forward(*args, dump_all=True)
torch.cuda.synchronize()
out = (C.c_uint64 * 8)()
lib = pkg.capi.load()
pkg.capi.check(lib.ee_debug_attn_stamps(out), None, "stamps")       # clears after the warm-up
eng.forward(*args, dump_all=True)
pkg.capi.check(lib.ee_debug_attn_stamps(out), None, "stamps")
v = np.array(list(out), dtype=np.float64)
names = ["dma_wait", "queue+doc", "bias_gather", "qk_mfma", "softmax_pv", "epilogue", "item_prologue", "barrier"]
am = d["attention_mask"].sum(1) + 197
items = int((-(-am // 128)).sum()) * cfg.num_attention_heads * cfg.num_hidden_layers
print(f"documents {B}, rows/doc {am.mean():.0f}, items {items} (a stamp = lane 0 of each of an item's 4 waves; idle waves stamp too)")
for n, x in zip(names, v):
    print(f"{n:14s} {x / v.sum():6.1%}   {x:.3e}   {x / (4 * items):9.0f} cycles per wave and item")
