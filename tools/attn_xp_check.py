"""Correctness screen of an attention experiment form (diagnostic library, MMEE_ATTN_XP=<bits>): runs 24 documents through every layer
(dump_all) and saves the logits; a second run with another setting compares against the saved file (GPU box only).

    MMEE_ATTN_XP=2 python tools/attn_xp_check.py save /tmp/a.pt && MMEE_ATTN_XP=6 python tools/attn_xp_check.py cmp /tmp/a.pt"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _diag  # noqa: F401  (diagnostic library)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

pkg = importlib.import_module("multi-modal-early-exit_amd")
cfg = pkg.ModelConfig.base(EE_config=dict(exits=[2, 4, 6, 8, 10], encoder_layer_strategy="ramp"))
eng = pkg.EarlyExitEngine(cfg, max_docs=24, max_text_len=512, xprobe=False)
eng.load_weights(pkg.synth.make_weights(cfg, seed=1234, head_gain=6.0))
d = pkg.synth.make_documents(cfg, 24, seed=5, text_len=512)
out = eng.forward(d["input_ids"], d["attention_mask"], d["bbox"], d["pixel_values"], dump_all=True, want_all=True)
logits = torch.nan_to_num(out.all_logits.float(), nan=0.0).cpu()
mode, path = sys.argv[1], sys.argv[2]
if mode == "save":
    torch.save(logits, path)
    print("saved", tuple(logits.shape))
else:
    ref = torch.load(path)
    err = (logits - ref).abs().max().item()
    print(f"max |dlogit| vs saved = {err:.3e}  (XP={os.environ.get('MMEE_ATTN_XP')})")
    sys.exit(0 if err < 1e-5 else 1)
