"""TEST INFRASTRUCTURE - NOT PRODUCT CODE.

The same CPU restatement as ``oracle/ee_oracle.py`` (``forward_all``), written with torch CPU float32 ops so that the
``cpu_baseline`` leg of ``bench.py`` times the reference's path the way the reference itself runs it on a CPU (PyTorch
kernels, all host threads) instead of numpy's slower elementwise code.  Pinned to the numpy oracle (and through it to the
golden vectors) by ``tests/test_oracle_golden.py::test_torch_oracle_matches_numpy_oracle``.  Only tests/ and bench.py's
cpu_baseline may import it.  Line references: see ee_oracle.py (identical structure).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Union

import numpy as np
import torch
import torch.nn.functional as F

from . import ee_oracle as O


def _t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


class TorchOracle:
    def __init__(self, cfg, W: Dict[str, np.ndarray]):
        self.cfg = cfg
        self.W = {k: _t(v) for k, v in W.items()}
        self.lut1 = _t(O.bucket_lut(1023, cfg.rel_pos_bins, cfg.max_rel_pos).astype(np.int64))
        self.lut2 = _t(O.bucket_lut(1023, cfg.rel_2d_pos_bins, cfg.max_rel_2d_pos).astype(np.int64))

    def head(self, x, name):
        W = self.W
        if f"{name}.dense.weight" in W:
            x = torch.tanh(F.linear(x, W[f"{name}.dense.weight"], W[f"{name}.dense.bias"]))
        return F.linear(x, W[f"{name}.out_proj.weight"], W[f"{name}.out_proj.bias"])

    @torch.no_grad()
    def forward_all(self, batch: Dict[str, np.ndarray], exits: Sequence[Union[str, int]], strategy: str = "ramp"):
        cfg, W = self.cfg, self.W
        p = "layoutlmv3."
        ids, bbox, pix = _t(batch["input_ids"]), _t(batch["bbox"]), _t(batch["pixel_values"])
        B, T = ids.shape
        am = _t(batch["attention_mask"]) if batch.get("attention_mask") is not None else torch.ones((B, T), dtype=torch.int64)
        emb_exits, enc_exits = O.split_exits(exits)
        ex, gi = [], []
        # A1
        vis = F.conv2d(pix, W[p + "patch_embed.proj.weight"], W[p + "patch_embed.proj.bias"], stride=cfg.patch_size)
        vis = vis.flatten(2).transpose(1, 2)
        vis = torch.cat([W[p + "cls_token"].expand(B, -1, -1), vis], 1) + W[p + "pos_embed"]
        vis = F.layer_norm(vis, (cfg.hidden_size,), W[p + "norm.weight"], W[p + "norm.bias"], 1e-6)
        Pv = vis.shape[1]
        if "vision_avg" in emb_exits:
            x = vis.mean(1); ex.append(self.head(x, p + "vision_exit_embeddings")); gi.append(x)
        # A2
        e = p + "embeddings."
        m = (ids != cfg.pad_token_id).long()
        pid = torch.cumsum(m, 1) * m + cfg.pad_token_id
        hi = cfg.max_2d_position_embeddings - 1
        X, Y = W[e + "x_position_embeddings.weight"], W[e + "y_position_embeddings.weight"]
        sp = torch.cat([X[bbox[..., 0]], Y[bbox[..., 1]], X[bbox[..., 2]], Y[bbox[..., 3]],
                        W[e + "h_position_embeddings.weight"][(bbox[..., 3] - bbox[..., 1]).clamp(0, hi)],
                        W[e + "w_position_embeddings.weight"][(bbox[..., 2] - bbox[..., 0]).clamp(0, hi)]], -1)
        txt = W[e + "word_embeddings.weight"][ids] + W[e + "token_type_embeddings.weight"][torch.zeros_like(ids)]
        txt = txt + W[e + "position_embeddings.weight"][pid] + sp
        txt = F.layer_norm(txt, (cfg.hidden_size,), W[e + "LayerNorm.weight"], W[e + "LayerNorm.bias"], cfg.layer_norm_eps)
        if "text_avg" in emb_exits:
            x = txt.mean(1); ex.append(self.head(x, p + "text_exit_embeddings")); gi.append(x)
        # A3
        x = torch.cat([txt, vis], 1)
        x = F.layer_norm(x, (cfg.hidden_size,), W[p + "LayerNorm.weight"], W[p + "LayerNorm.bias"], cfg.layer_norm_eps)
        if "text_visual_concat" in emb_exits:
            xm = x.mean(1); ex.append(self.head(xm, p + "concat_exit_embeddings")); gi.append(xm)
        mask = torch.cat([am, torch.ones((B, Pv), dtype=torch.int64)], 1)
        ext = (1.0 - mask[:, None, None, :].float()) * torch.finfo(torch.float32).min
        g = cfg.input_size // cfg.patch_size
        fb = torch.cat([bbox, _t(O.visual_bbox(g))[None].expand(B, -1, -1)], 1)
        fpos = torch.cat([torch.arange(T)[None].expand(B, -1), torch.arange(Pv)[None].expand(B, -1)], 1)
        # A4
        en = p + "encoder."
        def one(coord, lut, table):
            rel = coord[:, None, :] - coord[:, :, None]
            return W[en + table].t()[lut[rel + 1023]].permute(0, 3, 1, 2)
        bias = one(fpos, self.lut1, "rel_pos_bias.weight") + (one(fb[:, :, 0], self.lut2, "rel_pos_x_bias.weight") +
                                                              one(fb[:, :, 3], self.lut2, "rel_pos_y_bias.weight"))
        nh, d = cfg.num_attention_heads, cfg.hidden_size // cfg.num_attention_heads
        sd = math.sqrt(d)
        bias = bias / sd + ext
        k = 0
        for l in range(cfg.num_hidden_layers):
            q = f"{en}layer.{l}."
            def hd(t):
                return t.view(B, -1, nh, d).transpose(1, 2)
            Q = hd(F.linear(x, W[q + "attention.self.query.weight"], W[q + "attention.self.query.bias"]))
            K = hd(F.linear(x, W[q + "attention.self.key.weight"], W[q + "attention.self.key.bias"]))
            V = hd(F.linear(x, W[q + "attention.self.value.weight"], W[q + "attention.self.value.bias"]))
            s = torch.matmul(Q / sd, K.transpose(-1, -2)) + bias
            pr = torch.softmax(s, -1)
            ctx = torch.matmul(pr, V).permute(0, 2, 1, 3).reshape(B, -1, cfg.hidden_size)
            a = F.layer_norm(F.linear(ctx, W[q + "attention.output.dense.weight"], W[q + "attention.output.dense.bias"]) + x,
                             (cfg.hidden_size,), W[q + "attention.output.LayerNorm.weight"], W[q + "attention.output.LayerNorm.bias"],
                             cfg.layer_norm_eps)
            f = F.gelu(F.linear(a, W[q + "intermediate.dense.weight"], W[q + "intermediate.dense.bias"]))
            x = F.layer_norm(F.linear(f, W[q + "output.dense.weight"], W[q + "output.dense.bias"]) + a, (cfg.hidden_size,),
                             W[q + "output.LayerNorm.weight"], W[q + "output.LayerNorm.bias"], cfg.layer_norm_eps)
            if (l + 1) in enc_exits:
                c = x[:, 0, :]
                ex.append(self.head(c, f"{en}early_exits.{k}")); gi.append(c.clone()); k += 1
        logits = self.head(x[:, 0, :], "classifier")
        E, Kc = len(ex), logits.shape[1]
        store = np.zeros((E + 1, B, Kc), dtype=np.float64)
        for j in range(E):
            store[j] = (self.head(gi[j], "classifier") if strategy == "gate" else ex[j]).numpy()
        store[-1] = logits.numpy()
        return {"logits_store": store, "logits": logits.numpy(),
                "exit_logits": np.stack([e.numpy() for e in ex]) if ex else np.zeros((0, B, Kc), np.float32),      # exit_states[j][0]
                "gate_inputs": np.stack([g.numpy() for g in gi]) if gi else np.zeros((0, B, cfg.hidden_size), np.float32)}
