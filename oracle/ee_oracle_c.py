"""TEST INFRASTRUCTURE - NOT PRODUCT CODE.

ctypes wrapper of the C / OpenMP restatement ``oracle/ee_oracle_c.c`` (built by ``make -C oracle``; ``__graft_entry__.build()`` does that).
Same ``forward_all`` contract as ``oracle/ee_oracle.py`` / ``oracle/ee_oracle_torch.py`` for the fields the harness keeps
(``logits_store``, ``logits``, optionally ``hidden_cls``).  Only tests/ and bench.py's cpu_baseline leg may import it.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Sequence, Union

import numpy as np

from . import ee_oracle as O

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libee_oracle_c.so")


class _Cfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("H", "L", "heads", "I", "vocab", "max_pos", "pad_id", "max_2d", "cs", "ss", "bins1", "maxd1",
                                         "bins2", "maxd2", "R", "P", "C", "K", "T")] + [("eps", C.c_float)]


class _Head(C.Structure):
    _fields_ = [("dense_w", C.c_void_p), ("dense_b", C.c_void_p), ("out_w", C.c_void_p), ("out_b", C.c_void_p), ("out_dim", C.c_int32)]


_GLOBAL = ["embeddings.word_embeddings.weight", "embeddings.token_type_embeddings.weight", "embeddings.position_embeddings.weight",
           "embeddings.x_position_embeddings.weight", "embeddings.y_position_embeddings.weight", "embeddings.h_position_embeddings.weight",
           "embeddings.w_position_embeddings.weight", "embeddings.LayerNorm.weight", "embeddings.LayerNorm.bias", "patch_embed.proj.weight",
           "patch_embed.proj.bias", "cls_token", "pos_embed", "norm.weight", "norm.bias", "LayerNorm.weight", "LayerNorm.bias",
           "encoder.rel_pos_bias.weight", "encoder.rel_pos_x_bias.weight", "encoder.rel_pos_y_bias.weight"]
_LAYER = ["attention.self.query.weight", "attention.self.query.bias", "attention.self.key.weight", "attention.self.key.bias",
          "attention.self.value.weight", "attention.self.value.bias", "attention.output.dense.weight", "attention.output.dense.bias",
          "attention.output.LayerNorm.weight", "attention.output.LayerNorm.bias", "intermediate.dense.weight", "intermediate.dense.bias",
          "output.dense.weight", "output.dense.bias", "output.LayerNorm.weight", "output.LayerNorm.bias"]
_KIND = {"vision_avg": -1, "text_avg": -2, "text_visual_concat": -3}


def available() -> bool:
    return os.path.exists(_LIB)


class COracle:
    def __init__(self, cfg, W: Dict[str, np.ndarray], threads: int = 0):
        if not available():
            raise RuntimeError(f"{_LIB} is missing: make -C oracle")
        self.lib = C.CDLL(_LIB)
        self.lib.eec_forward.restype = C.c_int
        self.cfg = cfg
        self.W = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in W.items()}
        if threads:
            os.environ["OMP_NUM_THREADS"] = str(threads)
        self.lut1 = np.ascontiguousarray(O.bucket_lut(1023, cfg.rel_pos_bins, cfg.max_rel_pos), dtype=np.uint8)
        self.lut2 = np.ascontiguousarray(O.bucket_lut(1023, cfg.rel_2d_pos_bins, cfg.max_rel_2d_pos), dtype=np.uint8)
        # the nn.Linear bias tables are (heads, bins): the C code indexes [head][bucket]
        p = "layoutlmv3."
        self._g = (C.c_void_p * len(_GLOBAL))(*[self.W[p + n].ctypes.data for n in _GLOBAL])
        L = cfg.num_hidden_layers
        self._l = (C.c_void_p * (L * len(_LAYER)))(*[self.W[f"{p}encoder.layer.{l}.{n}"].ctypes.data for l in range(L) for n in _LAYER])

    def _head(self, name):
        W = self.W
        h = _Head()
        if f"{name}.dense.weight" in W:
            h.dense_w, h.dense_b = W[f"{name}.dense.weight"].ctypes.data, W[f"{name}.dense.bias"].ctypes.data
        h.out_w, h.out_b = W[f"{name}.out_proj.weight"].ctypes.data, W[f"{name}.out_proj.bias"].ctypes.data
        h.out_dim = W[f"{name}.out_proj.weight"].shape[0]
        return h

    def forward_all(self, batch: Dict[str, np.ndarray], exits: Sequence[Union[str, int]], strategy: str = "ramp",
                    return_hidden_cls: bool = False):
        cfg = self.cfg
        ids = np.ascontiguousarray(batch["input_ids"], dtype=np.int64)
        B, T = ids.shape
        am = batch.get("attention_mask")
        am = np.ascontiguousarray(am if am is not None else np.ones((B, T)), dtype=np.int64)
        bbox = np.ascontiguousarray(batch["bbox"], dtype=np.int64)
        pix = np.ascontiguousarray(batch["pixel_values"], dtype=np.float32)
        emb, enc = O.split_exits(exits)
        kinds = [_KIND[e] for e in emb] + list(enc)
        names = ["layoutlmv3." + O._EMB_HEAD[e] for e in emb] + [f"layoutlmv3.encoder.early_exits.{k}" for k in range(len(enc))]
        heads = (_Head * (len(kinds) + 1))(*([self._head(n) for n in names] + [self._head("classifier")]))
        c = _Cfg()
        for k, v in dict(H=cfg.hidden_size, L=cfg.num_hidden_layers, heads=cfg.num_attention_heads, I=cfg.intermediate_size, vocab=cfg.vocab_size,
                         max_pos=cfg.max_position_embeddings, pad_id=cfg.pad_token_id, max_2d=cfg.max_2d_position_embeddings,
                         cs=cfg.coordinate_size, ss=cfg.shape_size, bins1=cfg.rel_pos_bins, maxd1=cfg.max_rel_pos, bins2=cfg.rel_2d_pos_bins,
                         maxd2=cfg.max_rel_2d_pos, R=cfg.input_size, P=cfg.patch_size, C=cfg.num_channels, K=cfg.num_labels, T=T).items():
            setattr(c, k, int(v))
        c.eps = float(cfg.layer_norm_eps)
        E, K = len(kinds), cfg.num_labels
        store = np.zeros((E + 1, B, K), dtype=np.float64)
        hid = np.zeros((cfg.num_hidden_layers + 1, B, cfg.hidden_size), dtype=np.float32) if return_hidden_cls else None
        kinds_c = (C.c_int32 * max(1, E))(*kinds)
        rc = self.lib.eec_forward(C.byref(c), self._g, self._l, heads, E, kinds_c, 1 if strategy == "gate" else 0,
                                  C.c_void_p(self.lut1.ctypes.data), C.c_void_p(self.lut2.ctypes.data), B, C.c_void_p(ids.ctypes.data),
                                  C.c_void_p(am.ctypes.data), C.c_void_p(bbox.ctypes.data), C.c_void_p(pix.ctypes.data),
                                  C.c_void_p(store.ctypes.data), C.c_void_p(hid.ctypes.data) if hid is not None else None)
        if rc != 0:
            raise RuntimeError(f"eec_forward -> {rc}")
        out = {"logits_store": store, "logits": store[-1].astype(np.float32)}
        if hid is not None:
            out["hidden_cls"] = hid
        return out
