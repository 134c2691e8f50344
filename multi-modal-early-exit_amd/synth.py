"""Synthetic weights and RVL-CDIP-shaped documents (there are no checkpoints or datasets offline).

* Weights follow the HF parameter names of the reference model tree (EE/models/LayoutLMv3.py:308-356, 669-694;
  names as listed in EE/models/EELayoutLM_exit_named_parameters-wotherexits.json) and HF's N(0, 0.02) init, LN gamma=1
  beta=0.  Exit-head / classifier ``out_proj`` rows are scaled (``head_gain``) so max-softmax confidences spread over
  (1/K, 1) and every exit fires for a non-trivial share of documents.
* Documents follow the input contract of EE/utils.py:93-98 + EE/data/RVL_CDIP.py:223-246: ``input_ids (B,T) i64``
  (<s>=0, </s>=2, <pad>=1), ``attention_mask (B,T) i64``, ``bbox (B,T,4) i64 in [0,1000]``,
  ``pixel_values (B,3,R,R) f32 in [-1,1]``.

Everything is generated with ``numpy.random.default_rng(seed)`` so the same tensors can be rebuilt anywhere (the GPU
box has no access to fixtures larger than what is committed).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from .config import ModelConfig


def _normal(rng, shape, std=0.02):
    return (rng.standard_normal(size=shape, dtype=np.float32) * np.float32(std)).astype(np.float32)


def make_weights(cfg: ModelConfig, seed: int = 1234, head_gain: float = 30.0, ln_jitter: float = 0.1) -> Dict[str, np.ndarray]:
    """Random-init parameters keyed by HF parameter name (float32, C-contiguous).

    ``ln_jitter`` perturbs LayerNorm gamma/beta and biases away from 1/0 so that a kernel which forgets a bias or a
    beta cannot pass parity by accident.
    """
    rng = np.random.default_rng(seed)
    H, I, K = cfg.hidden_size, cfg.intermediate_size, cfg.num_labels
    ec = cfg.exit_config
    w: Dict[str, np.ndarray] = {}

    def lin(name, out_f, in_f, gain=1.0):
        w[f"{name}.weight"] = _normal(rng, (out_f, in_f)) * np.float32(gain)
        w[f"{name}.bias"] = _normal(rng, (out_f,), 0.02 if ln_jitter else 0.0) * np.float32(gain)

    def ln(name):
        w[f"{name}.weight"] = (1.0 + ln_jitter * rng.standard_normal(H, dtype=np.float32)).astype(np.float32)
        w[f"{name}.bias"] = (ln_jitter * rng.standard_normal(H, dtype=np.float32)).astype(np.float32)

    p = "layoutlmv3."
    w[p + "embeddings.word_embeddings.weight"] = _normal(rng, (cfg.vocab_size, H))
    w[p + "embeddings.word_embeddings.weight"][cfg.pad_token_id] = 0  # nn.Embedding(padding_idx) zeroes the row
    w[p + "embeddings.token_type_embeddings.weight"] = _normal(rng, (cfg.type_vocab_size, H))
    w[p + "embeddings.position_embeddings.weight"] = _normal(rng, (cfg.max_position_embeddings, H))
    w[p + "embeddings.position_embeddings.weight"][cfg.pad_token_id] = 0
    w[p + "embeddings.x_position_embeddings.weight"] = _normal(rng, (cfg.max_2d_position_embeddings, cfg.coordinate_size))
    w[p + "embeddings.y_position_embeddings.weight"] = _normal(rng, (cfg.max_2d_position_embeddings, cfg.coordinate_size))
    w[p + "embeddings.h_position_embeddings.weight"] = _normal(rng, (cfg.max_2d_position_embeddings, cfg.shape_size))
    w[p + "embeddings.w_position_embeddings.weight"] = _normal(rng, (cfg.max_2d_position_embeddings, cfg.shape_size))
    ln(p + "embeddings.LayerNorm")
    w[p + "patch_embed.proj.weight"] = _normal(rng, (H, cfg.num_channels, cfg.patch_size, cfg.patch_size))
    w[p + "patch_embed.proj.bias"] = _normal(rng, (H,))
    w[p + "cls_token"] = _normal(rng, (1, 1, H))
    w[p + "pos_embed"] = _normal(rng, (1, cfg.visual_len, H))
    ln(p + "norm")
    ln(p + "LayerNorm")
    # relative-position bias tables: nn.Linear(bins, heads, bias=False) -> weight (heads, bins); a larger std makes
    # the bias matter in the scores (HF:378-390)
    w[p + "encoder.rel_pos_bias.weight"] = _normal(rng, (cfg.num_attention_heads, cfg.rel_pos_bins), 0.5)
    w[p + "encoder.rel_pos_x_bias.weight"] = _normal(rng, (cfg.num_attention_heads, cfg.rel_2d_pos_bins), 0.5)
    w[p + "encoder.rel_pos_y_bias.weight"] = _normal(rng, (cfg.num_attention_heads, cfg.rel_2d_pos_bins), 0.5)
    for l in range(cfg.num_hidden_layers):
        q = f"{p}encoder.layer.{l}."
        # std 0.05 on Q/K so attention is not uniform (a uniform softmax would hide indexing mistakes)
        lin(q + "attention.self.query", H, H, gain=2.5)
        lin(q + "attention.self.key", H, H, gain=2.5)
        lin(q + "attention.self.value", H, H)
        lin(q + "attention.output.dense", H, H)
        ln(q + "attention.output.LayerNorm")
        lin(q + "intermediate.dense", I, H)
        lin(q + "output.dense", H, I)
        ln(q + "output.LayerNorm")

    out_dim = K if str(ec.encoder_layer_strategy) == "ramp" else 2
    two = ec.exit_head_num_layers == 2

    def head(name, od, gain):
        if two:
            lin(name + ".dense", H, H, gain=2.0)
        lin(name + ".out_proj", od, H, gain=gain)

    emb_names = {"vision_avg": "vision_exit_embeddings", "text_avg": "text_exit_embeddings",
                 "text_visual_concat": "concat_exit_embeddings"}
    for e in ec.embedding_exits:
        # mean-pooled inputs have small variance across documents; more gain keeps their confidences spread
        head(p + emb_names[e], out_dim, head_gain * 4)
    for k, _ in enumerate(ec.encoder_exit_layers):
        head(f"{p}encoder.early_exits.{k}", out_dim, head_gain)
    # final classifier = HF LayoutLMv3ClassificationHead (HF:799-823), always 2-layer
    lin("classifier.dense", H, H, gain=2.0)
    lin("classifier.out_proj", K, H, gain=head_gain)
    return w


def make_weights_beit(cfg: ModelConfig, seed: int = 1234, head_gain: float = 30.0, ln_jitter: float = 0.1) -> Dict[str, np.ndarray]:
    """Random-init BEiT / DiT parameters under the transformers-4.x names of the DiT checkpoints (``beit.encoder.layer.N.
    attention.attention.query`` ...), plus this build's per-layer exit heads ``beit.encoder.early_exits.k``."""
    rng = np.random.default_rng(seed)
    H, I, K = cfg.hidden_size, cfg.intermediate_size, cfg.num_labels
    ec = cfg.exit_config
    w: Dict[str, np.ndarray] = {}

    def lin(name, out_f, in_f, gain=1.0, bias=True):
        w[f"{name}.weight"] = _normal(rng, (out_f, in_f)) * np.float32(gain)
        if bias:
            w[f"{name}.bias"] = _normal(rng, (out_f,)) * np.float32(gain)

    def ln(name):
        w[f"{name}.weight"] = (1.0 + ln_jitter * rng.standard_normal(H, dtype=np.float32)).astype(np.float32)
        w[f"{name}.bias"] = (ln_jitter * rng.standard_normal(H, dtype=np.float32)).astype(np.float32)

    p = "beit."
    w[p + "embeddings.cls_token"] = _normal(rng, (1, 1, H))
    if cfg.use_absolute_position_embeddings:
        w[p + "embeddings.position_embeddings"] = _normal(rng, (1, cfg.visual_len, H))
    w[p + "embeddings.patch_embeddings.projection.weight"] = _normal(rng, (H, cfg.num_channels, cfg.patch_size, cfg.patch_size))
    w[p + "embeddings.patch_embeddings.projection.bias"] = _normal(rng, (H,))
    for l in range(cfg.num_hidden_layers):
        q = f"{p}encoder.layer.{l}."
        lin(q + "attention.attention.query", H, H, gain=2.5)
        lin(q + "attention.attention.key", H, H, gain=2.5, bias=False)        # BeitSelfAttention.key has no bias
        lin(q + "attention.attention.value", H, H)
        lin(q + "attention.output.dense", H, H)
        lin(q + "intermediate.dense", I, H)
        lin(q + "output.dense", H, I)
        ln(q + "layernorm_before")
        ln(q + "layernorm_after")
        if cfg.layer_scale_init_value > 0:
            w[q + "lambda_1"] = (cfg.layer_scale_init_value * (1.0 + 0.3 * rng.standard_normal(H, dtype=np.float32))).astype(np.float32)
            w[q + "lambda_2"] = (cfg.layer_scale_init_value * (1.0 + 0.3 * rng.standard_normal(H, dtype=np.float32))).astype(np.float32)
    ln(p + "pooler.layernorm")
    out_dim = K if str(ec.encoder_layer_strategy) == "ramp" else 2
    for k, _ in enumerate(ec.encoder_exit_layers):
        if ec.exit_head_num_layers == 2:
            lin(f"{p}encoder.early_exits.{k}.dense", H, H, gain=2.0)
        lin(f"{p}encoder.early_exits.{k}.out_proj", out_dim, H, gain=head_gain)
    lin("classifier", K, H, gain=head_gain)
    return w


def make_documents(cfg: ModelConfig, n_docs: int, seed: int = 1234, text_len: int = 512,
                   min_words: int = 16, max_words: Optional[int] = None, labels: bool = True) -> Dict[str, np.ndarray]:
    """RVL-CDIP-shaped synthetic batch (distributions of SURVEY.md section 8d)."""
    rng = np.random.default_rng(seed)
    T, R = text_len, cfg.input_size
    if max_words is None:
        max_words = T - 2
    max_words = min(max_words, T - 2)
    min_words = min(min_words, max_words)
    ids = np.full((n_docs, T), cfg.pad_token_id, dtype=np.int64)
    am = np.zeros((n_docs, T), dtype=np.int64)
    bbox = np.zeros((n_docs, T, 4), dtype=np.int64)
    nw = rng.integers(min_words, max_words + 1, size=n_docs)
    for b in range(n_docs):
        n = int(nw[b])
        ids[b, 0] = 0
        ids[b, 1:n + 1] = rng.integers(3, cfg.vocab_size, size=n)
        ids[b, n + 1] = 2
        am[b, :n + 2] = 1
        x0 = rng.integers(0, 951, size=n)
        y0 = np.sort(rng.integers(0, 951, size=n))  # reading order: roughly top to bottom
        bw = rng.integers(5, 201, size=n)
        bh = rng.integers(5, 41, size=n)
        bbox[b, 1:n + 1, 0] = x0
        bbox[b, 1:n + 1, 1] = y0
        bbox[b, 1:n + 1, 2] = np.minimum(x0 + bw, 1000)
        bbox[b, 1:n + 1, 3] = np.minimum(y0 + bh, 1000)
    # greyscale page replicated to 3 channels: mostly white (+1) with 10-20 % dark strokes, values on the 8-bit grid
    dark = rng.uniform(0.10, 0.20, size=(n_docs, 1, 1))
    u = rng.random(size=(n_docs, R, R), dtype=np.float32)
    grey8 = np.where(u < dark, rng.integers(0, 96, size=(n_docs, R, R)), rng.integers(200, 256, size=(n_docs, R, R)))
    page = ((grey8.astype(np.float32) / np.float32(255.0)) - np.float32(0.5)) / np.float32(0.5)
    pix = np.ascontiguousarray(np.broadcast_to(page[:, None, :, :], (n_docs, cfg.num_channels, R, R))).astype(np.float32)
    out = {"input_ids": ids, "attention_mask": am, "bbox": bbox, "pixel_values": pix}
    if labels:
        out["labels"] = rng.integers(0, cfg.num_labels, size=n_docs).astype(np.int64)
    return out


def make_page_image(seed: int, h: int, w: int, channels: int = 1) -> np.ndarray:
    """Synthetic raw page (uint8, (h,w) greyscale or (h,w,3)): white paper, dark strokes, sensor noise — the input of the
    device-side preprocessing (RVL-CDIP pages are greyscale scans of at most 1000 px)."""
    rng = np.random.default_rng(seed)
    a = np.full((h, w, channels), 255, np.uint8)
    for _ in range(40):
        y0, x0 = int(rng.integers(0, h)), int(rng.integers(0, w))
        a[y0:y0 + int(rng.integers(1, 6)), x0:x0 + int(rng.integers(5, 120))] = int(rng.integers(0, 90))
    a = np.clip(a.astype(np.int32) + rng.integers(-20, 21, a.shape), 0, 255).astype(np.uint8)
    return a[:, :, 0] if channels == 1 else a


class RawDocumentStream:
    """``n_docs`` DISTINCT raw documents for the device-side feed (feed.DeviceFeeder): what a dataset reader hands over
    before any preprocessing — a uint8 greyscale page of RVL-CDIP size (1000 x 762), the ragged token ids ``<s> w1 .. wn </s>``
    with one box per token (distributions of SURVEY.md section 8d), a label.  Token ids / boxes are drawn once, vectorised; a page
    is a window of one of ``n_pages`` pre-drawn scans at a per-document offset (a view: no per-document generation cost, every
    document still gets different pixels), so iterating costs what reading decoded pages from memory would."""

    PAGE_H, PAGE_W, MARGIN = 1000, 762, 96

    def __init__(self, cfg: ModelConfig, n_docs: int, seed: int = 1234, text_len: int = 512, min_words: int = 16, n_pages: int = 64):
        rng = np.random.default_rng(seed)
        self.n = int(n_docs)
        T = text_len
        nw = rng.integers(min(min_words, T - 2), T - 1, size=self.n)             # words per document, U{16 .. T-2}
        cnt = nw + 2
        self.off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
        tot = int(self.off[-1])
        doc = np.repeat(np.arange(self.n), cnt)
        first = np.zeros(tot, bool); first[self.off[:-1]] = True
        last = np.zeros(tot, bool); last[self.off[1:] - 1] = True
        word = ~(first | last)
        ids = rng.integers(3, cfg.vocab_size, size=tot).astype(np.int64)
        ids[first] = 0
        ids[last] = 2
        x0 = rng.integers(0, 951, size=tot)
        y0 = rng.integers(0, 951, size=tot)
        y0 = np.sort(doc * 1024 + np.where(word, y0 + 1, np.where(first, 0, 1023))) - doc * 1024 - 1     # reading order per document
        bw = rng.integers(5, 201, size=tot)
        bh = rng.integers(5, 41, size=tot)
        box = np.stack([x0, y0, np.minimum(x0 + bw, 1000), np.minimum(y0 + bh, 1000)], axis=1).astype(np.int64)
        box[~word] = 0
        self.ids, self.box = ids, box
        self.labels = rng.integers(0, cfg.num_labels, size=self.n).astype(np.int64)
        H, W, M = self.PAGE_H, self.PAGE_W, self.MARGIN
        self.pages = [make_page_image(seed + 7919 * (i + 1), H + M, W + M, 1) for i in range(n_pages)]
        self.page_of = rng.integers(0, n_pages, size=self.n)
        self.dy = rng.integers(0, M, size=self.n)
        self.dx = rng.integers(0, M, size=self.n)

    def __len__(self):
        return self.n

    def sample(self, i: int):
        a, b = int(self.off[i]), int(self.off[i + 1])
        dy, dx = int(self.dy[i]), int(self.dx[i])
        return {"image": self.pages[int(self.page_of[i])][dy:dy + self.PAGE_H, dx:dx + self.PAGE_W],
                "input_ids": self.ids[a:b], "bbox": self.box[a:b], "labels": int(self.labels[i])}

    def __iter__(self):
        for i in range(self.n):
            yield self.sample(i)
