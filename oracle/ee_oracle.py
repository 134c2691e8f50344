"""TEST INFRASTRUCTURE - NOT PRODUCT CODE.

CPU restatement (numpy float32 / float64) of the reference's early-exit evaluation path.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import this module; the product path
(``multi-modal-early-exit_amd``) never does and fails loudly when its HIP library is missing.

Parity pinning: the reference ships no tests / golden vectors for this path, its own EE forward
cannot run on the installed transformers 5.x, and its arithmetic lives in the un-vendored
third-party dependency ``transformers`` (reference pin ``^4.26.0``, pyproject.toml:24; container has 5.15.0).  This
restatement is therefore pinned against outputs of the reference's own importable classes run here
(``LayoutLMv3Exit``, ``max_confidence``, ``entropy`` from EE/models; ``Policy`` from EE/policy.py) composed with the
installed HF ``LayoutLMv3Model`` — see ``tests/golden/make_golden.py`` (generator, runs only where /root/reference
exists) and the committed ``tests/golden/*.npz`` it produced.

Every function cites the reference lines it follows (``EE/...`` = /root/reference/EE/..., ``HF:`` = transformers
models/layoutlmv3/modeling_layoutlmv3.py at 5.15.0).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
from scipy.special import erf as _erf

F32 = np.float32
EMBEDDING_EXITS = ("vision_avg", "text_avg", "text_visual_concat")
_EMB_HEAD = {"vision_avg": "vision_exit_embeddings", "text_avg": "text_exit_embeddings",
             "text_visual_concat": "concat_exit_embeddings"}


# ------------------------------------------------------------------------------------------------------------------
# small ops
# ------------------------------------------------------------------------------------------------------------------
def layer_norm(x: np.ndarray, g: np.ndarray, b: np.ndarray, eps: float) -> np.ndarray:
    """torch.nn.LayerNorm over the last axis, float32 (biased variance)."""
    x = x.astype(F32, copy=False)
    mu = x.mean(axis=-1, keepdims=True, dtype=F32)
    xc = x - mu
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=F32)
    return (xc / np.sqrt(var + F32(eps)) * g + b).astype(F32)


def linear(x: np.ndarray, w: np.ndarray, b: Optional[np.ndarray]) -> np.ndarray:
    """torch.nn.Linear: x @ w.T + b, float32."""
    y = x.astype(F32, copy=False) @ w.T
    if b is not None:
        y = y + b
    return y.astype(F32, copy=False)


def gelu(x: np.ndarray) -> np.ndarray:
    """ACT2FN["gelu"] = exact erf GELU (HF:485-497)."""
    x = x.astype(F32, copy=False)
    return (x * F32(0.5) * (F32(1.0) + _erf(x * F32(1.0 / math.sqrt(2.0))).astype(F32))).astype(F32)


def softmax64(x: np.ndarray, axis: int = -1) -> np.ndarray:
    """scipy.special.softmax on float64 (EE/policy.py:30-32)."""
    x = np.asarray(x, dtype=np.float64)
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True)


# ------------------------------------------------------------------------------------------------------------------
# A4: relative-position buckets                                                              HF:392-413
# ------------------------------------------------------------------------------------------------------------------
def relative_position_bucket(rel: np.ndarray, num_buckets: int, max_distance: int) -> np.ndarray:
    """Bidirectional bucket of HF:392-413 in float32 with truncation to integer, exactly as torch evaluates it:
    ``max_exact + (log(n.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).to(long)``.
    """
    rel = np.asarray(rel, dtype=np.int64)
    nb = num_buckets // 2
    ret = (rel > 0).astype(np.int64) * nb
    n = np.abs(rel)
    max_exact = nb // 2
    is_small = n < max_exact
    nf = np.maximum(n, 1).astype(F32)  # log(0) only occurs on the is_small branch, whose value is discarded
    v = np.log(nf / F32(max_exact)) / F32(math.log(max_distance / max_exact)) * F32(nb - max_exact)
    val_if_large = max_exact + v.astype(F32).astype(np.int64)  # .to(torch.long) truncates toward zero
    val_if_large = np.minimum(val_if_large, nb - 1)
    return ret + np.where(is_small, n, val_if_large)


def bucket_lut(max_delta: int, num_buckets: int, max_distance: int) -> np.ndarray:
    """LUT over delta in [-max_delta, max_delta] -> bucket (uint8).  Index = delta + max_delta."""
    d = np.arange(-max_delta, max_delta + 1, dtype=np.int64)
    return relative_position_bucket(d, num_buckets, max_distance).astype(np.uint8)


def visual_bbox(grid: int, max_len: int = 1000) -> np.ndarray:
    """HF:575-596 ``create_visual_bbox``: cls box [1,1,999,999] then row-major patch boxes (integer trunc division)."""
    xs = (np.arange(0, max_len * (grid + 1), max_len) // grid).astype(np.int64)
    ys = xs.copy()
    x0 = np.tile(xs[:-1], (grid, 1))
    y0 = np.tile(ys[:-1], (grid, 1)).T
    x1 = np.tile(xs[1:], (grid, 1))
    y1 = np.tile(ys[1:], (grid, 1)).T
    vb = np.stack([x0, y0, x1, y1], axis=-1).reshape(-1, 4)
    return np.concatenate([np.array([[1, 1, max_len - 1, max_len - 1]], dtype=np.int64), vb], axis=0)


# ------------------------------------------------------------------------------------------------------------------
# configuration helpers
# ------------------------------------------------------------------------------------------------------------------
def split_exits(exits: Sequence[Union[str, int]]) -> Tuple[List[str], List[int]]:
    """Reference evaluation order: vision_avg, text_avg, text_visual_concat (EE/models/LayoutLMv3.py:465-605), then
    encoder layers ascending (:181-248)."""
    emb = [e for e in EMBEDDING_EXITS if e in exits]
    enc = sorted(int(e) for e in exits if isinstance(e, (int, np.integer)))
    return emb, enc


def exit_head(x: np.ndarray, W: Dict[str, np.ndarray], name: str) -> np.ndarray:
    """``LayoutLMv3Exit.forward`` (EE/models/LayoutLMv3.py:86-93) == HF classification head (HF:799-823) in eval mode:
    out_proj(tanh(dense(x))) when the head has a ``dense`` layer, else out_proj(x)."""
    if f"{name}.dense.weight" in W:
        x = np.tanh(linear(x, W[f"{name}.dense.weight"], W[f"{name}.dense.bias"])).astype(F32)
    return linear(x, W[f"{name}.out_proj.weight"], W[f"{name}.out_proj.bias"])


def max_confidence(logits: np.ndarray) -> np.ndarray:
    """EE/models/EE_modules.py:157-160 (float32 softmax over dim 1, max)."""
    x = logits.astype(F32)
    m = x.max(axis=1, keepdims=True)
    e = np.exp(x - m, dtype=F32)
    return (e / e.sum(axis=1, keepdims=True, dtype=F32)).max(axis=1).astype(F32)


def entropy(logits: np.ndarray) -> np.ndarray:
    """EE/models/EE_modules.py:149-154 — no max shift (overflows for large logits, replicated on purpose)."""
    x = logits.astype(F32)
    with np.errstate(over="ignore", invalid="ignore"):
        ex = np.exp(x, dtype=F32)
        A = ex.sum(axis=1, dtype=F32)
        B = (x * ex).sum(axis=1, dtype=F32)
        return (np.log(A) - B / A).astype(F32)


# ------------------------------------------------------------------------------------------------------------------
# the forward pass
# ------------------------------------------------------------------------------------------------------------------
def position_ids_from_input_ids(input_ids: np.ndarray, pad: int) -> np.ndarray:
    """HF:138-146: cumsum(ids != pad) * (ids != pad) + pad."""
    m = (input_ids != pad).astype(np.int64)
    return np.cumsum(m, axis=1) * m + pad


def text_embeddings(cfg, W, input_ids, bbox, token_type_ids=None, position_ids=None, inputs_embeds=None) -> np.ndarray:
    """A2 — HF:160-199 (+ HF:112-136 spatial concat), LayerNorm eps = layer_norm_eps.  ``inputs_embeds`` (B,T,H) replaces the word rows
    (HF:185-186); without ``input_ids`` the default position ids are the sequential ones of HF:148-158, 174-175."""
    p = "layoutlmv3.embeddings."
    if position_ids is None:
        if input_ids is not None:
            position_ids = position_ids_from_input_ids(input_ids, cfg.pad_token_id)
        else:
            Bq, Tq = inputs_embeds.shape[:2]
            position_ids = np.broadcast_to(np.arange(cfg.pad_token_id + 1, Tq + cfg.pad_token_id + 1, dtype=np.int64)[None], (Bq, Tq))
    if token_type_ids is None:
        token_type_ids = np.zeros(position_ids.shape, dtype=np.int64)
    we = W[p + "word_embeddings.weight"][input_ids] if inputs_embeds is None else np.asarray(inputs_embeds, dtype=F32)
    e = we + W[p + "token_type_embeddings.weight"][token_type_ids]
    e = e + W[p + "position_embeddings.weight"][position_ids]
    X, Y = W[p + "x_position_embeddings.weight"], W[p + "y_position_embeddings.weight"]
    Hh, Ww = W[p + "h_position_embeddings.weight"], W[p + "w_position_embeddings.weight"]
    hi = cfg.max_2d_position_embeddings - 1
    sp = np.concatenate([
        X[bbox[..., 0]], Y[bbox[..., 1]], X[bbox[..., 2]], Y[bbox[..., 3]],
        Hh[np.clip(bbox[..., 3] - bbox[..., 1], 0, hi)], Ww[np.clip(bbox[..., 2] - bbox[..., 0], 0, hi)]], axis=-1)
    e = (e + sp).astype(F32)
    return layer_norm(e, W[p + "LayerNorm.weight"], W[p + "LayerNorm.bias"], cfg.layer_norm_eps)


def image_embeddings(cfg, W, pixel_values) -> np.ndarray:
    """A1 — ``forward_image`` (EE/models/LayoutLMv3.py:358-373; HF:71-83, 603-618): Conv2d(k=s=patch) as a GEMM over
    (c, ky, kx); prepend cls_token; + pos_embed; LayerNorm eps 1e-6."""
    p = "layoutlmv3."
    B, C, R, _ = pixel_values.shape
    P, g = cfg.patch_size, cfg.input_size // cfg.patch_size
    x = pixel_values.astype(F32).reshape(B, C, g, P, g, P).transpose(0, 2, 4, 1, 3, 5).reshape(B, g * g, C * P * P)
    wt = W[p + "patch_embed.proj.weight"].reshape(cfg.hidden_size, C * P * P)
    pe = linear(x, wt, W[p + "patch_embed.proj.bias"])
    cls = np.broadcast_to(W[p + "cls_token"].reshape(1, 1, -1), (B, 1, cfg.hidden_size))
    e = np.concatenate([cls, pe], axis=1) + W[p + "pos_embed"].reshape(1, -1, cfg.hidden_size)
    return layer_norm(e.astype(F32), W[p + "norm.weight"], W[p + "norm.bias"], 1e-6)


def attention_bias(cfg, W, position_ids: np.ndarray, bbox: np.ndarray) -> np.ndarray:
    """A4 — ``_cal_1d_pos_emb`` + ``_cal_2d_pos_emb`` (HF:415-457; called at EE/models/LayoutLMv3.py:170-179):
    rel[b,i,j] = p[j] - p[i]; 1D buckets (rel_pos_bins, max_rel_pos), 2D buckets on x0 and on y1 (!); tables are the
    transposed nn.Linear weights.  Returns rel_pos + rel_2d_pos, shape (B, heads, S, S) float32."""
    e = "layoutlmv3.encoder."
    def one(coord, nb, md, table):
        rel = coord[:, None, :] - coord[:, :, None]
        bk = relative_position_bucket(rel, nb, md)
        return W[e + table].T[bk].transpose(0, 3, 1, 2).astype(F32)
    r1 = one(position_ids, cfg.rel_pos_bins, cfg.max_rel_pos, "rel_pos_bias.weight")
    rx = one(bbox[:, :, 0], cfg.rel_2d_pos_bins, cfg.max_rel_2d_pos, "rel_pos_x_bias.weight")
    ry = one(bbox[:, :, 3], cfg.rel_2d_pos_bins, cfg.max_rel_2d_pos, "rel_pos_y_bias.weight")
    return (r1 + (rx + ry)).astype(F32)


def encoder_layer(cfg, W, l: int, x: np.ndarray, bias: np.ndarray, ext_mask: np.ndarray, head_mask_l: Optional[np.ndarray] = None,
                  probs_out: Optional[list] = None) -> np.ndarray:
    """E1-E3 — one ``LayoutLMv3Layer`` (HF:235-303, 343-368, 485-512).  ``head_mask_l`` (heads,): the layer's slice of ``head_mask``
    (EE/models/LayoutLMv3.py:185, 631: ``get_head_mask`` -> one factor per layer and head; transformers 4.26 ``LayoutLMv3SelfAttention.forward``
    applies it as ``attention_probs = attention_probs * head_mask`` after the dropout, before ``attention_probs @ value``; 5.x dropped the
    argument, so this one line is restated from 4.26 and has no fixture — parity unpinned for ``head_mask``).  ``probs_out``: a list that
    receives the layer's (masked) attention probabilities (B, heads, S, S), what ``output_attentions=True`` returns (:219-220)."""
    q = f"layoutlmv3.encoder.layer.{l}."
    B, S, H = x.shape
    nh, d = cfg.num_attention_heads, cfg.hidden_size // cfg.num_attention_heads
    def heads(t):
        return t.reshape(B, S, nh, d).transpose(0, 2, 1, 3)
    Q = heads(linear(x, W[q + "attention.self.query.weight"], W[q + "attention.self.query.bias"]))
    K = heads(linear(x, W[q + "attention.self.key.weight"], W[q + "attention.self.key.bias"]))
    V = heads(linear(x, W[q + "attention.self.value.weight"], W[q + "attention.self.value.bias"]))
    sd = F32(math.sqrt(d))
    s = (Q / sd) @ K.transpose(0, 1, 3, 2)                     # HF:263
    s = s + bias / sd                                          # HF:265-268
    s = (s + ext_mask).astype(F32)                             # HF:270-272
    # CogView PB-relax softmax, HF:223-233
    alpha = F32(32.0)
    sc = s / alpha
    mx = sc.max(axis=-1, keepdims=True)
    z = (sc - mx) * alpha
    ez = np.exp(z - z.max(axis=-1, keepdims=True), dtype=F32)
    probs = (ez / ez.sum(axis=-1, keepdims=True, dtype=F32)).astype(F32)
    if head_mask_l is not None:
        probs = (probs * np.asarray(head_mask_l, dtype=F32)[None, :, None, None]).astype(F32)
    if probs_out is not None:
        probs_out.append(probs.copy())
    ctx = (probs @ V).transpose(0, 2, 1, 3).reshape(B, S, H)
    a = linear(ctx, W[q + "attention.output.dense.weight"], W[q + "attention.output.dense.bias"])
    a = layer_norm(a + x, W[q + "attention.output.LayerNorm.weight"], W[q + "attention.output.LayerNorm.bias"],
                   cfg.layer_norm_eps)                          # HF:299-303
    f = gelu(linear(a, W[q + "intermediate.dense.weight"], W[q + "intermediate.dense.bias"]))
    f = linear(f, W[q + "output.dense.weight"], W[q + "output.dense.bias"])
    return layer_norm(f + a, W[q + "output.LayerNorm.weight"], W[q + "output.LayerNorm.bias"], cfg.layer_norm_eps)


def forward_all(cfg, W: Dict[str, np.ndarray], batch: Dict[str, np.ndarray], exits: Sequence[Union[str, int]],
                strategy: str = "ramp", criterion: str = "max_confidence", return_hidden_cls: bool = False,
                return_hidden_states: bool = False,
                max_layers: Optional[int] = None, head_mask: Optional[np.ndarray] = None,
                return_attentions: bool = False) -> Dict[str, np.ndarray]:
    """Full-depth forward with every exit evaluated, as the reference does at eval time
    (``LayoutLMv3EEForSequenceClassification.forward`` EE/models/LayoutLMv3.py:696-749, 871-896 ->
    ``LayoutLMv3ModelEE.forward`` :375-665 -> ``LayoutLMv3EncoderEE.forward`` :151-305).

    Returns
      exit_logits   (E, B, K or 2)  float32   exit_states[j][0]
      exit_crit     (E, B)          float32   exit_states[j][1]   (criterion on the exit head's own logits)
      gated_logits  (E, B, K)       float32   classifier(gate_inputs[j])  — gate strategy only (:764-792)
      logits        (B, K)          float32   final classifier
      final_crit    (B,)            float32   exit_criteria[-1] (:871-872)
      logits_store  (E+1, B, K)     float64   what the harness keeps (EE/utils.py:160-193): gated_logits[j] for gates,
                                              exit_states[j][0] for ramps, final logits last
    """
    p = "layoutlmv3."
    ids, bbox, pix = batch.get("input_ids"), batch.get("bbox"), batch["pixel_values"]
    am = batch.get("attention_mask")
    emb_in = batch.get("inputs_embeds")                                         # :414-417
    B, T = ids.shape if ids is not None else emb_in.shape[:2]
    if bbox is None:
        bbox = np.zeros((B, T, 4), dtype=np.int64)                              # :433-436
    if am is None:
        am = np.ones((B, T), dtype=np.int64)
    emb_exits, enc_exits = split_exits(exits)
    crit_fn = max_confidence if criterion == "max_confidence" else entropy
    ex_logits: List[np.ndarray] = []
    gate_inputs: List[np.ndarray] = []

    vis = image_embeddings(cfg, W, pix)                                         # :445
    Pv = vis.shape[1]
    if "vision_avg" in emb_exits:                                               # :465-483
        xin = vis.mean(axis=1, dtype=F32)
        ex_logits.append(exit_head(xin, W, p + _EMB_HEAD["vision_avg"])); gate_inputs.append(xin)
    txt = text_embeddings(cfg, W, ids, bbox, batch.get("token_type_ids"), batch.get("position_ids"), emb_in)   # :511-517
    if "text_avg" in emb_exits:                                                 # :519-534
        xin = txt.mean(axis=1, dtype=F32)
        ex_logits.append(exit_head(xin, W, p + _EMB_HEAD["text_avg"])); gate_inputs.append(xin)
    x = np.concatenate([txt, vis], axis=1)                                      # :550
    mask = np.concatenate([am, np.ones((B, Pv), dtype=np.int64)], axis=1)       # :553
    g = cfg.input_size // cfg.patch_size
    fb = np.concatenate([bbox, np.broadcast_to(visual_bbox(g)[None], (B, Pv, 4))], axis=1)            # :556
    fpos = np.concatenate([np.broadcast_to(np.arange(T)[None], (B, T)),
                           np.broadcast_to(np.arange(Pv)[None], (B, Pv))], axis=1)                    # :559-563
    x = layer_norm(x, W[p + "LayerNorm.weight"], W[p + "LayerNorm.bias"], cfg.layer_norm_eps)         # :565
    if "text_visual_concat" in emb_exits:                                       # :581-605 (mean over ALL positions)
        xin = x.mean(axis=1, dtype=F32)
        ex_logits.append(exit_head(xin, W, p + _EMB_HEAD["text_visual_concat"])); gate_inputs.append(xin)
    ext = ((1.0 - mask[:, None, None, :].astype(F32)) * np.finfo(F32).min).astype(F32)                # :622-624
    bias = attention_bias(cfg, W, fpos, fb)                                     # :170-179
    cls_rows = [x[:, 0, :].copy()]
    all_hidden = [x.copy()] if return_hidden_states else None                   # :182-183 (the state ENTERING each layer)
    L = cfg.num_hidden_layers if max_layers is None else max_layers
    k = 0
    hm = None
    if head_mask is not None:                                                   # get_head_mask: (heads,) -> every layer; (L, heads) as is
        hm = np.asarray(head_mask, dtype=F32)
        hm = np.broadcast_to(hm, (cfg.num_hidden_layers, cfg.num_attention_heads)) if hm.ndim == 1 else hm
    all_probs = [] if return_attentions else None
    for l in range(L):                                                          # :181
        x = encoder_layer(cfg, W, l, x, bias, ext, None if hm is None else hm[l], all_probs)
        cls_rows.append(x[:, 0, :].copy())
        if return_hidden_states:
            all_hidden.append(x.copy())                                         # ... and the last layer's output, :284-285
        if (l + 1) in enc_exits:                                                # :222-248
            xin = x[:, 0, :]
            ex_logits.append(exit_head(xin, W, f"{p}encoder.early_exits.{k}")); gate_inputs.append(xin.copy())
            k += 1
    logits = exit_head(x[:, 0, :], W, "classifier")                             # :730-731
    out: Dict[str, np.ndarray] = {"logits": logits, "final_crit": crit_fn(logits)}
    E = len(ex_logits)
    K = logits.shape[1]
    if E:
        out["exit_logits"] = np.stack(ex_logits)
        out["exit_crit"] = np.stack([crit_fn(z) for z in ex_logits])
    else:
        out["exit_logits"] = np.zeros((0, B, K), F32); out["exit_crit"] = np.zeros((0, B), F32)
    store = np.zeros((E + 1, B, K), dtype=np.float64)
    if strategy == "gate":
        gl = [exit_head(gi, W, "classifier") for gi in gate_inputs]             # :764-782
        out["gated_logits"] = np.stack(gl) if gl else np.zeros((0, B, K), F32)
        for j in range(E):
            store[j] = gl[j]
    else:
        for j in range(E):
            store[j] = ex_logits[j]
    store[-1] = logits                                                          # EE/utils.py:193
    out["logits_store"] = store
    if return_hidden_cls:
        out["hidden_cls"] = np.stack(cls_rows)
    if return_hidden_states:
        out["hidden_states"] = np.stack(all_hidden)                             # (L+1, B, T+Pv, H)
    if return_attentions:
        out["attentions"] = np.stack(all_probs)                                 # (L, B, heads, T+Pv, T+Pv)
    return out


# ------------------------------------------------------------------------------------------------------------------
# C2 / P1 / P2: temperature, policies
# ------------------------------------------------------------------------------------------------------------------
def temperature_scale(logits_store: np.ndarray, temperatures: Sequence[float]) -> np.ndarray:
    """``TemperatureScaler.temperature_scale`` per exit (EE/generic_scaling.py:54-61 applied at EE/eval.py:321-323)."""
    t = np.asarray(temperatures, dtype=np.float64).reshape(-1, 1, 1)
    return np.asarray(logits_store, dtype=np.float64) / t


def policy_scan(logits_store: np.ndarray, thresholds: Union[float, Sequence[float]]):
    """Vectorised statement of the scan both policies share (EE/policy.py:28-45 / :87-104): first exit whose
    max-softmax (float64) is STRICTLY greater than its threshold, else the last exit.
    Returns (exits int32 (N,), predictions float64 (N,K), confidence float64 (N,))."""
    L = np.asarray(logits_store, dtype=np.float64)
    E1, N, K = L.shape
    thr = np.broadcast_to(np.asarray(thresholds, dtype=np.float64).reshape(-1), (E1,)) if np.ndim(thresholds) else \
        np.full((E1,), float(thresholds))
    conf = softmax64(L, axis=-1).max(axis=-1)                 # (E1, N)
    hit = conf > thr[:, None]
    hit[-1, :] = True
    ex = hit.argmax(axis=0).astype(np.int32)
    idx = np.arange(N)
    return ex, L[ex, idx, :], conf[ex, idx]


def exit_distribution(exits: np.ndarray, num_exits_plus_1: int) -> Dict[int, float]:
    """EE/policy.py:48-51."""
    n = len(exits)
    return {e: float(np.count_nonzero(exits == e)) / n for e in range(num_exits_plus_1)}


def heuristic_thresholds(accuracy: Sequence[float], ece: Sequence[float], epsilon: float) -> np.ndarray:
    """``accuracy_calibration_heuristic`` thresholds (EE/policy.py:68-79)."""
    metrics = np.array([1 - (accuracy[i] / ece[i]) for i in range(len(accuracy))])
    return (metrics - (metrics.min() - epsilon)) / ((metrics.max() + epsilon) - (metrics.min() - epsilon))


def policy_loop(logits_store: np.ndarray, thresholds: Union[float, Sequence[float]]):
    """Literal per-sample loop of EE/policy.py:28-45 (small N only) — checks ``policy_scan``."""
    L = np.asarray(logits_store, dtype=np.float64)
    E1, N, K = L.shape
    thr = [float(thresholds)] * E1 if not np.ndim(thresholds) else [float(t) for t in thresholds]
    ex = np.zeros(N, dtype=np.int32)
    pred = np.zeros((N, K), dtype=np.float64)
    for s in range(N):
        for e in range(E1):
            score = softmax64(L[e, s]).max()
            if score > thr[e]:
                ex[s] = e; pred[s] = L[e, s]
                break
            if e == E1 - 1:
                ex[s] = e; pred[s] = L[e, s]
    return ex, pred


# ------------------------------------------------------------------------------------------------------------------
# the early-exit contract the MI355X path returns: (logits, exit_layer, confidence)
# ------------------------------------------------------------------------------------------------------------------
def early_exit(cfg, W, batch, exits, thresholds, temperatures=None, strategy="ramp"):
    """Simulated early exit == reference semantics: dump all exits (``forward_all``), optional per-exit temperature,
    then the policy scan.  A document's decision only depends on its own prefix of exits (EE/policy.py:29-39), so
    this equals a real compute-skipping exit."""
    out = forward_all(cfg, W, batch, exits, strategy=strategy)
    store = out["logits_store"]
    if temperatures is not None:
        store = temperature_scale(store, temperatures)
    ex, pred, conf = policy_scan(store, thresholds)
    return {"exit_layer": ex, "logits": pred, "confidence": conf, "logits_store": store}


# ------------------------------------------------------------------------------------------------------------------
# N3: multi-threshold search                                            EE/thresh.py:55-57, 184-215; EE/large_scale.py:42-96
# ------------------------------------------------------------------------------------------------------------------
def msp_table(logits_store: np.ndarray, references: Optional[np.ndarray] = None):
    """CSF "msp" table (EE/thresh.py:55-57) and per-exit correctness (EE/large_scale.py:92-95)."""
    L = np.asarray(logits_store, dtype=np.float64)
    conf = softmax64(L, axis=-1).max(-1)
    correct = None if references is None else (L.argmax(-1) == np.asarray(references)[None, :]).astype(np.uint8)
    return conf, correct


def threshold_sweep(conf: np.ndarray, correct: np.ndarray, thresholds_2d: np.ndarray):
    """``check_2D_threshold`` + ``evaluate_exit_logits`` (EE/large_scale.py:42-43, 87-96) for every threshold vector:
    exits = (CSF >= thr[:, None]).argmax(0); accuracy = mean(correct[exits, n]); average_exit = mean(exits)."""
    E1, N = conf.shape
    acc, mex, hist = [], [], []
    idx = np.arange(N)
    for thr in np.asarray(thresholds_2d, dtype=np.float64):
        ex = (conf >= thr[:, None]).argmax(0)
        acc.append(correct[ex, idx].mean())
        mex.append(ex.mean())
        hist.append(np.bincount(ex, minlength=E1))
    return np.array(acc), np.array(mex), np.array(hist)


# ------------------------------------------------------------------------------------------------------------------
# N4: temperature fit                                                              EE/generic_scaling.py:64-111
# ------------------------------------------------------------------------------------------------------------------
def nll_at_temperature(logits: np.ndarray, labels: np.ndarray, T: float) -> float:
    z = np.asarray(logits, dtype=np.float64) / T
    z = z - z.max(-1, keepdims=True)
    lse = np.log(np.exp(z).sum(-1))
    return float((lse - z[np.arange(len(labels)), labels]).mean())


def fit_temperature(logits: np.ndarray, labels: np.ndarray) -> float:
    """``TemperatureScaler.set_temperature``: L-BFGS-B on the mean NLL of softmax(logits / T) from T = 1, bounds (1e-32, inf)
    (sklearn.log_loss equals this NLL up to its probability clipping)."""
    from scipy.optimize import minimize
    r = minimize(lambda t: nll_at_temperature(logits, labels, float(t[0])), x0=np.ones(1), method="L-BFGS-B",
                 bounds=[(1e-32, None)], options={"ftol": 1e-15, "gtol": 1e-10})
    return float(r.x[0])


# ------------------------------------------------------------------------------------------------------------------
# N2: image preprocessing                     LayoutLMv3 image processor as the reference calls it (EE/data/RVL_CDIP.py:246-262)
# ------------------------------------------------------------------------------------------------------------------
# Third-party boundary: Pillow's Image.resize(resample=BILINEAR) (libImaging/Resample.c), called by the HF LayoutLMv3
# image processor the reference constructs at EE/models/LayoutLMv3.py:674-677.  Pillow is not vendored by the reference;
# the algorithm below is Resample.c's 8-bit path (precompute_coeffs / normalize_coeffs_8bpc /
# ImagingResampleHorizontal_8bpc / Vertical_8bpc) and is pinned against Pillow 12.2 outputs in tests/golden/preprocess.npz.
_PRECISION_BITS = 32 - 8 - 2


def pil_bilinear_coeffs(in_size: int, out_size: int):
    """Per output index: (xmin, xmax, int32 weights[ksize]) exactly as precompute_coeffs + normalize_coeffs_8bpc."""
    scale = in_size / out_size                     # double, in0 = 0, in1 = in_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale                    # bilinear filter support = 1
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.zeros(ksize, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            a = (x + xmin - center + 0.5) * ss
            a = -a if a < 0.0 else a
            v = 1.0 - a if a < 1.0 else 0.0
            w[x] = v
            ww += v
        if ww != 0.0:
            w[:xmax] = w[:xmax] / ww
        for x in range(ksize):
            kk[xx, x] = int(-0.5 + w[x] * (1 << _PRECISION_BITS)) if w[x] < 0 else int(0.5 + w[x] * (1 << _PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def pil_bilinear_resize_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """img (H,W,C) uint8 -> (out_h,out_w,C) uint8: horizontal pass then vertical pass, each rounded to 8 bits
    (ImagingResample with both passes needed; a pass whose size does not change is skipped, as in Resample.c)."""
    img = np.asarray(img, dtype=np.uint8)
    H, W, C = img.shape
    cur = img
    if W != out_w:
        b, kk = pil_bilinear_coeffs(W, out_w)
        tmp = np.zeros((H, out_w, C), dtype=np.uint8)
        for xx in range(out_w):
            xmin, xmax = b[xx]
            acc = (cur[:, xmin:xmin + xmax, :].astype(np.int64) * kk[xx, :xmax][None, :, None]).sum(1) + (1 << (_PRECISION_BITS - 1))
            tmp[:, xx, :] = np.clip(acc >> _PRECISION_BITS, 0, 255)
        cur = tmp
    if H != out_h:
        b, kk = pil_bilinear_coeffs(H, out_h)
        tmp = np.zeros((out_h, cur.shape[1], C), dtype=np.uint8)
        for yy in range(out_h):
            ymin, ymax = b[yy]
            acc = (cur[ymin:ymin + ymax, :, :].astype(np.int64) * kk[yy, :ymax][:, None, None]).sum(0) + (1 << (_PRECISION_BITS - 1))
            tmp[yy] = np.clip(acc >> _PRECISION_BITS, 0, 255)
        cur = tmp
    return cur


def rescale_normalize_lut() -> np.ndarray:
    """uint8 -> float32 exactly as HF rescale (uint8 * (1/255) in float64, cast to float32) then normalize with
    mean = std = 0.5 in float32 (IMAGENET_STANDARD_MEAN/STD of the LayoutLMv3 image processor)."""
    v = (np.arange(256, dtype=np.uint8) * (1 / 255)).astype(np.float32)
    return ((v - np.float32(0.5)) / np.float32(0.5)).astype(np.float32)


def preprocess_image(img: np.ndarray, size: int = 224) -> np.ndarray:
    """(H,W,3) uint8 RGB -> (3,size,size) float32 pixel_values."""
    r = pil_bilinear_resize_u8(img, size, size)
    return rescale_normalize_lut()[r].transpose(2, 0, 1).copy()


# ------------------------------------------------------------------------------------------------------------------
# BASELINE configs[4]: image-only DiT (BEiT architecture) with per-layer exit heads
# ------------------------------------------------------------------------------------------------------------------
# The reference's "dit" branch loads a stock AutoModelForImageClassification (EE/configs.py:429-449) and has NO exit
# heads for it (SURVEY.md section 8d calls config 5 an extrapolation).  The encoder below restates HF BEiT
# (``BEIT:N`` = transformers/models/beit/modeling_beit.py at 5.15.0; parameter names are the 4.x ones of the DiT
# checkpoints); exit k = LayoutLMv3Exit (EE/models/LayoutLMv3.py:86-93) on the CLS row after encoder layer k.
def forward_all_beit(cfg, W: Dict[str, np.ndarray], pixel_values: np.ndarray, exits: Sequence[int], strategy: str = "ramp",
                     criterion: str = "max_confidence", return_hidden_cls: bool = False) -> Dict[str, np.ndarray]:
    p = "beit."
    B, C, R, _ = pixel_values.shape
    P, g = cfg.patch_size, cfg.input_size // cfg.patch_size
    H = cfg.hidden_size
    nh, d = cfg.num_attention_heads, cfg.hidden_size // cfg.num_attention_heads
    x = pixel_values.astype(F32).reshape(B, C, g, P, g, P).transpose(0, 2, 4, 1, 3, 5).reshape(B, g * g, C * P * P)
    x = linear(x, W[p + "embeddings.patch_embeddings.projection.weight"].reshape(H, C * P * P),
               W[p + "embeddings.patch_embeddings.projection.bias"])                                   # BEIT:63-90
    x = np.concatenate([np.broadcast_to(W[p + "embeddings.cls_token"].reshape(1, 1, H), (B, 1, H)), x], axis=1)
    if p + "embeddings.position_embeddings" in W:
        x = x + W[p + "embeddings.position_embeddings"].reshape(1, -1, H)                             # BEIT:168-172
    x = x.astype(F32)
    enc_exits = sorted(int(e) for e in exits)
    crit_fn = max_confidence if criterion == "max_confidence" else entropy
    ex_logits, gate_inputs, cls_rows = [], [], [x[:, 0, :].copy()]
    k = 0
    for l in range(cfg.num_hidden_layers):                                                            # BEIT:406-444
        q = f"{p}encoder.layer.{l}."
        hN = layer_norm(x, W[q + "layernorm_before.weight"], W[q + "layernorm_before.bias"], cfg.layer_norm_eps)
        def heads(t):
            return t.reshape(B, -1, nh, d).transpose(0, 2, 1, 3)
        Q = heads(linear(hN, W[q + "attention.attention.query.weight"], W[q + "attention.attention.query.bias"]))
        K = heads(linear(hN, W[q + "attention.attention.key.weight"], None))                          # no key bias, BEIT:306
        V = heads(linear(hN, W[q + "attention.attention.value.weight"], W[q + "attention.attention.value.bias"]))
        s = (Q @ K.transpose(0, 1, 3, 2)) * F32(d ** -0.5)                                            # BEIT:281-284
        s = s - s.max(axis=-1, keepdims=True)
        es = np.exp(s, dtype=F32)
        pr = (es / es.sum(axis=-1, keepdims=True, dtype=F32)).astype(F32)
        ctx = (pr @ V).transpose(0, 2, 1, 3).reshape(B, -1, H)
        a = linear(ctx, W[q + "attention.output.dense.weight"], W[q + "attention.output.dense.bias"])
        if q + "lambda_1" in W:
            a = W[q + "lambda_1"] * a
        x = (a + x).astype(F32)
        hN = layer_norm(x, W[q + "layernorm_after.weight"], W[q + "layernorm_after.bias"], cfg.layer_norm_eps)
        f = gelu(linear(hN, W[q + "intermediate.dense.weight"], W[q + "intermediate.dense.bias"]))
        f = linear(f, W[q + "output.dense.weight"], W[q + "output.dense.bias"])
        if q + "lambda_2" in W:
            f = W[q + "lambda_2"] * f
        x = (f + x).astype(F32)
        cls_rows.append(x[:, 0, :].copy())
        if (l + 1) in enc_exits:
            xin = x[:, 0, :]
            ex_logits.append(exit_head(xin, W, f"{p}encoder.early_exits.{k}")); gate_inputs.append(xin.copy())
            k += 1
    pooled = layer_norm(x[:, 1:, :].mean(axis=1, dtype=F32), W[p + "pooler.layernorm.weight"], W[p + "pooler.layernorm.bias"],
                        cfg.layer_norm_eps)                                                           # BEIT:563-572
    logits = linear(pooled, W["classifier.weight"], W["classifier.bias"])
    E, Kc = len(ex_logits), logits.shape[1]
    out: Dict[str, np.ndarray] = {"logits": logits, "final_crit": crit_fn(logits)}
    out["exit_logits"] = np.stack(ex_logits) if E else np.zeros((0, B, Kc), F32)
    out["exit_crit"] = np.stack([crit_fn(z) for z in ex_logits]) if E else np.zeros((0, B), F32)
    store = np.zeros((E + 1, B, Kc), dtype=np.float64)
    for j in range(E):
        store[j] = ex_logits[j] if strategy != "gate" else linear(
            layer_norm(gate_inputs[j], W[p + "pooler.layernorm.weight"], W[p + "pooler.layernorm.bias"], cfg.layer_norm_eps),
            W["classifier.weight"], W["classifier.bias"])
    store[-1] = logits
    out["logits_store"] = store
    if return_hidden_cls:
        out["hidden_cls"] = np.stack(cls_rows)
    return out
