"""Host-side driver of the HIP path: owns an ``ee_handle`` and moves pointers across the C-ABI.

PyTorch is used for device memory (input / output tensors), streams and nothing else; all arithmetic of the path
runs in libmmee_hip.so.  ``EarlyExitEngine.forward`` is the (logits, exit_layer, confidence) contract of the north
star; ``LayoutLMv3EEForSequenceClassification`` (modeling.py) wraps it behind the reference's ``forward`` signature.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from dataclasses import dataclass
from typing import Dict, Mapping, Optional, Sequence, Union

import numpy as np

from . import capi
from .config import ModelConfig

try:  # torch is plumbing (device tensors / streams); importing it must not be the reason the product "works"
    import torch
except Exception as _e:  # pragma: no cover
    torch = None
    _torch_err = _e


@dataclass
class EngineOutput:
    logits: "torch.Tensor"                  # (B,K) float32 — logits at the exit each document left through
    exit_layer: "torch.Tensor"              # (B,)  int32   — index into the exit list, E = final classifier
    confidence: "torch.Tensor"              # (B,)  float32 — criterion at that exit
    all_logits: Optional["torch.Tensor"] = None   # (E+1,B,K) policy logits of every evaluated exit (NaN = not reached)
    all_crit: Optional["torch.Tensor"] = None     # (E+1,B)
    head_logits: Optional["torch.Tensor"] = None  # (E,B,Kh) raw exit-head logits (exit_states[j][0])
    head_crit: Optional["torch.Tensor"] = None    # (E,B)    (exit_states[j][1])
    hidden_cls: Optional["torch.Tensor"] = None   # (L+1,B,H)
    hidden_states: Optional["torch.Tensor"] = None   # (L+1,B,T+Pv,H): the state entering every layer and the last layer's output
    attentions: Optional["torch.Tensor"] = None      # (L,B,heads,T+Pv,T+Pv): attention probabilities of every layer (after the head mask)


def _require_torch_cuda(device=None):
    if torch is None:
        raise capi.MMEEUnavailable(f"PyTorch-ROCm is required for device tensors: {_torch_err}")
    if not torch.cuda.is_available():
        raise capi.MMEEUnavailable("no MI355X visible (torch.cuda.is_available() is False); the HIP path has no CPU fallback")
    return torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")


class EarlyExitEngine:
    """One handle on one GPU.  Not thread-safe (same as the reference's one-model-per-process use)."""

    def __init__(self, cfg: ModelConfig, max_docs: int = 64, max_text_len: int = 512, precision: str = "auto",
                 device=None, xprobe: Optional[bool] = None):
        """``xprobe``: what ``forward(xprobe=None)`` runs at probe-first exit layers.  ``None`` (default): the X-space CLS probe
        (MMEE_FLAG_XPROBE, csrc/xprobe.hip) wherever the library has it -- split precision, LayoutLMv3-base / -large shapes; the C side
        falls back to the K | V probe elsewhere.  It is the faster valid schedule and the one ``bench.py`` measures: exit indices equal,
        logits within the 1e-4 bar, but an exit's row is a re-association of the whole-layer arithmetic, NOT bit-identical to the dump-all
        row.  ``False`` pins the K | V probe, whose rows are bit-identical to whole layers (the parity suite's bit-identity tests ask
        for that)."""
        self.lib = capi.load()
        self.device = _require_torch_cuda(device)
        self.cfg = cfg
        self.exit_config = cfg.exit_config
        ec = self.exit_config
        self.max_docs, self.max_text_len = int(max_docs), int(max_text_len)
        self.E = ec.num_exits
        self.K = cfg.num_labels
        self.Kh = cfg.num_labels if str(ec.encoder_layer_strategy) == "ramp" else 2
        c = capi.EEConfig()
        c.abi_version = capi.ABI_VERSION
        for f in ("hidden_size", "num_hidden_layers", "num_attention_heads", "intermediate_size", "vocab_size",
                  "max_position_embeddings", "type_vocab_size", "pad_token_id", "max_2d_position_embeddings",
                  "coordinate_size", "shape_size", "rel_pos_bins", "max_rel_pos", "rel_2d_pos_bins", "max_rel_2d_pos",
                  "input_size", "patch_size", "num_channels", "num_labels"):
            setattr(c, f, int(getattr(cfg, f)))
        c.layer_norm_eps = float(cfg.layer_norm_eps)
        emb = ec.embedding_exits
        c.n_embedding_exits = len(emb)
        for i, e in enumerate(emb):
            c.embedding_exits[i] = capi.EXIT_KIND[e]
        enc = ec.encoder_exit_layers
        if len(enc) > capi.MAX_ENCODER_EXITS:
            raise ValueError("too many encoder exits")
        c.n_encoder_exits = len(enc)
        for i, l in enumerate(enc):
            c.encoder_exit_layers[i] = int(l)
        c.exit_head_num_layers = int(ec.exit_head_num_layers)
        c.strategy = 0 if str(ec.encoder_layer_strategy) == "ramp" else 1
        c.criterion = ec.inference_strategy.code
        c.max_docs, c.max_text_len = self.max_docs, self.max_text_len
        if precision == "auto":      # split-f16 GEMMs where the shapes allow it (hidden / intermediate sizes multiples of 256)
            precision = "split" if (cfg.hidden_size % 256 == 0 and cfg.intermediate_size % 256 == 0) else "fp32"
        c.precision = {"fp32": 0, "f32": 0, "bf16": 1, "split": 2, "f32_split": 2}[precision]
        self.beit = cfg.arch == "beit"
        self.xprobe_default = (precision in ("split", "f32_split") and not self.beit) if xprobe is None else bool(xprobe)
        c.arch = 1 if self.beit else 0
        c.use_abs_pos = int(cfg.use_absolute_position_embeddings)
        c.layer_scale = int(cfg.layer_scale_init_value > 0)
        c.use_mean_pooling = int(cfg.use_mean_pooling)
        if self.beit:
            c.max_text_len = self.max_text_len = 0
        self.precision = precision          # resolved: "fp32" or "split"
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            capi.check(self.lib.ee_create(C.byref(c), C.byref(self._h)), None, "ee_create")
        self._finalized = False

    # ---- lifetime ------------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.ee_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- parameters ----------------------------------------------------------------------------------------------
    def expected_tensors(self):
        n = self.lib.ee_num_expected_tensors(self._h)
        return [self.lib.ee_expected_tensor_name(self._h, i).decode() for i in range(n)]

    def load_weights(self, weights: Mapping[str, Union[np.ndarray, "torch.Tensor"]], strict: bool = True):
        """Copy parameters (HF names) into the handle.  Extra entries are ignored unless ``strict`` finds one missing."""
        expected = self.expected_tensors()
        missing = [n for n in expected if n not in weights]
        if missing and strict:
            raise KeyError(f"{len(missing)} parameter(s) missing from the checkpoint, e.g. {missing[:4]}")
        with torch.cuda.device(self.device):
            for name in expected:
                if name not in weights:
                    continue
                t = weights[name]
                if torch is not None and isinstance(t, torch.Tensor):
                    t = t.detach()
                    if t.dtype not in (torch.float32, torch.float16, torch.bfloat16):
                        t = t.float()
                    t = t.contiguous()
                    dt = {torch.float32: capi.DT_F32, torch.float16: capi.DT_F16, torch.bfloat16: capi.DT_BF16}[t.dtype]
                    if t.is_cuda:
                        if dt != capi.DT_F32:
                            t = t.float()
                            dt = capi.DT_F32
                        # ee_load_tensor copies on the null stream: whatever produced ``t`` on torch's current stream (the
                        # cast above, or the caller's own kernels) must have finished first
                        torch.cuda.current_stream(t.device).synchronize()
                    shape = (C.c_int64 * t.dim())(*t.shape)
                    rc = self.lib.ee_load_tensor(self._h, name.encode(), C.c_void_p(t.data_ptr()), shape, t.dim(), dt,
                                                 1 if t.is_cuda else 0)
                else:
                    a = np.ascontiguousarray(np.asarray(t), dtype=np.float32)
                    shape = (C.c_int64 * a.ndim)(*a.shape)
                    rc = self.lib.ee_load_tensor(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim,
                                                 capi.DT_F32, 0)
                capi.check(rc, self._h, f"ee_load_tensor({name})")
            capi.check(self.lib.ee_finalize(self._h), self._h, "ee_finalize")
        self._finalized = True

    @classmethod
    def from_pretrained(cls, path: str, max_docs: int = 64, max_text_len: int = 512, precision: str = "auto",
                        device=None, ee_config: Optional[dict] = None) -> "EarlyExitEngine":
        """Load a local HF-format checkpoint directory (config.json with ``EE_config`` + safetensors / .bin), the
        counterpart of ``LayoutLMv3EEForSequenceClassification.from_pretrained`` at EE/configs.py:404-411."""
        cfg = ModelConfig.from_pretrained(path)
        if ee_config:
            cfg.EE_config.update(ee_config)
        eng = cls(cfg, max_docs=max_docs, max_text_len=max_text_len, precision=precision, device=device)
        eng.load_weights(load_checkpoint_tensors(path))
        return eng

    # ---- the hot path --------------------------------------------------------------------------------------------
    def _dev(self, x, dtype, name, required=True):
        if x is None:
            if required:
                raise ValueError(f"{name} is required")
            return None
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(x)
        if not x.is_cuda or x.device != self.device:
            x = x.to(self.device, non_blocking=True)
        if x.dtype != dtype:
            x = x.to(dtype)
        return x.contiguous()

    def forward(self, input_ids=None, attention_mask=None, bbox=None, pixel_values=None, token_type_ids=None,
                position_ids=None, thresholds: Optional[Union[float, Sequence[float]]] = None,
                temperatures: Optional[Sequence[float]] = None, dump_all: bool = False, dense_rows: bool = False,
                want_all: bool = False, want_head: bool = False, want_hidden_cls: bool = False,
                validate: bool = False, whole_layers: bool = False, probe_always: bool = False, xprobe: Optional[bool] = None,
                one_term: bool = False, inputs_embeds=None, want_hidden_states: bool = False, out=None, head_mask=None,
                want_attentions: bool = False, _capture: bool = False) -> EngineOutput:
        """``out``: optional preallocated ``(logits (B,K) f32, exit_layer (B,) i32, confidence (B,) f32)`` device tensors (contiguous; row
        slices of larger tensors qualify) the kernels write into instead of fresh allocations -- MicroBatchedEngine hands each half its slice."""
        if not self._finalized:
            raise capi.MMEEError("load_weights() has not been called")
        R = self.cfg.input_size
        px = self._dev(pixel_values, torch.float32, "pixel_values")
        emb = None
        if self.beit:                                   # image-only: (B,3,R,R) is the whole input
            if inputs_embeds is not None:
                raise ValueError("inputs_embeds: an image-only model has no text embeddings")
            ids = am = bb = tt = ps = None
            B, T = px.shape[0], 0
            if tuple(px.shape) != (B, self.cfg.num_channels, R, R):
                raise ValueError(f"pixel_values must be (B,{self.cfg.num_channels},{R},{R})")
        else:
            if inputs_embeds is not None:
                # EE/models/LayoutLMv3.py:414-417 -> HF:160-199: the rows replace word_embeddings(input_ids).  Without input_ids the position
                # ids are the sequential ones of HF:148-158 (nothing says which position is padding); the C-ABI still wants token ids for
                # its validation, so it gets pad-free dummies.  With BOTH, HF takes the position ids from input_ids and the rows from here.
                emb = self._dev(inputs_embeds, torch.float32, "inputs_embeds")
                if emb.dim() != 3 or emb.shape[2] != self.cfg.hidden_size:
                    raise ValueError(f"inputs_embeds must be (B,T,{self.cfg.hidden_size})")
                if input_ids is None:
                    Bq, Tq = emb.shape[:2]
                    pad = self.cfg.pad_token_id
                    input_ids = torch.full((Bq, Tq), 0 if pad != 0 else 1, dtype=torch.int64, device=self.device)
                    if position_ids is None:
                        position_ids = torch.arange(pad + 1, Tq + pad + 1, dtype=torch.int64, device=self.device).unsqueeze(0).expand(Bq, Tq)
            ids = self._dev(input_ids, torch.int64, "input_ids")
            B, T = ids.shape
            if emb is not None and tuple(emb.shape[:2]) != (B, T):
                raise ValueError("inputs_embeds and input_ids disagree on (B,T)")
            if bbox is None:
                bbox = torch.zeros((B, T, 4), dtype=torch.int64, device=self.device)   # EE/models/LayoutLMv3.py:433-436
            am = self._dev(attention_mask, torch.int64, "attention_mask", required=False)
            bb = self._dev(bbox, torch.int64, "bbox")
            tt = self._dev(token_type_ids, torch.int64, "token_type_ids", required=False)
            ps = self._dev(position_ids, torch.int64, "position_ids", required=False)
            if tuple(bb.shape) != (B, T, 4) or tuple(px.shape) != (B, self.cfg.num_channels, R, R):
                raise ValueError(f"bbox must be (B,T,4) and pixel_values (B,{self.cfg.num_channels},{R},{R}); got "
                                 f"{tuple(bb.shape)} / {tuple(px.shape)}")
        for t, n in ((am, "attention_mask"), (tt, "token_type_ids"), (ps, "position_ids")):
            if t is not None and tuple(t.shape) != (B, T):
                raise ValueError(f"{n} must be (B,T)")
        E, K = self.E, self.K
        if thresholds is None:
            thresholds = self.exit_config.global_threshold
        thr = np.broadcast_to(np.asarray(thresholds, dtype=np.float64).reshape(-1), (E + 1,)).copy() \
            if np.ndim(thresholds) else np.full((E + 1,), float(thresholds))
        thr_c = (C.c_double * (E + 1))(*thr.tolist())
        tmp_c = None
        if temperatures is not None:
            tm = np.asarray(temperatures, dtype=np.float64).reshape(-1)
            if tm.shape[0] != E + 1:
                raise ValueError(f"temperatures must have {E + 1} entries")
            tmp_c = (C.c_double * (E + 1))(*tm.tolist())
        dev = self.device
        if out is not None:
            out_logits, out_exit, out_conf = out
            for t, shp, dt in ((out_logits, (B, K), torch.float32), (out_exit, (B,), torch.int32), (out_conf, (B,), torch.float32)):
                if tuple(t.shape) != shp or t.dtype != dt or not t.is_contiguous() or t.device != dev:
                    raise ValueError(f"out: expected a contiguous {dt} tensor of shape {shp} on {dev}")
        else:
            out_logits = torch.empty((B, K), dtype=torch.float32, device=dev)
            out_exit = torch.empty((B,), dtype=torch.int32, device=dev)
            out_conf = torch.empty((B,), dtype=torch.float32, device=dev)
        nan = float("nan")
        all_logits = torch.full((E + 1, B, K), nan, dtype=torch.float32, device=dev) if want_all else None
        all_crit = torch.full((E + 1, B), nan, dtype=torch.float32, device=dev) if want_all else None
        head_logits = torch.full((E, B, self.Kh), nan, dtype=torch.float32, device=dev) if want_head else None
        head_crit = torch.full((E, B), nan, dtype=torch.float32, device=dev) if want_head else None
        hidden = torch.full((self.cfg.num_hidden_layers + 1, B, self.cfg.hidden_size), nan, dtype=torch.float32,
                            device=dev) if want_hidden_cls else None
        if xprobe is None:
            xprobe = self.xprobe_default
        hs = None
        if want_hidden_states:
            # output_hidden_states of the reference (EE/models/LayoutLMv3.py:182-183, 284-285): dump-all, whole layers; dense rows so that
            # the positions the mask drops hold what the reference computes for them
            if not dump_all:
                raise ValueError("want_hidden_states needs dump_all=True (nobody may leave early)")
            whole_layers, dense_rows, xprobe = True, True, False
            S = T + (self.cfg.input_size // self.cfg.patch_size) ** 2 + 1
            hs = torch.empty((self.cfg.num_hidden_layers + 1, B, S, self.cfg.hidden_size), dtype=torch.float32, device=dev)
        hm = att = None
        if head_mask is not None or want_attentions:
            # head_mask / output_attentions of the reference signature (EE/models/LayoutLMv3.py:382-385, 631-641): side kernels of the dump-all,
            # whole-layers forward (csrc/attention_maps.hip); the fused attention kernels of the hot path never see either
            if self.beit:
                raise NotImplementedError("head_mask / attention maps are built for the LayoutLMv3 layers only")
            if not dump_all:
                raise ValueError("head_mask / want_attentions need dump_all=True (they belong to model.forward, not to the early-exit path)")
            whole_layers, xprobe = True, False
            L_, nh = self.cfg.num_hidden_layers, self.cfg.num_attention_heads
            if head_mask is not None:
                # get_head_mask (transformers 4.26 modeling_utils): (heads,) is broadcast over the layers, (L, heads) is taken as is
                hm = self._dev(head_mask, torch.float32, "head_mask")
                if hm.dim() == 1:
                    hm = hm.unsqueeze(0).expand(L_, -1)
                if tuple(hm.shape) != (L_, nh):
                    raise ValueError(f"head_mask must be ({nh},) or ({L_}, {nh})")
                hm = hm.contiguous()
            if want_attentions:
                dense_rows = True
                S = T + (self.cfg.input_size // self.cfg.patch_size) ** 2 + 1
                att = torch.empty((L_, B, nh, S, S), dtype=torch.float32, device=dev)
        flags = ((capi.FLAG_NO_EXIT if dump_all else 0) | (capi.FLAG_DENSE_ROWS if dense_rows else 0) |
                 (capi.FLAG_WHOLE_LAYERS if whole_layers else 0) | (capi.FLAG_PROBE_ALWAYS if probe_always else 0) |
                 (capi.FLAG_XPROBE if xprobe else 0) | (capi.FLAG_ONE_TERM if one_term else 0))
        # one_term: REPORTED low-precision mode (one f16 MFMA term per MAC instead of three in the layer GEMMs and the attention); outside the
        # 1e-4 bar by construction, exit indices may flip -- bench.py's `lowprec` field, never a result to rely on
        # xprobe: probe-first layers take the CLS context in X space (no Q | K | V for documents that leave); same exits, logits within
        # tolerance, not bit-identical to whole layers
        # whole_layers / probe_always override the handle's schedule for this call (default: every decision layer probed first, or the mask pin_schedule() set)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        if _capture:
            # ee_graph_capture: the same call eagerly (the outputs hold its results), then its launch list as a hipGraph bound to THESE tensors
            if emb is not None or hs is not None or hm is not None or att is not None or validate:
                raise ValueError("capture(): inputs_embeds / hidden states / head_mask / attention maps / validate belong to eager calls")
            gid = C.c_int32(-1)
            with torch.cuda.device(dev):
                cur = torch.cuda.current_stream()
                if getattr(self, "_cap_stream", None) is None:
                    self._cap_stream = torch.cuda.Stream(device=dev)      # the legacy null stream (torch's default) cannot be captured
                self._cap_stream.wait_stream(cur)
                rc = self.lib.ee_graph_capture(self._h, p(ids), p(am), p(bb), p(px), p(tt), p(ps), B, T, thr_c, tmp_c, flags,
                                               p(out_logits), p(out_exit), p(out_conf), p(all_logits), p(all_crit),
                                               p(head_logits), p(head_crit), p(hidden), C.c_void_p(self._cap_stream.cuda_stream), C.byref(gid))
                capi.check(rc, self._h, "ee_graph_capture")
                cur.wait_stream(self._cap_stream)
            res = EngineOutput(out_logits, out_exit, out_conf, all_logits, all_crit, head_logits, head_crit, hidden, None, None)
            ins = dict(input_ids=ids, attention_mask=am, bbox=bb, pixel_values=px, token_type_ids=tt, position_ids=ps)
            return CapturedForward(self, gid.value, {k: v for k, v in ins.items() if v is not None}, res)
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            if emb is not None:
                capi.check(self.lib.ee_set_inputs_embeds(self._h, p(emb)), self._h, "ee_set_inputs_embeds")
            if hs is not None:
                capi.check(self.lib.ee_set_hidden_states_out(self._h, p(hs)), self._h, "ee_set_hidden_states_out")
            if hm is not None:
                capi.check(self.lib.ee_set_head_mask(self._h, p(hm)), self._h, "ee_set_head_mask")
            if att is not None:
                capi.check(self.lib.ee_set_attentions_out(self._h, p(att)), self._h, "ee_set_attentions_out")
            rc = self.lib.ee_forward(self._h, p(ids), p(am), p(bb), p(px), p(tt), p(ps), B, T, thr_c, tmp_c, flags,
                                     p(out_logits), p(out_exit), p(out_conf), p(all_logits), p(all_crit),
                                     p(head_logits), p(head_crit), p(hidden), stream)
        capi.check(rc, self._h, "ee_forward")
        self._keepalive = (ids, am, bb, px, tt, ps, emb, hm)   # borrowed by the enqueued kernels until the stream drains
        if validate:
            self.stage_counts()                        # synchronises; raises on out-of-range inputs
        return EngineOutput(out_logits, out_exit, out_conf, all_logits, all_crit, head_logits, head_crit, hidden, hs, att)

    __call__ = forward

    def capture(self, *args, **kw) -> "CapturedForward":
        """The forward as a captured launch list (ee_graph_capture; round 6): same arguments as ``forward``.  The call runs once eagerly --
        the returned object's ``outputs`` hold its results -- and its launch list is instantiated as a hipGraph bound to the device tensors
        of THIS call: ``captured.inputs`` are static buffers (``captured.inputs["input_ids"].copy_(next_batch)``), ``captured.outputs`` are
        rewritten by every ``captured.launch(thresholds=..., temperatures=...)``.  Thresholds and temperatures are arguments of the launch,
        everything else ((B, T), flags, which optional outputs exist, the pinned exit-layer schedule) is part of the capture.  For the
        reference's operating point (eval_batch_size = 1, EE/configs.py:36; EE/utils.py:169-193): ~185 kernel launches per forward become
        one graph launch.  Replays return the bits of the eager call on the same inputs.  Device tensors of the right dtype are BORROWED as they are
        (no copy, as in ``forward``): hand in clones when the caller's own tensors must not become the graph's buffers."""
        return self.forward(*args, _capture=True, **kw)

    def check(self):
        """Synchronise and raise ``MMEEError`` if the last forward flagged out-of-range inputs or a split-precision overflow
        (``forward`` itself only enqueues; an unchecked error is also reported by the next ``forward`` call)."""
        self.stage_counts()

    # ---- statistics of the last forward (both synchronise) ----------------------------------------------------------
    def stage_counts(self):
        n = self.E + 1
        docs, rows, ns = (C.c_int32 * n)(), (C.c_int32 * n)(), C.c_int32()
        with torch.cuda.device(self.device):
            stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            capi.check(self.lib.ee_last_stage_counts(self._h, docs, rows, n, C.byref(ns), stream), self._h,
                       "ee_last_stage_counts")
        return {"docs": list(docs)[:ns.value], "rows": list(rows)[:ns.value]}

    def profile(self, enable: bool = True):
        """Arm / disarm per-kernel HIP-event timing of the following forward calls."""
        capi.check(self.lib.ee_profile(self._h, 1 if enable else 0), self._h, "ee_profile")

    def profile_read(self):
        """{role: {"symbol", "ms", "launches"}} of the last forward run with profiling armed (synchronises)."""
        out = {}
        buf = C.create_string_buffer(160)
        idx = 0
        while True:
            ms, n = C.c_double(), C.c_int32()
            rc = self.lib.ee_profile_read(self._h, idx, buf, 160, C.byref(ms), C.byref(n))
            if rc != 0:
                break
            role, _, sym = buf.value.decode().partition("|")
            out[role] = {"symbol": sym, "ms": ms.value, "launches": n.value}
            idx += 1
        return out

    def flops(self):
        g, a = C.c_double(), C.c_double()
        with torch.cuda.device(self.device):
            stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            capi.check(self.lib.ee_last_flops(self._h, C.byref(g), C.byref(a), stream), self._h, "ee_last_flops")
        plan = self.layer_plan()
        return {"gemm": g.value, "attention": a.value, "probe": plan["probe_flops"],
                "total": g.value + a.value + plan["probe_flops"]}

    def pin_schedule(self, probe_layers=None, xprobe: Optional[bool] = None):
        """Pin which exit layers are probed first (ee_set_probe_mask).  The DEFAULT schedule probes every layer that ends in a decision and
        never changes by itself: the same call always issues the same launches and returns the same bits (round 5; rounds 2-4 let the
        library choose from whichever earlier forward had finished, a timing-dependent decision).  ``probe_layers``: iterable of 0-based layer
        indices; ``None`` asks the library's cost model which layers pay, judged from the stage populations of the LAST forward, which must
        have been a thresholded one (ee_suggest_probe_mask; synchronises), and pins that; ``False`` returns to the default.  Returns the
        pinned list (None for the default).  The mask is part of the handle's state until changed."""
        if probe_layers is False:
            capi.check(self.lib.ee_set_probe_mask(self._h, 0, 0), self._h, "ee_set_probe_mask")
            self._pinned = False
            return None
        if probe_layers is None:
            xp = self.xprobe_default if xprobe is None else bool(xprobe)
            m = C.c_uint64()
            with torch.cuda.device(self.device):
                stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
                capi.check(self.lib.ee_suggest_probe_mask(self._h, capi.FLAG_XPROBE if xp else 0, C.byref(m), stream), self._h,
                           "ee_suggest_probe_mask")
            probe_layers = [l for l in range(self.cfg.num_hidden_layers) if (m.value >> l) & 1]
        layers = sorted(int(l) for l in probe_layers)
        mask = 0
        for l in layers:
            mask |= 1 << l
        capi.check(self.lib.ee_set_probe_mask(self._h, 1, mask), self._h, "ee_set_probe_mask")
        self._pinned = True
        return layers

    def set_criterion(self, strategy):
        """Exit criterion of every later forward ("max_confidence" / "entropy"; ee_set_criterion).  The reference's driver overrides
        ``model.config.exit_config["inference_strategy"]`` after construction (EE/utils.py:62-78); modeling.py forwards that write here."""
        from .config import EarlyExitInference
        st = strategy if isinstance(strategy, EarlyExitInference) else EarlyExitInference(str(strategy))
        capi.check(self.lib.ee_set_criterion(self._h, st.code), self._h, "ee_set_criterion")
        self.exit_config.inference_strategy = st
        return st

    def clock_stamp(self):
        """Device tensor of capi.CLOCK_STAMP_WORDS int64: one (s_memtime, s_memrealtime) pair per CU as seen by one-wave workgroups enqueued on
        the current stream (ee_clock_stamp).  ``clock_ghz(a, b)`` turns two stamps into the shader clock held between them."""
        t = torch.zeros(capi.CLOCK_STAMP_WORDS, dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            capi.check(self.lib.ee_clock_stamp(C.c_void_p(t.data_ptr()), stream), None, "ee_clock_stamp")
        return t

    @staticmethod
    def clock_ghz(stamp_a, stamp_b):
        """(mean GHz, per-XCD mean GHz list) between two ``clock_stamp()`` results (synchronises through .cpu()): d(s_memtime) /
        d(s_memrealtime) x 0.1 GHz of every CU slot filled in BOTH stamps (the CUs' counters are not aligned with each other, so only
        same-CU differences are taken), averaged per XCD and over the chip."""
        a = stamp_a.cpu().numpy().reshape(8, -1, 2).astype(np.float64)
        b = stamp_b.cpu().numpy().reshape(8, -1, 2).astype(np.float64)
        ok = (a[..., 1] > 0) & (b[..., 1] > a[..., 1])
        if not ok.any():
            return None, []
        ghz = np.where(ok, 0.1 * (b[..., 0] - a[..., 0]) / np.where(ok, b[..., 1] - a[..., 1], 1.0), 0.0)
        per = [float(ghz[x][ok[x]].mean()) for x in range(8) if ok[x].any()]
        return float(ghz[ok].mean()), per

    def layer_plan(self):
        """How the last forward ran each encoder layer (ee_last_layer_plan): rows through Q|K|V, rows through the rest of
        the layer, documents whose CLS row was probed before the layer's decision."""
        L = self.cfg.num_hidden_layers
        q, m, p = (C.c_int32 * L)(), (C.c_int32 * L)(), (C.c_int32 * L)()
        pf = C.c_double()
        with torch.cuda.device(self.device):
            stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            capi.check(self.lib.ee_last_layer_plan(self._h, q, m, p, L, C.byref(pf), stream), self._h, "ee_last_layer_plan")
        return {"rows_qkv": list(q), "rows_main": list(m), "docs_probe": list(p), "probe_flops": pf.value}


class CapturedForward:
    """One captured configuration of ``EarlyExitEngine.forward`` (see ``EarlyExitEngine.capture``)."""

    def __init__(self, engine: EarlyExitEngine, graph_id: int, inputs: Dict[str, "torch.Tensor"], outputs: EngineOutput):
        self.engine, self.graph_id, self.inputs, self.outputs = engine, graph_id, inputs, outputs

    def launch(self, thresholds: Optional[Union[float, Sequence[float]]] = None, temperatures: Optional[Sequence[float]] = None,
               validate: bool = False) -> EngineOutput:
        """Replay on torch's current stream with this launch's thresholds / temperatures; returns ``self.outputs`` (the same tensors every
        time: copy what must outlive the next launch)."""
        eng = self.engine
        E = eng.E
        if thresholds is None:
            thresholds = eng.exit_config.global_threshold
        thr = np.broadcast_to(np.asarray(thresholds, dtype=np.float64).reshape(-1), (E + 1,)) if np.ndim(thresholds) else np.full((E + 1,), float(thresholds))
        thr_c = (C.c_double * (E + 1))(*thr.tolist())
        tmp_c = None
        if temperatures is not None:
            tm = np.asarray(temperatures, dtype=np.float64).reshape(-1)
            if tm.shape[0] != E + 1:
                raise ValueError(f"temperatures must have {E + 1} entries")
            tmp_c = (C.c_double * (E + 1))(*tm.tolist())
        with torch.cuda.device(eng.device):
            stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            capi.check(eng.lib.ee_graph_launch(eng._h, self.graph_id, thr_c, tmp_c, stream), eng._h, "ee_graph_launch")
        if validate:
            eng.stage_counts()
        return self.outputs

    __call__ = launch

    def close(self):
        if self.graph_id >= 0 and getattr(self.engine, "_h", None) is not None and self.engine._h.value:
            self.engine.lib.ee_graph_destroy(self.engine._h, self.graph_id)
        self.graph_id = -1


def load_checkpoint_tensors(path: str) -> Dict[str, "torch.Tensor"]:
    """Read every tensor of a local HF checkpoint directory (safetensors preferred, then pytorch_model.bin)."""
    st = [f for f in sorted(os.listdir(path)) if f.endswith(".safetensors")]
    out: Dict[str, "torch.Tensor"] = {}
    if st:
        from safetensors import safe_open
        for f in st:
            with safe_open(os.path.join(path, f), framework="pt", device="cpu") as fh:
                for k in fh.keys():
                    out[k] = fh.get_tensor(k)
        return out
    bins = [f for f in sorted(os.listdir(path)) if f.endswith(".bin") or f.endswith(".pt")]
    if not bins:
        raise FileNotFoundError(f"no *.safetensors / *.bin under {path}")
    for f in bins:
        out.update(torch.load(os.path.join(path, f), map_location="cpu", weights_only=True))
    return out


def save_checkpoint(path: str, cfg: ModelConfig, weights: Mapping[str, np.ndarray]):
    """Write an HF-format checkpoint directory (config.json with EE_config + model.safetensors)."""
    from safetensors.numpy import save_file
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg.to_hf_dict(), f, indent=2)
    save_file({k: np.ascontiguousarray(v) for k, v in weights.items()}, os.path.join(path, "model.safetensors"))
