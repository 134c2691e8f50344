"""Two (or more) micro-batches of one forward on side streams: the tail of every kernel of one half overlaps the other half's kernels.

Every kernel of the path is a persistent launch that ends in a partly filled last round of tiles / work items (43, 32 and 11 rounds per
GEMM at the bench shape), and one handle has one forward in flight (its workspace is shared).  ``MicroBatchedEngine`` owns ``n`` handles,
each sized for ``ceil(max_docs / n)`` documents, and runs a batch as ``n`` contiguous slices on ``n`` HIP streams; a document's arithmetic
does not depend on which documents share its launches, so the results are the single-handle results BIT FOR BIT under the same exit-layer
schedule (tests/test_gpu_round5.py::test_micro_batched_engine_is_bit_identical).  Measured on one box: +0.9 % docs/s at two slices, nothing at
three, -3 % at four (profiles/r05_micro_batches_ab.txt); it is what ``bench.py`` runs by default since round 5.

Stream semantics are those of ``EarlyExitEngine.forward``: the call only enqueues; the side streams wait for everything the caller's
current stream holds at the call (inputs, the memory of the output tensors), and the current stream waits for both halves before anything
enqueued after the call runs -- to the caller the forward behaves as if it ran on the current stream.

The reference has no counterpart (batch size 1, one stream: EE/utils.py:169-193).
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np

from .engine import EarlyExitEngine, EngineOutput, torch


class MicroBatchedEngine:
    def __init__(self, cfg, max_docs: int = 64, max_text_len: int = 512, precision: str = "auto", device=None,
                 xprobe: Optional[bool] = None, micro_batches: int = 2):
        if micro_batches < 1:
            raise ValueError("micro_batches must be >= 1")
        self.n = int(micro_batches)
        self.max_docs = int(max_docs)
        per = (self.max_docs + self.n - 1) // self.n
        self.engines: List[EarlyExitEngine] = [EarlyExitEngine(cfg, max_docs=per, max_text_len=max_text_len, precision=precision,
                                                               device=device, xprobe=xprobe) for _ in range(self.n)]
        e0 = self.engines[0]
        self.device, self.cfg, self.exit_config = e0.device, e0.cfg, e0.exit_config
        self.E, self.K, self.Kh, self.precision, self.beit = e0.E, e0.K, e0.Kh, e0.precision, e0.beit
        self.xprobe_default = e0.xprobe_default
        self.max_text_len = e0.max_text_len
        with torch.cuda.device(self.device):
            self.streams = [torch.cuda.Stream(device=self.device) for _ in range(self.n)]
        self._sizes: List[int] = []
        self._ran = [False] * self.n          # handles that have run a forward (check() visits every one of them, not only the last call's)

    # ---- lifetime / parameters -------------------------------------------------------------------------------------------
    def close(self):
        for e in self.engines:
            e.close()

    def load_weights(self, weights, strict: bool = True):
        for e in self.engines:          # every handle holds its own copy (0.84 GB for LayoutLMv3-base with its split planes: 288 GB of HBM)
            e.load_weights(weights, strict=strict)

    def expected_tensors(self):
        return self.engines[0].expected_tensors()

    # ---- the hot path ---------------------------------------------------------------------------------------------------
    def split_sizes(self, B: int) -> List[int]:
        """Contiguous slices, as even as possible, never more than a handle holds; empty slices are dropped."""
        if B < 1:
            raise ValueError(f"B={B}: a forward needs at least one document")
        n = min(self.n, B)
        base, extra = divmod(B, n)
        return [base + (1 if i < extra else 0) for i in range(n)]

    def forward(self, input_ids=None, attention_mask=None, bbox=None, pixel_values=None, token_type_ids=None, position_ids=None,
                inputs_embeds=None, serial: bool = False, **kw) -> EngineOutput:
        """Same arguments and result as ``EarlyExitEngine.forward``.  ``serial``: run the slices one after the other on the CURRENT stream
        (per-kernel HIP-event profiling needs launches that do not overlap; same results)."""
        ref = pixel_values if pixel_values is not None else input_ids
        if ref is None:
            raise ValueError("pixel_values is required")
        B = int(ref.shape[0])
        if B > self.max_docs:
            raise ValueError(f"B={B} exceeds max_docs={self.max_docs}")
        if kw.get("want_hidden_states") or kw.get("out") is not None:
            raise ValueError("want_hidden_states / out are per-handle features: use EarlyExitEngine")
        sizes = self.split_sizes(B)
        self._sizes = sizes
        # validate synchronises (ee_last_stage_counts): done once, AFTER every slice has been enqueued and joined -- inside the loop it would
        # run the slices one after the other
        validate = bool(kw.pop("validate", False))
        dev = self.device
        e0 = self.engines[0]
        to_dev = lambda x, dt, nm: None if x is None else e0._dev(x, dt, nm)
        tens = dict(input_ids=to_dev(input_ids, torch.int64, "input_ids"), attention_mask=to_dev(attention_mask, torch.int64, "attention_mask"),
                    bbox=to_dev(bbox, torch.int64, "bbox"), pixel_values=to_dev(pixel_values, torch.float32, "pixel_values"),
                    token_type_ids=to_dev(token_type_ids, torch.int64, "token_type_ids"),
                    position_ids=to_dev(position_ids, torch.int64, "position_ids"),
                    inputs_embeds=to_dev(inputs_embeds, torch.float32, "inputs_embeds"))
        out_logits = torch.empty((B, self.K), dtype=torch.float32, device=dev)
        out_exit = torch.empty((B,), dtype=torch.int32, device=dev)
        out_conf = torch.empty((B,), dtype=torch.float32, device=dev)
        cur = torch.cuda.current_stream(dev)
        ev_in = None
        if not serial:
            ev_in = torch.cuda.Event()
            ev_in.record(cur)
        parts = []
        lo = 0
        for i, n in enumerate(sizes):
            sl = {k: (v[lo:lo + n] if v is not None else None) for k, v in tens.items()}
            outs = (out_logits[lo:lo + n], out_exit[lo:lo + n], out_conf[lo:lo + n])
            if serial:
                parts.append(self.engines[i].forward(**sl, out=outs, **kw))
            else:
                st = self.streams[i]
                st.wait_event(ev_in)
                with torch.cuda.stream(st):
                    parts.append(self.engines[i].forward(**sl, out=outs, **kw))
                done = torch.cuda.Event()
                done.record(st)
                cur.wait_event(done)
                # the optional outputs were allocated on the side stream: their memory must not be handed out again before the current
                # stream (their reader) has passed this point
                for t in (parts[-1].all_logits, parts[-1].all_crit, parts[-1].head_logits, parts[-1].head_crit, parts[-1].hidden_cls, parts[-1].attentions):
                    if t is not None:
                        t.record_stream(cur)
            self._ran[i] = True
            lo += n
        if validate:
            self.check()
        cat = lambda xs, d: None if xs[0] is None else (xs[0] if len(xs) == 1 else torch.cat(xs, dim=d))
        return EngineOutput(out_logits, out_exit, out_conf, cat([p.all_logits for p in parts], 1), cat([p.all_crit for p in parts], 1),
                            cat([p.head_logits for p in parts], 1), cat([p.head_crit for p in parts], 1),
                            cat([p.hidden_cls for p in parts], 1), None, cat([p.attentions for p in parts], 1))

    __call__ = forward

    # ---- statistics of the last forward: sums over the slices -----------------------------------------------------------------
    def _active(self):
        return self.engines[:len(self._sizes)] if self._sizes else self.engines[:1]

    def check(self):
        """Synchronise and raise for EVERY handle that has run a forward: a pending device error of a slice that the last (smaller) call did
        not use is reported too."""
        for e, ran in zip(self.engines, self._ran):
            if ran:
                e.check()

    def stage_counts(self):
        cs = [e.stage_counts() for e in self._active()]
        return {k: [int(sum(c[k][i] for c in cs)) for i in range(len(cs[0][k]))] for k in ("docs", "rows")}

    def flops(self):
        fs = [e.flops() for e in self._active()]
        return {k: float(sum(f[k] for f in fs)) for k in fs[0]}

    def layer_plan(self):
        ps = [e.layer_plan() for e in self._active()]
        out = {k: [int(sum(p[k][l] for p in ps)) for l in range(len(ps[0][k]))] for k in ("rows_qkv", "rows_main", "docs_probe")}
        out["probe_flops"] = float(sum(p["probe_flops"] for p in ps))
        return out

    def profile(self, enable: bool = True):
        for e in self.engines:
            e.profile(enable)

    def profile_read(self):
        """Sums over the slices (run the profiled forward with ``serial=True``: overlapping launches would be timed twice)."""
        acc = {}
        for e in self._active():
            for role, v in e.profile_read().items():
                a = acc.setdefault(role, {"symbol": v["symbol"], "ms": 0.0, "launches": 0})
                a["ms"] += v["ms"]
                a["launches"] += v["launches"]
        return acc

    def pin_schedule(self, probe_layers=None, xprobe: Optional[bool] = None):
        """One schedule for every slice: ``None`` asks slice 0's handle (cost model on ITS last forward's stage populations)."""
        if probe_layers is None:
            probe_layers = self.engines[0].pin_schedule(None, xprobe=xprobe)
        out = None
        for e in self.engines:
            out = e.pin_schedule(probe_layers)
        self._pinned = probe_layers is not False
        return out

    def set_criterion(self, strategy):
        st = None
        for e in self.engines:
            st = e.set_criterion(strategy)
        return st

    def clock_stamp(self):
        return self.engines[0].clock_stamp()

    clock_ghz = staticmethod(EarlyExitEngine.clock_ghz)
