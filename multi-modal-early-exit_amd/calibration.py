"""Per-exit temperature scaling on the device: counterpart of EE/generic_scaling.py (``TemperatureScaler``) and of the
loop in ``calibrate()`` (EE/eval.py:277-346) that fits one temperature per exit on validation logits and divides the test
logits by it.

``fit_temperatures`` returns what ``calibrate`` stores in ``config["calibration_metrics"]`` except ``ece``: the reference
computes ECE with a remote metric (``evaluate.load("jordyvl/ece")``, EE/metrics.py:479-498) that cannot be fetched
offline, so it is not restated here.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict

import numpy as np

from . import capi
from .engine import _require_torch_cuda, torch


def fit_temperatures(logits, labels, max_iter: int = 100, device=None) -> Dict[str, np.ndarray]:
    """``logits`` (E1,N,K) validation logits (numpy / torch, evaluated as float64), ``labels`` (N,).  Returns numpy arrays
    ``temperature``, ``nll``, ``accuracy``, ``average_confidence`` (each (E1,)) and ``iterations``."""
    lib = capi.load()
    dev = _require_torch_cuda(device)
    to = lambda x, dt: (torch.from_numpy(np.ascontiguousarray(x)) if isinstance(x, np.ndarray) else x).to(dev, dt).contiguous()
    L, y = to(logits, torch.float64), to(labels, torch.int64).view(-1)
    if L.dim() == 2:
        L = L.unsqueeze(0)
    E1, N, K = L.shape
    if y.shape[0] != N:
        raise ValueError("labels must have one entry per sample")
    if int(y.min()) < 0 or int(y.max()) >= K:
        raise ValueError("labels out of range")
    out = {k: torch.empty((E1,), dtype=torch.float64, device=dev) for k in ("temperature", "nll", "accuracy", "average_confidence")}
    iters = torch.zeros((E1,), dtype=torch.int32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        capi.check(lib.ee_temperature_fit(p(L), p(y), E1, N, K, max_iter, p(out["temperature"]), p(out["nll"]),
                                          p(out["accuracy"]), p(out["average_confidence"]), p(iters), stream), None,
                   "ee_temperature_fit")
    res = {k: v.cpu().numpy() for k, v in out.items()}
    res["iterations"] = iters.cpu().numpy()
    return res


class TemperatureScaler:
    """Same surface as EE/generic_scaling.py:37-111 (``fit``, ``transform``, ``temperature_scale``, ``.temperature``)."""

    def __init__(self, temperature=None):
        self.temperature = np.ones(1) if not temperature else np.ones(1) * temperature

    def fit(self, labels, logits):
        return self.set_temperature(labels, logits)

    def temperature_scale(self, logits):
        return np.asarray(logits) / np.resize(self.temperature, np.asarray(logits).shape)      # generic_scaling.py:54-61

    def transform(self, logits):
        z = self.temperature_scale(logits)
        z = z - z.max(-1, keepdims=True)
        e = np.exp(z)
        return e / e.sum(-1, keepdims=True)

    def set_temperature(self, labels, logits):
        self.temperature = fit_temperatures(np.asarray(logits)[None], labels)["temperature"][:1]
        return self.temperature
