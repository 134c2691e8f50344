"""Per-exit temperature scaling on the device: counterpart of EE/generic_scaling.py (``TemperatureScaler``) and of the
loop in ``calibrate()`` (EE/eval.py:277-346) that fits one temperature per exit on validation logits and divides the test
logits by it.

``fit_temperatures`` returns what ``calibrate`` stores in ``config["calibration_metrics"]``: temperature, accuracy, average
confidence and ECE per exit.  ECE — PARITY UNPINNED: the reference computes it with a remote metric
(``evaluate.load("jordyvl/ece")``, EE/metrics.py:479-498) whose code is neither in the reference tree nor fetchable offline;
``expected_calibration_error`` restates the metric from the arguments the reference passes (equal-mass bins,
``n_bins = min(N - 1, 100)``, ``bin_range = [0, 1]``, upper-edge proxy, p = 1) and is pinned only by hand-computed cases in
tests/test_host.py.  It is host-side numpy on N confidences per exit (a sort and 100 bin means), not part of the hot path.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import warnings

import numpy as np

from . import capi
from .engine import _require_torch_cuda, torch


def _softmax64(z):
    z = np.asarray(z, dtype=np.float64)
    z = z - z.max(-1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(-1, keepdims=True)


def expected_calibration_error(references, predictions, n_bins: int = None, scheme: str = "equal-mass", bin_range=(0.0, 1.0),
                               proxy: str = "upper-edge", p: int = 1) -> float:
    """Top-label calibration error with the arguments of ``ece_logits`` (EE/metrics.py:479-498); PARITY UNPINNED (module
    docstring).  ``predictions`` (N,K): probabilities, or logits (softmaxed when the rows do not sum to 1, as the reference
    does).  Bins over the top-label confidence: equal-mass edges ``sorted_conf[floor(k N / n_bins)]``, k = 0..n_bins-1, plus the
    right end of ``bin_range``; a bin is [edge_k, edge_k+1), the last one closed; empty or duplicate-edge bins carry no weight.
    The calibrated accuracy of a bin is its ``proxy`` ("upper-edge": edge_k+1; "center": the midpoint);
    ECE = (sum_k w_k |acc_k - proxy_k|^p)^(1/p) with w_k = the share of samples in bin k."""
    P = np.asarray(predictions, dtype=np.float64)
    y = np.asarray(references).reshape(-1)
    if P.ndim != 2 or P.shape[0] != y.shape[0]:
        raise ValueError("predictions must be (N,K) with one reference per row")
    N = P.shape[0]
    if not np.isclose(np.sum(P), N):                        # EE/metrics.py:480-481
        P = _softmax64(P)
    if n_bins is None:
        n_bins = min(N - 1, 100)                            # EE/metrics.py:485
    n_bins = max(1, int(n_bins))
    conf = P.max(-1)
    correct = (P.argmax(-1) == y).astype(np.float64)
    if scheme == "equal-mass":
        srt = np.sort(conf)
        edges = np.concatenate([srt[(np.arange(n_bins) * N) // n_bins], [float(bin_range[1])]])
    elif scheme == "equal-range":
        edges = np.linspace(float(bin_range[0]), float(bin_range[1]), n_bins + 1)
    else:
        raise ValueError("scheme must be 'equal-mass' or 'equal-range'")
    idx = np.searchsorted(edges, conf, side="right") - 1     # right-continuous bins, as numpy.digitize
    idx = np.clip(idx, 0, n_bins - 1)                        # the right end belongs to the last bin
    cnt = np.bincount(idx, minlength=n_bins).astype(np.float64)
    acc = np.divide(np.bincount(idx, weights=correct, minlength=n_bins), cnt, out=np.zeros(n_bins), where=cnt > 0)
    target = edges[1:] if proxy == "upper-edge" else 0.5 * (edges[:-1] + edges[1:])
    w = cnt / N
    return float(np.sum(w * np.abs(acc - target) ** p) ** (1.0 / p))


def fit_temperatures(logits, labels, max_iter: int = 100, device=None, with_ece: bool = True) -> Dict[str, np.ndarray]:
    """``logits`` (E1,N,K) validation logits (numpy / torch, evaluated as float64), ``labels`` (N,).  Returns numpy arrays
    ``temperature``, ``nll``, ``accuracy``, ``average_confidence``, ``ece`` (each (E1,); accuracy / confidence / ECE of the SCALED
    input logits — the reference records these three for the scaled TEST logits instead, EE/eval.py:321-337: see ``calibrate``) and
    ``iterations``."""
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
    if with_ece:
        Lh, yh = L.cpu().numpy(), y.cpu().numpy()
        res["ece"] = np.array([expected_calibration_error(yh, Lh[e] / res["temperature"][e]) for e in range(E1)])
    return res


def calibrate(validation_logits, validation_references, test_logits, device=None, metrics_on: Optional[str] = None):
    """The loop of ``calibrate()`` (EE/eval.py:277-346) without its file cache: one temperature per exit fitted on the
    validation logits, the test logits divided by it, and ``calibration_metrics`` = {ece, accuracy, temperature,
    average_confidence} (lists, one entry per exit) — the dictionary ``Policy.accuracy_calibration_heuristic`` reads
    (EE/policy.py:59-79).  Returns (calibrated_test_logits, metrics).

    ``metrics_on`` says which logits the three metrics are taken from:

    * ``None`` (default): ``"reference"`` whenever the reference itself could run, i.e. when the test and validation sets have the same
      number of samples — a drop-in caller gets the thresholds the reference derives; otherwise ``"validation"`` with a warning.
    * ``"validation"`` (a documented DEVIATION from the reference): the scaled VALIDATION logits against the validation references —
      the set the temperatures were fitted on and the labels belong to.
    * ``"reference"``: exactly what EE/eval.py:321-337 computes — ``ece_logits(validation_references, calibrated_logits[i])``,
      ``softmax(calibrated_logits[i]).max(-1).mean()`` and ``mean(calibrated_logits[i].argmax(-1) == validation_references)`` with
      ``calibrated_logits[i]`` the scaled TEST logits, i.e. test predictions scored against validation labels.  That only runs
      when both sets have the same number of samples (it raises otherwise, as numpy does in the reference) and only means
      something when they are the same samples; the thresholds the heuristic derives differ between the two modes.
    """
    if metrics_on is None:
        same = np.asarray(test_logits).shape[1] == np.asarray(validation_references).reshape(-1).shape[0]
        metrics_on = "reference" if same else "validation"
        if not same:
            warnings.warn("calibrate(): the test and validation sets differ in length, so EE/eval.py:321-337 (scaled TEST logits scored "
                          "against the VALIDATION references) cannot be evaluated; the calibration metrics are taken from the scaled "
                          'validation logits instead (metrics_on="validation")', stacklevel=2)
        else:
            # equal lengths do not make them the same samples.  The reference-compatible default stays (a drop-in caller gets the
            # thresholds the reference derives), but never silently: this mode scores TEST predictions against VALIDATION labels.
            warnings.warn('calibrate(): metrics_on resolved to "reference" because the test and validation sets have the same length: as '
                          "EE/eval.py:321-337 does, the scaled TEST logits are scored against the VALIDATION references, which only means "
                          'something when both are the same samples in the same order.  Pass metrics_on="validation" for metrics on the set '
                          'the temperatures were fitted on, or metrics_on="reference" to state the choice and silence this warning',
                          stacklevel=2)
    if metrics_on not in ("validation", "reference"):
        raise ValueError('metrics_on must be None, "validation" or "reference"')
    fit = fit_temperatures(validation_logits, validation_references, device=device, with_ece=metrics_on == "validation")
    T = fit["temperature"]
    cal = np.asarray(test_logits, dtype=np.float64) / T[:, None, None]
    if metrics_on == "reference":
        refs = np.asarray(validation_references).reshape(-1)
        if cal.shape[1] != refs.shape[0]:
            raise ValueError(f"metrics_on='reference' scores the {cal.shape[1]} scaled test logits against the {refs.shape[0]} validation "
                             "references (EE/eval.py:327-337): the two sets must have the same length")
        z = cal - cal.max(-1, keepdims=True)
        sm = np.exp(z)
        sm /= sm.sum(-1, keepdims=True)
        ece = [expected_calibration_error(refs, cal[e]) for e in range(cal.shape[0])]
        acc = [float(np.mean(cal[e].argmax(-1) == refs)) for e in range(cal.shape[0])]
        conf = [float(sm[e].max(-1).mean()) for e in range(cal.shape[0])]
    else:
        ece, acc, conf = fit["ece"], fit["accuracy"], fit["average_confidence"]
    metrics = {"ece": [float(v) for v in ece], "accuracy": [float(v) for v in acc],
               "temperature": [float(v) for v in T], "average_confidence": [float(v) for v in conf]}
    return cal, metrics


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
