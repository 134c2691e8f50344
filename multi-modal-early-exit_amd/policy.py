"""``Policy`` — same constructor, method names, config keys and return contract as EE/policy.py:7-111, evaluated on
the MI355X (ee_policy_scan) instead of a nested Python loop over N x (E+1) scipy softmaxes.

    policy = Policy(logits=logits, config=config)           # logits: np.ndarray (E+1, N, K)
    exits_store, predictions, exit_distribution = getattr(policy, config["exit_policy"])()     # EE/eval.py:91-98

Returns ``(np.int32 (N,), torch.float64 (N,K) on config["device"], {exit_id: fraction})`` exactly as the reference.
There is no CPU fallback: without the HIP library / a GPU the call raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .engine import _require_torch_cuda, torch


def policy_scan_device(logits, thresholds, device=None, want_conf: bool = False):
    """First exit whose float64 max-softmax is strictly above its threshold, else the last exit.

    ``logits`` (E1,N,K) numpy / torch (any float dtype; evaluated as float64 like the harness' store,
    EE/utils.py:160-164); ``thresholds`` scalar or (E1,).  Returns device tensors (exits int32, predictions float64,
    confidence float64 | None, counts int32)."""
    lib = capi.load()
    dev = _require_torch_cuda(device)
    if isinstance(logits, np.ndarray):
        L = torch.from_numpy(np.ascontiguousarray(logits)).to(dev, dtype=torch.float64)
    else:
        L = logits.to(dev, dtype=torch.float64).contiguous()
    if L.dim() != 3:
        raise ValueError("logits must have shape (num_exits + 1, num_samples, num_labels)")
    E1, N, K = L.shape
    thr = np.broadcast_to(np.asarray(thresholds, dtype=np.float64).reshape(-1), (E1,)) if np.ndim(thresholds) \
        else np.full((E1,), float(thresholds))
    thr_c = (C.c_double * E1)(*[float(t) for t in thr])
    exits = torch.empty((N,), dtype=torch.int32, device=dev)
    pred = torch.empty((N, K), dtype=torch.float64, device=dev)
    conf = torch.empty((N,), dtype=torch.float64, device=dev) if want_conf else None
    counts = torch.zeros((E1,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = lib.ee_policy_scan(C.c_void_p(L.data_ptr()), E1, N, K, thr_c, C.c_void_p(exits.data_ptr()),
                                C.c_void_p(pred.data_ptr()), C.c_void_p(conf.data_ptr()) if conf is not None else None,
                                C.c_void_p(counts.data_ptr()), stream)
    capi.check(rc, None, "ee_policy_scan")
    return exits, pred, conf, counts


class Policy:
    def __init__(self, logits, config) -> None:
        self.logits = logits
        self.config = config

    def _finish(self, thresholds):
        num_exits, num_samples = self.logits.shape[0], self.logits.shape[1]
        exits, pred, _, counts = policy_scan_device(self.logits, thresholds)
        exits_store = exits.cpu().numpy().astype(np.int32)
        tgt = self.config.get("device", "cpu")
        predictions = pred.to(tgt) if str(tgt) != str(pred.device) else pred
        c = counts.cpu().numpy()
        exit_distribution = {exit_id: int(c[exit_id]) / num_samples for exit_id in range(0, num_exits)}
        return exits_store, predictions, exit_distribution

    def max_confidence_global_thresholding_policy(self):
        """EE/policy.py:12-53: one global threshold ``config["exit_threshold"]``."""
        return self._finish(float(self.config["exit_threshold"]))

    def accuracy_calibration_heuristic(self):
        """EE/policy.py:55-111: per-exit thresholds minmax_eps(1 - accuracy/ece)."""
        if "calibration_metrics" not in self.config:
            raise Exception("calibration_metrics not in config -> Set calibrate flag to True")
        num_exits = self.logits.shape[0]
        accuracies = self.config["calibration_metrics"]["accuracy"]
        ece = self.config["calibration_metrics"]["ece"]
        metrics = [1 - (accuracies[i] / ece[i]) for i in range(0, num_exits)]
        epsilon = self.config["epsilon"]
        thresholds = (np.array(metrics) - (np.min(metrics) - epsilon)) / (
            (np.max(metrics) + epsilon) - (np.min(metrics) - epsilon))
        return self._finish(thresholds)
