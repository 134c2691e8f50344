"""Device-side multi-threshold search: the counterpart of EE/thresh.py:184-215 / EE/large_scale.py:42-128.

The reference builds a CSF table ``msp = max softmax`` of the dumped logits once, derives candidate threshold vectors
from its percentiles, and then, for every threshold vector, computes ``(CSF >= thr[:, None]).argmax(0)`` and the
accuracy / mean exit of the induced exits in an 8-process CPU pool.  Here the table stays in HBM and one workgroup
handles one threshold vector.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .engine import _require_torch_cuda, torch


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def msp_table(logits, references=None, device=None):
    """(conf float64 (E1,N), correct uint8 (E1,N) | None) on the device from logits (E1,N,K)."""
    lib = capi.load()
    dev = _require_torch_cuda(device)
    L = (torch.from_numpy(np.ascontiguousarray(logits)) if isinstance(logits, np.ndarray) else logits).to(dev, torch.float64).contiguous()
    E1, N, K = L.shape
    conf = torch.empty((E1, N), dtype=torch.float64, device=dev)
    refs = corr = None
    if references is not None:
        refs = (torch.from_numpy(np.ascontiguousarray(references)) if isinstance(references, np.ndarray) else references).to(dev, torch.int64).contiguous()
        corr = torch.empty((E1, N), dtype=torch.uint8, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    with torch.cuda.device(dev):
        capi.check(lib.ee_msp_table(p(L), p(refs), E1, N, K, p(conf), p(corr), _stream()), None, "ee_msp_table")
    return conf, corr


def threshold_sweep(conf, correct, thresholds, want_hist: bool = False, device=None):
    """For each threshold vector v: exits = (conf >= thr[v][:, None]).argmax(0); returns device tensors
    ``(accuracy (V,), mean_exit (V,), hist (V,E1) | None)``."""
    lib = capi.load()
    dev = _require_torch_cuda(device)
    to = lambda x, dt: (torch.from_numpy(np.ascontiguousarray(x)) if isinstance(x, np.ndarray) else x).to(dev, dt).contiguous()
    cf, cr, th = to(conf, torch.float64), to(correct, torch.uint8), to(thresholds, torch.float64)
    E1, N = cf.shape
    if th.dim() != 2 or th.shape[1] != E1 or tuple(cr.shape) != (E1, N):
        raise ValueError("conf (E1,N), correct (E1,N), thresholds (V,E1)")
    V = th.shape[0]
    acc = torch.empty((V,), dtype=torch.float64, device=dev)
    mex = torch.empty((V,), dtype=torch.float64, device=dev)
    hist = torch.empty((V, E1), dtype=torch.int32, device=dev) if want_hist else None
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    with torch.cuda.device(dev):
        capi.check(lib.ee_threshold_sweep(p(cf), p(cr), E1, N, p(th), V, p(acc), p(mex), p(hist), _stream()), None,
                   "ee_threshold_sweep")
    return acc, mex, hist
