"""Micro-benchmark of the path's GEMM kernel through ee_debug_gemm (GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _diag  # noqa: F401  (diagnostic library)
import ctypes as C
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np

pkg = importlib.import_module("multi-modal-early-exit_amd")
lib = pkg.capi.load()
dev = torch.device("cuda:0")
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None


def run(M, N, K, epi=0, wgs=2, iters=10, check=False, fold=0):
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    R = torch.randn(M, N, device=dev) if (epi & 15) == 2 else None
    Cc = torch.empty(M, N, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    clk = torch.zeros(2 * 4096, dtype=torch.int64, device=dev)
    rs = (torch.arange(M, device=dev, dtype=torch.int32) % fold).contiguous() if fold else None
    f = lambda: pkg.capi.check(lib.ee_debug_gemm(p(A), p(W), p(b), p(R), p(Cc), M, N, K, epi, wgs, p(rs), p(clk), st), None, "gemm")
    f()
    torch.cuda.synchronize()
    if check:
        ref = (A[rs.long()] if fold else A) @ W.t() + b
        if (epi & 15) == 1: ref = torch.nn.functional.gelu(ref)
        if (epi & 15) == 2: ref = ref + R
        if (epi & 15) == 3: ref = torch.tanh(ref)
        print("  max err vs torch", (Cc - ref).abs().max().item())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    tf = 2.0 * M * N * K / ms / 1e9
    c = clk.cpu().numpy().reshape(-1, 2)
    c = c[c[:, 1] > 0]
    ghz = float((c[:, 0] / c[:, 1]).mean()) * 0.1 if len(c) else 0.0
    if len(c) >= 512:
        t = c[:, 1] / 100.0      # microseconds each workgroup was alive
        print(f"   per-WG alive time (us): first half mean {t[:len(t)//2].mean():.0f}  second half mean {t[len(t)//2:].mean():.0f}  "
              f"min {t.min():.0f} max {t.max():.0f}; by (wg>>3)&63 parity: {t[(np.arange(len(t))>>3)%2==0].mean():.0f} / {t[(np.arange(len(t))>>3)%2==1].mean():.0f}")
    print(f"M={M} N={N} K={K} epi={epi} fold={fold} wgs/cu={wgs}: {ms:.3f} ms  {tf:.1f} TFLOP/s ({tf / 157.3:.1%})  clock {ghz:.2f} GHz -> "
          f"{(tf / (157.3 * ghz / 2.4)) if ghz else 0.0:.1%} of the peak at that clock", flush=True)
    return tf


MB = 512 * 462


def cmd_variants():
    """register-staged vs LDS-DMA kernel (epi | 256), correctness + rate at the layer shapes"""
    for flag in (256, 0):
        print("LDS-DMA" if flag else "register-staged")
        run(4096 + 77, 768, 768, check=True, epi=2 | flag)
        run(300, 768, 768, check=True, epi=1 | flag)
        run(4096 + 77, 768, 768, check=True, epi=0 | flag, fold=37)
        for N, K, e in ((3072, 768, 1), (768, 3072, 2), (2304, 768, 0)):
            run(MB, N, K, epi=e | 64 | flag, wgs=0, iters=5)


def cmd_diag():
    """timing variants of the LDS-DMA kernel (WRONG results): what the k-loop barrier, the DMA issue and the wait cost"""
    for flag, name in ((256, "DMA"), (256 | 32, "DMA, no in-loop DMA issue"), (256 | 512, "DMA, no k-loop barrier"), (256 | 1024, "DMA, no vmcnt wait"),
                       (256 | 1024 | 512, "DMA, no vmcnt wait, no barrier")):
        print(name)
        run(MB, 768, 3072, epi=0 | 64 | flag, wgs=0, iters=5)
        run(MB, 2304, 768, epi=0 | 64 | flag, wgs=0, iters=5)
        run(MB, 768, 3072, epi=0 | flag, wgs=0, iters=5)


def cmd_stagger():
    """does starting the second workgroup of each CU out of phase help? (epi bits 12..: stagger)"""
    for st in (0, 6, 12, 18):
        print("stagger", st)
        for N, K, e in ((3072, 768, 1), (768, 3072, 2), (2304, 768, 0)):
            run(MB, N, K, epi=e | 256 | (st << 12), wgs=0, iters=5)


def cmd_queue():
    """static grid stride (epi | 16) vs XCD-local work queues"""
    run(4096 + 77, 768, 768, check=True, epi=2)
    run(300, 768, 768, check=True, epi=1)
    for mode, name in ((16, "static grid stride"), (0, "XCD-local queues")):
        print(name)
        for N, K, e in ((3072, 768, 1), (768, 3072, 2), (2304, 768, 0)):
            run(MB, N, K, epi=e | mode | 64, wgs=0, iters=5)




if __name__ == "__main__":
    # python tools/gemm_probe.py [variants|diag|stagger|queue]  (rounds 1-2 had one ten-line script per experiment)
    if len(sys.argv) > 1:
        {"variants": cmd_variants, "diag": cmd_diag, "stagger": cmd_stagger, "queue": cmd_queue}[sys.argv[1]]()
    else:
        run(4096, 768, 768, check=True)
        for wgs in (1, 2):
            run(MB, 3072, 768, epi=1, wgs=wgs)
            run(MB, 768, 3072, epi=2, wgs=wgs)
            run(MB, 2304, 768, epi=0, wgs=wgs)
