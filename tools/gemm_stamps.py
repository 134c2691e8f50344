"""In-kernel stamp shares of the GEMM main loop (diagnostic build; read SHARES, not run time)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _diag  # noqa: F401  (diagnostic library)
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
pkg = importlib.import_module("multi-modal-early-exit_amd")
lib = pkg.capi.load(); dev = torch.device("cuda:0")
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
def run(M, N, K, epi, wgs):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    Cc = torch.empty(M, N, device=dev); clk = torch.zeros(8 * 1024, dtype=torch.int64, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        pkg.capi.check(lib.ee_debug_gemm(p(A), p(W), p(b), None, p(Cc), M, N, K, epi, -wgs, None, p(clk), st), None, "gemm")
    torch.cuda.synchronize()
    c = clk.cpu().numpy().reshape(-1, 8)[: wgs * 256].astype(np.float64)
    tot = c[:, 0].mean(); names = ["issue", "compute", "wait+ds_write", "barrier", "prologue", "epilogue"]
    tiles = ((M + 127) // 128) * (N // 128) / (wgs * 256); stages = tiles * K / 32
    print(f"M={M} N={N} K={K} epi={epi} wgs/cu={wgs}: total {tot:.3g} cyc/WG, {tiles:.1f} tiles, {stages:.0f} stages per WG")
    for i, n in enumerate(names):
        v = c[:, 2 + i].mean()
        per = v / (tiles if i >= 4 else stages)
        print(f"   {n:14s} {v / tot:6.1%}   {per:9.0f} cycles per {'tile' if i >= 4 else 'stage'}")
M = 512 * 462
for wgs in (1, 2):
    run(M, 768, 768, 0, wgs)
    run(M, 3072, 768, 1, wgs)
