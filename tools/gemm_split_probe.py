"""Correctness + rate of the split-precision GEMM kernel through ee_debug_gemm_split (GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _diag  # noqa: F401  (diagnostic library)
import ctypes as C
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

pkg = importlib.import_module("multi-modal-early-exit_amd")
lib = pkg.capi.load()
dev = torch.device("cuda:0")
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None


def decode_split(buf, N, scale):
    """[M, N] f32-sized rows holding (hi N f16 | lo N f16) -> f64 values"""
    h = buf.view(torch.float16).view(buf.shape[0], N // 16, 2, 16)      # 64-byte groups [hi 16 | lo 16]
    return (h[:, :, 0].double() + h[:, :, 1].double()).reshape(buf.shape[0], N) / scale


def run(M, N, K, epi=0, out_split=0, iters=1, check=False, fold=0, a_std=1.0):
    A = torch.randn(M, K, device=dev) * a_std
    W = torch.randn(N, K, device=dev) * 0.02
    b = torch.randn(N, device=dev)
    R = torch.randn(M, N, device=dev) if (epi & 15) == 2 else None
    Cc = torch.zeros(M, N, device=dev)
    rs = ((torch.arange(M, device=dev, dtype=torch.int32) * 3) // 4).contiguous() if fold else None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ms = (C.c_float * 2)(0, 0)
    pkg.capi.check(lib.ee_debug_gemm_split(p(A), p(W), p(b), p(R), p(Cc), M, N, K, epi, out_split, 16.0, 256.0, 16.0, p(rs), M, iters,
                                           ms, st), None, "gemm_split")
    torch.cuda.synchronize()
    if check:
        Ad = (A[rs.long()] if fold else A).double()
        ref = Ad @ W.double().t() + b.double()
        if epi == 1: ref = torch.nn.functional.gelu(ref)
        if epi == 2: ref = ref + (R[rs.long()] if fold else R).double()
        if epi == 3: ref = torch.tanh(ref)
        got = decode_split(Cc, N, 16.0) if out_split else Cc.double()
        f32 = ((A[rs.long()] if fold else A) @ W.t() + b)
        if epi == 1: f32 = torch.nn.functional.gelu(f32)
        if epi == 2: f32 = f32 + (R[rs.long()] if fold else R)
        if epi == 3: f32 = torch.tanh(f32)
        print(f"  M={M} N={N} K={K} epi={epi} out_split={out_split} fold={fold}: max |err| vs f64 {float((got - ref).abs().max()):.3e}"
              f"   (torch f32 GEMM: {float((f32.double() - ref).abs().max()):.3e})", flush=True)
    if iters > 1:
        tf = 2.0 * M * N * K / ms[0] / 1e9
        clk = f", clock {ms[1]:.2f} GHz -> {tf / (2500 / 3 * ms[1] / 2.4):.1%} of the peak at that clock" if ms[1] else ""
        print(f"M={M} N={N} K={K} epi={epi} out_split={out_split}: {ms[0]:.3f} ms  {tf:.1f} TFLOP/s algorithmic "
              f"({tf / 157.3:.2f} x the f32 MFMA peak, {tf / (2500 / 3):.1%} of f16 peak / 3{clk})", flush=True)


if __name__ == "__main__":
    run(300, 256, 64, check=True)
    run(4096 + 77, 768, 768, epi=2, check=True)
    run(1000, 3072, 768, epi=1, out_split=1, check=True)
    run(1000, 768, 3072, epi=2, check=True, fold=1)
    run(515, 2304, 768, epi=0, check=True, a_std=0.05)
    M = 512 * 462
    run(M, 3072, 768, epi=1, out_split=1, iters=6)
    run(M, 768, 3072, epi=2, iters=6)
    run(M, 2304, 768, epi=0, iters=6)
    run(M, 768, 768, epi=2, iters=6)
