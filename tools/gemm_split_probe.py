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


MB = 512 * 462


def cmd_diag():
    """where the split GEMM's time goes: timing-only variants (WRONG results with a diagnostic bit set).  For numbers that can be compared
    between builds use tools/gemm_ab.py (interleaved rounds in one process: a single short run moves by 10 % with the order it is started in)."""
    for bits, name in ((256, "full (diagnostic build)"), (16, "no in-loop DMA"), (32, "no barrier"), (64, "no DMA wait"), (32 | 64, "no DMA wait, no barrier"),
                       (128, "no epilogue"), (16 | 128, "no DMA, no epilogue"), (16 | 32 | 128, "MFMA + LDS reads only")):
        print(name)
        run(MB, 3072, 768, epi=1 | (bits << 4), out_split=1, iters=4)
        run(MB, 768, 3072, epi=2 | (bits << 4), iters=4)


def cmd_epi():
    """FFN-up shape: epilogue variants"""
    for name, e, o in (("GELU + split rows out", 1, 1), ("GELU + f32 out", 1, 0), ("bias + split rows out", 0, 1), ("bias + f32 out", 0, 0)):
        print(name)
        run(MB, 3072, 768, epi=e, out_split=o, iters=5)


def cmd_pmc():
    """a few launches at the FFN shapes, for rocprofv3 --pmc passes"""
    run(MB, 3072, 768, epi=1, out_split=1, iters=3)
    run(MB, 768, 3072, epi=2, iters=3)


def cmd_tail():
    """FFN-down shape at tile counts around a multiple of the 256 CUs: what the last, partly filled round costs"""
    for tiles_m in (1024, 939, 920, 854, 768, 512, 342, 320, 256):      # x 3 N-tiles: 12, 11.004, 10.78, 10.008, 9, 6, 4.008, 3.75, 3 rounds
        print(f"tiles_m={tiles_m}: {tiles_m * 3 / 256:.3f} rounds of 256 workgroups")
        run(tiles_m * 256, 768, 3072, epi=2, iters=6)


def cmd_why_attn_out():
    """why the attention-output GEMM (N = 768, K = 768, + residual) is the slowest shape: working set, epilogue, K, N one at a time"""
    print("# working set: M rows of A (3 KB), residual (3 KB), output (3 KB)")
    for M in (16384, 65536, MB):
        run(M, 768, 768, epi=2, iters=8)
    print("# no residual read (bias epilogue)")
    run(MB, 768, 768, epi=0, iters=8)
    run(16384, 768, 768, epi=0, iters=8)
    print("# K at N = 768")
    for K in (1536, 3072):
        run(MB, 768, K, epi=2, iters=6)
    print("# N at K = 768, f32 output")
    for N in (1536, 2304, 3072):
        run(MB, N, 768, epi=0, iters=6)
    print("# N = 2304 with the residual epilogue")
    run(MB, 2304, 768, epi=2, iters=6)


if __name__ == "__main__":
    if len(sys.argv) > 1:      # python tools/gemm_split_probe.py diag|epi|pmc|tail|why_attn_out
        {"diag": cmd_diag, "epi": cmd_epi, "pmc": cmd_pmc, "tail": cmd_tail, "why_attn_out": cmd_why_attn_out}[sys.argv[1]]()
        raise SystemExit(0)
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
