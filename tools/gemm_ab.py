"""Same-process A/B of the split GEMM between builds of the library (GPU box only):

    python tools/gemm_ab.py [--rounds 3] [--iters 24] [--m 236544] lib_a.so lib_b.so ...      ("" / "tree" = the in-tree release library)

Every layer shape (QKV, attention output, FFN up, FFN down at LayoutLMv3-base widths) is timed through `ee_debug_gemm_split` for each
library in turn, `rounds` times over, so that clock drift of the box hits all builds alike (guide rule 24: interleaved rounds in ONE
process; the single 5-iteration runs of tools/gemm_split_shapes.py moved by 10 % with the order they were started in).  Prints the
median and best algorithmic TFLOP/s per (shape, library)."""
import argparse
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

SHAPES = (("qkv", 2304, 768, 0, 1), ("attn_out", 768, 768, 2, 0), ("ffn_up", 3072, 768, 1, 1), ("ffn_down", 768, 3072, 2, 0))


def load(path):
    if path in ("", "tree"):
        path = os.path.join(ROOT, "multi-modal-early-exit_amd", "libmmee_hip.so")
    elif not os.path.exists(path):
        path = os.path.join(ROOT, "tools", "bin", f"libmmee_hip_{path}.so")
    lib = C.CDLL(path)
    f = lib.ee_debug_gemm_split
    f.restype = C.c_int
    f.argtypes = [C.c_void_p] * 5 + [C.c_int32] * 5 + [C.c_float] * 3 + [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_float), C.c_void_p]
    return lib, os.path.basename(path)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=24)
    ap.add_argument("--m", type=int, default=512 * 462)
    ap.add_argument("--shapes", default="qkv,attn_out,ffn_up,ffn_down")
    ap.add_argument("--dbg", default="0", help="comma list of timing-variant bits (diagnostic libraries only; wrong results): 1 no in-loop DMA, "
                                               "8 no epilogue, 16 diagnostic build with nothing removed")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    libs = [load(p) for p in a.libs]
    M = a.m
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    for name, N, K, epi, osp in SHAPES:
        if name not in a.shapes.split(","):
            continue
        A = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) * 0.02
        b = torch.randn(N, device=dev)
        R = torch.randn(M, N, device=dev) if epi == 2 else None
        Cc = torch.zeros(M, N, device=dev)
        dbgs = [int(x) for x in a.dbg.split(",")]
        res = {(tag, d): [] for _, tag in libs for d in dbgs}
        clk = {}
        for r in range(a.rounds):
            for lib, tag in libs:
                for d in dbgs:
                    ms = (C.c_float * 8)(*([0.0] * 8))
                    rc = lib.ee_debug_gemm_split(p(A), p(W), p(b), p(R), p(Cc), M, N, K, epi | (d << 4), osp, 16.0, 256.0, 16.0, None, M, a.iters, ms, st)
                    torch.cuda.synchronize()
                    if rc != 0:
                        raise SystemExit(f"{tag}: ee_debug_gemm_split failed ({rc})")
                    res[(tag, d)].append(2.0 * M * N * K / ms[0] / 1e9)
                    clk[(tag, d)] = (ms[1], [round(ms[i]) for i in range(2, 7)])
        for _, tag in libs:
            for d in dbgs:
                v = res[(tag, d)]
                print(f"{name:9s} N={N} K={K}  {tag:28s} dbg {d:2d}  median {statistics.median(v):6.1f}  best {max(v):6.1f}  all {[round(x, 1) for x in v]}"
                      + (f"  clock {clk[(tag, d)][0]:.2f} GHz" if clk[(tag, d)][0] else "")
                      + (f"  cycles/tile loop, epilogue, wait, barrier, tiles/WG {clk[(tag, d)][1]}" if clk[(tag, d)][1][4] else ""), flush=True)
        del A, W, b, R, Cc


if __name__ == "__main__":
    main()
