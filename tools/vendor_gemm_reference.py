"""Context measurement, NOT a product path (the path uses no vendor BLAS): what torch.matmul (hipBLASLt / rocBLAS) sustains on this box for PLAIN f16 / bf16 GEMMs of the four
layer shapes, against which the split kernel's raw matrix-pipe rate (3 f16 MFMA terms per algorithmic MAC + a fused epilogue) can be read.  GPU box only.

    python tools/vendor_gemm_reference.py [rows]"""
import sys, time
import torch

M = int(sys.argv[1]) if len(sys.argv) > 1 else 476160          # rows of one micro-batch's first stage at the bench mix
dev = torch.device("cuda:0")
for name, N, K in (("qkv", 2304, 768), ("attn_out", 768, 768), ("ffn_up", 3072, 768), ("ffn_down", 768, 3072)):
    for dt in (torch.float16, torch.bfloat16):
        A = torch.randn(M, K, device=dev, dtype=dt)
        W = torch.randn(N, K, device=dev, dtype=dt) * 0.02
        for _ in range(3):
            C = A @ W.t()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            C = A @ W.t()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
        print(f"{name:9s} M={M} N={N} K={K} {str(dt):15s}: {ms:7.3f} ms  {tf:7.1f} TFLOP/s  (output {M * N * 2 / 1e9:.2f} GB at {M * N * 2 / (ms * 1e-3) / 1e12:.2f} TB/s)")
        del A, W, C
