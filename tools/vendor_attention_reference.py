"""Context measurement, NOT a product path: what torch's scaled_dot_product_attention (the vendor flash-attention kernels of this PyTorch-ROCm build) sustains for PLAIN f16
attention of the path's shape (12 heads x 64, 460-row documents), without a bias and with a materialised additive bias tensor (what the reference feeds: rel_pos + rel_2d_pos,
EE/models/LayoutLMv3.py:170-179), against which the path's kernel (three f16 MFMA terms per MAC, bias from the pair index, f32 softmax) can be read.  GPU box only."""
import sys
import torch
import torch.nn.functional as F

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S, NH, D = 460, 12, 64
dev = torch.device("cuda:0")
q = torch.randn(B, NH, S, D, device=dev, dtype=torch.float16)
k = torch.randn(B, NH, S, D, device=dev, dtype=torch.float16)
v = torch.randn(B, NH, S, D, device=dev, dtype=torch.float16)
bias = torch.randn(B, NH, S, S, device=dev, dtype=torch.float16)
flops = 4.0 * B * NH * S * S * D


def run(name, fn):
    try:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{name}: {ms:8.3f} ms  {flops / (ms * 1e-3) / 1e12:7.1f} TFLOP/s (B={B}, {NH} heads, S={S}, d={D}, f16)")
    except Exception as ex:  # noqa: BLE001
        print(f"{name}: not available ({type(ex).__name__}: {str(ex)[:120]})")


run("sdpa, no bias              ", lambda: F.scaled_dot_product_attention(q, k, v))
run("sdpa, additive bias tensor ", lambda: F.scaled_dot_product_attention(q, k, v, attn_mask=bias))
run("unfused matmul-softmax-matmul with bias", lambda: torch.softmax((q @ k.transpose(-1, -2)) * 0.125 + bias, dim=-1) @ v)
