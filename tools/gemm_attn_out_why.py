"""Why the attention-output GEMM (N = 768, K = 768, + residual) runs at ~315 algorithmic TFLOP/s where the QKV projection of the same K
runs at ~380 (GPU box only; timing diagnostics).  Varies one thing at a time: the size of the A / residual / output working set (Infinity
Cache-resident vs HBM), K at fixed N (per-tile fixed cost against k-loop length), N at fixed K (consumers per A panel), the epilogue."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_split_probe import run
MB = 512 * 462
print("# working set: M rows of A (3 KB), residual (3 KB), output (3 KB)")
for M in (16384, 65536, MB):
    run(M, 768, 768, epi=2, iters=8)
print("# no residual read (bias epilogue)")
run(MB, 768, 768, epi=0, iters=8)
run(16384, 768, 768, epi=0, iters=8)
print("# K at N = 768, M = 236544")
for K in (1536, 3072):
    run(MB, 768, K, epi=2, iters=6)
print("# N at K = 768, M = 236544, f32 output")
for N in (1536, 2304, 3072):
    run(MB, N, 768, epi=0, iters=6)
print("# N = 2304 with the residual epilogue")
run(MB, 2304, 768, epi=2, iters=6)
