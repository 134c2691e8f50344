"""A/B of the register-staged GEMM against the LDS-DMA variant (epi | 256) through ee_debug_gemm (GPU box only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_probe import run
M = 512 * 462
for flag in (256, 0):
    print("LDS-DMA" if flag else "register-staged")
    run(4096 + 77, 768, 768, check=True, epi=2 | flag)
    run(300, 768, 768, check=True, epi=1 | flag)
    run(4096 + 77, 768, 768, check=True, epi=0 | flag, fold=37)
    run(M, 3072, 768, epi=1 | 64 | flag, wgs=0, iters=5)
    run(M, 768, 3072, epi=2 | 64 | flag, wgs=0, iters=5)
    run(M, 2304, 768, epi=0 | 64 | flag, wgs=0, iters=5)
