"""Where does the split GEMM's time go?  Timing-only variants (results are wrong with a diagnostic bit set)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_split_probe import run
M = 512 * 462
for bits, name in ((256, "full (diagnostic build)"), (16, "no in-loop DMA"), (32, "no barrier"), (64, "no DMA wait"), (32 | 64, "no DMA wait, no barrier"), (128, "no epilogue"), (16 | 128, "no DMA, no epilogue"),
                   (16 | 32 | 128, "MFMA + LDS reads only")):
    print(name)
    run(M, 3072, 768, epi=1 | bits, out_split=1, iters=4)
    run(M, 768, 3072, epi=2 | bits, iters=4)
