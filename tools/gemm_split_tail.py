"""FFN-down shaped split GEMM at tile counts around a multiple of the 256 CUs: what the last, partly filled round costs (GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _diag  # noqa: F401  (diagnostic library)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_split_probe import run
for tiles_m in (1024, 939, 920, 854, 768, 512, 342, 320, 256):      # x 3 N-tiles: 12, 11.004, 10.78, 10.008, 9, 6, 4.008, 3.75, 3 rounds
    print(f"tiles_m={tiles_m}: {tiles_m * 3 / 256:.3f} rounds of 256 workgroups")
    run(tiles_m * 256, 768, 3072, epi=2, iters=6)
