"""A few launches of the split GEMM at the FFN shapes, for rocprofv3 --pmc passes (GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _diag  # noqa: F401  (diagnostic library)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_split_probe import run
M = 512 * 462
run(M, 3072, 768, epi=1, out_split=1, iters=3)
run(M, 768, 3072, epi=2, iters=3)
