import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_probe import run
M = 512 * 462
run(M, 768, 768, epi=0, wgs=2, iters=3)
run(M, 3072, 768, epi=1, wgs=2, iters=3)
run(M, 768, 3072, epi=0, wgs=2, iters=3)
