"""Does starting the second workgroup of each CU out of phase help? (LDS-DMA GEMM; GPU box only)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_probe import run
M = 512 * 462
for st in (0, 6, 12, 18):
    print("stagger", st)
    run(M, 3072, 768, epi=1 | 256 | (st << 12), wgs=0, iters=5)
    run(M, 768, 3072, epi=2 | 256 | (st << 12), wgs=0, iters=5)
    run(M, 2304, 768, epi=0 | 256 | (st << 12), wgs=0, iters=5)
