"""Timing diagnostics of the LDS-DMA GEMM: what do the k-loop barrier and the DMA issue cost? (results are wrong with the
diagnostic flags; GPU box only)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_probe import run
M = 512 * 462
for flag, name in ((256, "DMA"), (256 | 32, "DMA, no in-loop DMA issue"), (256 | 512, "DMA, no k-loop barrier"), (256 | 1024, "DMA, no vmcnt wait"), (256 | 1024 | 512, "DMA, no vmcnt wait, no barrier")):
    print(name)
    run(M, 768, 3072, epi=0 | 64 | flag, wgs=0, iters=5)
    run(M, 2304, 768, epi=0 | 64 | flag, wgs=0, iters=5)
    run(M, 768, 3072, epi=0 | flag, wgs=0, iters=5)
