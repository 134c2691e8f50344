import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_probe import run
M = 512 * 462
for mode, name in ((0, "no priority"), (64, "prio: odd wave slot"), (128, "prio: second half of grid")):
    print(name)
    run(M, 3072, 768, epi=1 | mode, wgs=0, iters=5)
    run(M, 768, 3072, epi=0 | mode, wgs=0, iters=5)
    run(M, 2304, 768, epi=0 | mode, wgs=0, iters=5)
