import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_probe import run
M = 512 * 462
run(4096 + 77, 768, 768, check=True, epi=2)
run(300, 768, 768, check=True, epi=1)
for mode, name in ((16, "static grid stride"), (0, "XCD-local queues")):
    print(name)
    run(M, 3072, 768, epi=1 | mode | 64, wgs=0, iters=5)
    run(M, 768, 3072, epi=2 | mode | 64, wgs=0, iters=5)
    run(M, 2304, 768, epi=0 | mode | 64, wgs=0, iters=5)
