"""FFN-up shaped split GEMM: epilogue variants (GPU box only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_split_probe import run
M = 512 * 462
print("GELU + split rows out"); run(M, 3072, 768, epi=1, out_split=1, iters=5)
print("GELU + f32 out"); run(M, 3072, 768, epi=1, out_split=0, iters=5)
print("bias + split rows out (CfgB)"); run(M, 3072, 768, epi=0, out_split=1, iters=5)
print("bias + f32 out (CfgB)"); run(M, 3072, 768, epi=0, out_split=0, iters=5)
