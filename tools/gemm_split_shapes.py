"""The four layer GEMM shapes on the default split configuration, with and without their epilogue (GPU box only; timing diagnostics)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_split_probe import run
M = 512 * 462
DIAG, NOEPI = 256 << 4, (256 | 128) << 4
for name, N, K, epi, osp in (("qkv", 2304, 768, 0, 1), ("attn_out", 768, 768, 2, 0), ("ffn_up", 3072, 768, 1, 1), ("ffn_down", 768, 3072, 2, 0)):
    print(name, "path build"); run(M, N, K, epi=epi, out_split=osp, iters=6)
    print(name, "diag build"); run(M, N, K, epi=epi | DIAG, out_split=osp, iters=6)
    print(name, "no epilogue"); run(M, N, K, epi=epi | NOEPI, out_split=osp, iters=6)
