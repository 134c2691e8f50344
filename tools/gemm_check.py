"""Quick correctness screen of the split GEMM through ee_debug_gemm_split against float64 (GPU box only): ragged M, every epilogue, split
output, several launches back to back (the tile hand-over paths), prints max |error| and where the bad elements sit."""
import sys, os, ctypes as C, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
pkg = importlib.import_module("multi-modal-early-exit_amd")
lib = pkg.capi.load()
dev = torch.device("cuda:0")
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
bad_total = 0
for (M, N, K, epi, osp) in ((1, 256, 32, 0, 0), (130, 256, 96, 0, 0), (40, 256, 64, 0, 1), (3000, 768, 768, 2, 0), (5000, 2304, 768, 0, 1), (70000, 768, 768, 2, 0),
                            (70000, 3072, 768, 1, 1), (66000, 2304, 768, 0, 1)):
    gen = torch.Generator(device="cpu").manual_seed(1)
    A = torch.randn(M, K, generator=gen).to(dev); W = (torch.randn(N, K, generator=gen) * 0.02).to(dev); b = torch.randn(N, generator=gen).to(dev)
    R = torch.randn(M, N, generator=gen).to(dev) if epi == 2 else None
    out = torch.full((M, N), float("nan"), device=dev)
    pkg.capi.check(lib.ee_debug_gemm_split(p(A), p(W), p(b), p(R), p(out), M, N, K, epi, osp, 16.0, 256.0, 16.0, None, M, 3, None,
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream)), None, "x")
    torch.cuda.synchronize()
    ref = A.double() @ W.double().t() + b.double()
    if epi == 1: ref = torch.nn.functional.gelu(ref)
    if epi == 2: ref = ref + R.double()
    if osp:
        h = out.view(torch.float16).view(M, N // 16, 2, 16); got = (h[:, :, 0].double() + h[:, :, 1].double()).reshape(M, N) / 16
    else:
        got = out.double()
    err = (got - ref).abs()
    nbad = int((~(err < 1e-4)).sum().item())
    bad_total += nbad
    print(M, N, K, "epi", epi, "split", osp, "max err", float(torch.nan_to_num(err, nan=1e9).max()), "bad", nbad, flush=True)
    if nbad:
        idx = torch.nonzero(~(err < 1e-4))[:, 0]
        print("   bad rows (first 20):", sorted(set(idx.cpu().tolist()))[:20])
sys.exit(1 if bad_total else 0)
