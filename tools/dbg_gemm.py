import sys, os, ctypes as C, importlib
sys.path.insert(0, os.getcwd())
import torch, numpy as np
pkg = importlib.import_module("multi-modal-early-exit_amd")
lib = pkg.capi.load()
dev = torch.device("cuda:0")
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
for (M, N, K, epi, osp) in ((1, 256, 32, 0, 0), (130, 256, 96, 0, 0), (40, 256, 64, 0, 1)):
    gen = torch.Generator(device="cpu").manual_seed(1)
    A = torch.randn(M, K, generator=gen).to(dev); W = (torch.randn(N, K, generator=gen) * 0.02).to(dev); b = torch.randn(N, generator=gen).to(dev)
    out = torch.full((M, N), float("nan"), device=dev)
    pkg.capi.check(lib.ee_debug_gemm_split(p(A), p(W), p(b), None, p(out), M, N, K, epi, osp, 16.0, 256.0, 16.0, None, M, 1, None,
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream)), None, "x")
    torch.cuda.synchronize()
    ref = (A.double() @ W.double().t() + b.double())
    if osp:
        h = out.view(torch.float16).view(M, N // 16, 2, 16); got = (h[:, :, 0].double() + h[:, :, 1].double()).reshape(M, N) / 16
    else:
        got = out.double()
    err = (got - ref).abs().cpu().numpy()
    bad = np.argwhere(~(err < 1e-4))
    print(M, N, K, epi, osp, "max err", np.nanmax(err), "bad count", len(bad), "nan", int(np.isnan(err).sum()))
    if len(bad):
        rows = sorted(set(bad[:, 0].tolist())); cols = sorted(set(bad[:, 1].tolist()))
        print("  bad rows", rows[:40]); print("  bad cols", cols[:80])
        r, c = bad[0]
        # where does the wrong value come from?
        v = got[r, c].item(); d = (ref - v).abs(); rr, cc = np.unravel_index(d.cpu().numpy().argmin(), d.shape)
        print("  first bad", (r, c), "value", v, "closest ref at", (rr, cc), "diff", d[rr, cc].item())
