"""Two half-batches side by side: plain streams against CU-partitioned streams (hipExtStreamCreateWithCUMask, called through ctypes; word i of
the mask = XCD i on MI355X, the same bits in every word keep a partition balanced over the XCDs).  GPU box only.
    python tools/two_stream_probe.py [bits_for_partition_A ...]      e.g. 16 24 (CUs of every XCD given to partition A; B gets the rest)
Round 3, one box: one stream B=512 6307 docs/s, two plain streams 6317, two partitions 128 / 128 CUs 6047, 192 / 64 CUs 4015, 64 / 192 3748:
confining two forwards to halves of the chip LOSES 4 %; the split GEMM is power-limited chip-wide, not per partition."""
import ctypes as C, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

pkg = importlib.import_module("multi-modal-early-exit_amd")
lib = pkg.capi.load()
ee = dict(exits=[2, 4, 6, 8, 10], encoder_layer_strategy="ramp")
cfg = pkg.ModelConfig.base(EE_config=ee)
W = pkg.synth.make_weights(cfg, seed=1234, head_gain=6.0)
B = int(os.environ.get("B", "512"))
docs = pkg.synth.make_documents(cfg, B, seed=1234, text_len=512)
dev = torch.device("cuda:0")
T = {k: torch.as_tensor(v).to(dev) for k, v in docs.items()}
thr = [0.520424, 0.550076, 0.50362, 0.428325, 0.883209, 2.0]


def mk(n):
    e = pkg.EarlyExitEngine(cfg, max_docs=n, max_text_len=512)
    e.load_weights(W)
    return e


hip = C.CDLL("libamdhip64.so")


def masked_stream(bits_lo, bits_n):
    words = (C.c_uint32 * 8)(*[(((1 << bits_n) - 1) << bits_lo) & 0xffffffff] * 8)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), C.c_uint32(8), words)
    assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
    return torch.cuda.ExternalStream(s.value, device=dev)


def run(engines, streams, parts, steps=6):
    def step():
        for e, s, (lo, hi) in zip(engines, streams, parts):
            with torch.cuda.stream(s):
                e.forward(T["input_ids"][lo:hi], T["attention_mask"][lo:hi], T["bbox"][lo:hi], T["pixel_values"][lo:hi], thresholds=thr)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    for e in engines:
        e.pin_schedule()
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return B * steps / (time.perf_counter() - t0)


if B <= 1536:          # a split-precision handle addresses at most 4 GiB of rows (~1900 base-size documents)
    one = mk(B)
    print(f"one stream, B={B}:", round(run([one], [torch.cuda.Stream()], [(0, B)]), 1), "docs/s", flush=True)
    one.close()
h = B // 2
a, b = mk(h), mk(h)
print(f"two plain streams, 2 x {h}:", round(run([a, b], [torch.cuda.Stream(), torch.cuda.Stream()], [(0, h), (h, B)]), 1), "docs/s", flush=True)
for na in [int(x) for x in sys.argv[1:]] or [16]:
    sa, sb = masked_stream(0, na), masked_stream(na, 32 - na)
    print(f"two CU-masked streams ({8 * na} / {8 * (32 - na)} CUs):",
          round(run([a, b], [sa, sb], [(0, h), (h, B)]), 1), "docs/s", flush=True)
a.close(); b.close()
