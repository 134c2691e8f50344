"""GPU: (1) BASELINE configs[2] at FULL size — LayoutLMv3-large, L = 24, I = 4096, gate exit after every layer (23 exits),
per-exit temperatures, T = 512 — against the torch-CPU restatement and through size-independent properties;
(2) the split-precision arithmetic off the happy path: checkpoint-like weight statistics (LayerNorm gains with outlier
channels, heavy-tailed weights, large position embeddings) must either meet the tolerance or fail loudly, never clamp silently.
Reference path: EE/models/LayoutLMv3.py:181-248 (layer loop + exits), :764-792 (gate path), EE/generic_scaling.py:54-61
(temperature apply), EE/policy.py:12-53 (policy)."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
LOGIT_TOL = 1e-4


def _np(t):
    return t.detach().cpu().numpy()


def _gap_thresholds(conf, release=0.12):
    """Per-exit thresholds in gaps between neighbouring confidences: every exit releases ~`release` of its arrivals."""
    E1, n = conf.shape
    thr = np.full(E1, 2.0)
    active = np.ones(n, dtype=bool)
    margin = 1.0
    for e in range(E1 - 1):
        c = np.sort(conf[e, active])
        if len(c) < 2:
            break
        k = min(max(int(round((1.0 - release) * len(c))), 1), len(c) - 1)
        lo, hi = max(1, k - 3), min(len(c) - 1, k + 3)
        j = lo + int(np.argmax(c[lo:hi + 1] - c[lo - 1:hi]))
        thr[e] = 0.5 * (c[j - 1] + c[j])
        margin = min(margin, float(np.abs(conf[e, active] - thr[e]).min()))     # only documents that reach the exit are tested
        active &= ~(conf[e] > thr[e])
    return thr, margin


def _config3(pkg):
    ee = dict(exits=list(range(1, 24)), encoder_layer_strategy="gate", inference_strategy="max_confidence")
    cfg = pkg.ModelConfig.large(EE_config=ee)
    assert (cfg.hidden_size, cfg.num_hidden_layers, cfg.intermediate_size, cfg.num_attention_heads) == (1024, 24, 4096, 16)
    W = pkg.synth.make_weights(cfg, seed=1234, head_gain=6.0)
    temps = np.random.default_rng(1234).uniform(0.5, 3.0, 24)
    return ee, cfg, W, temps


def test_config3_full_size_against_cpu_restatement(pkg, oracle):
    """4 documents, every one of the 23 gate exits + the final classifier, against oracle.ee_oracle_torch (B = 1 per forward,
    full depth, as the reference evaluates): policy logits <= 1e-4 abs at all 24 exits, 2-way gate logits <= 1e-4, and the
    early-exit run (temperatures applied before the test) leaves at bit-identical exit indices."""
    import torch
    otorch = importlib.import_module("oracle.ee_oracle_torch")
    ee, cfg, W, temps = _config3(pkg)
    N = 4
    docs = pkg.synth.make_documents(cfg, N, seed=77, text_len=512)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    tor = otorch.TorchOracle(cfg, W)
    refs = [tor.forward_all({k: v[i:i + 1] for k, v in docs.items()}, ee["exits"], strategy="gate") for i in range(N)]
    store = np.concatenate([r["logits_store"] for r in refs], axis=1)                  # (24, N, 16)
    eng = pkg.EarlyExitEngine(cfg, max_docs=N, max_text_len=512, xprobe=False)
    assert eng.precision == "split"
    eng.load_weights(W)
    args = (docs["input_ids"], docs["attention_mask"], docs["bbox"], docs["pixel_values"])
    full = eng.forward(*args, dump_all=True, want_all=True, want_head=True, validate=True)
    np.testing.assert_allclose(_np(full.all_logits), store, rtol=0, atol=LOGIT_TOL)
    if "exit_logits" in refs[0]:
        gate = np.concatenate([r["exit_logits"] for r in refs], axis=1)
        np.testing.assert_allclose(_np(full.head_logits), gate, rtol=0, atol=LOGIT_TOL)
    scaled = oracle.temperature_scale(store, temps)
    conf = oracle.softmax64(scaled).max(-1)
    # a spread of exit depths over 4 documents: thresholds placed so that documents leave at different exits, each in a gap
    thr = np.full(24, 2.0)
    order = np.argsort(-conf[5])                                                     # most confident document leaves at exit 5, ...
    for rank, e in enumerate((5, 11, 17)):
        d = order[rank]
        others = np.delete(conf[e], [order[r] for r in range(rank + 1)])
        lo = others.max() if len(others) and others.max() < conf[e, d] else conf[e, d] - 0.02
        thr[e] = 0.5 * (conf[e, d] + lo)
    ex, pred, _ = oracle.policy_scan(scaled, thr)
    margin = np.abs(conf[:-1] - thr[:-1, None]).min()
    assert margin > 1e-5, margin
    out = eng.forward(*args, thresholds=thr, temperatures=temps, validate=True)
    assert np.array_equal(_np(out.exit_layer), ex), (_np(out.exit_layer), ex)
    # the policy's `predictions` are rows of the temperature-scaled store (EE/eval.py:321-323 scales, EE/policy.py:36 picks)
    np.testing.assert_allclose(_np(out.logits), pred, rtol=0, atol=LOGIT_TOL)
    # the X-space probe on the 16-head x 1024 instantiation (csrc/xprobe.hip): same exits, same tolerance against the CPU restatement
    outx = eng.forward(*args, thresholds=thr, temperatures=temps, xprobe=True, probe_always=True, validate=True)
    assert np.array_equal(_np(outx.exit_layer), ex)
    np.testing.assert_allclose(_np(outx.logits), pred, rtol=0, atol=LOGIT_TOL)
    plan = eng.layer_plan()
    assert sum(plan["docs_probe"]) > 0 and plan["rows_qkv"][-1] == 0        # probed, and the last layer projected nothing
    eng.close()


def test_config3_full_size_properties(pkg, oracle):
    """64 ragged documents through the full-size config 3 (23 compaction stages): early-exit rows are BIT-identical to the
    dump-all rows at the chosen exit, permuting the batch permutes the outputs bit for bit, stage populations are the
    survivors, every document leaves exactly once."""
    ee, cfg, W, temps = _config3(pkg)
    B = 64
    docs = pkg.synth.make_documents(cfg, B, seed=78, text_len=512)
    eng = pkg.EarlyExitEngine(cfg, max_docs=B, max_text_len=512, xprobe=False)
    eng.load_weights(W)
    args = (docs["input_ids"], docs["attention_mask"], docs["bbox"], docs["pixel_values"])
    raw = eng.forward(*args, dump_all=True, want_all=True, validate=True)
    full = eng.forward(*args, dump_all=True, want_all=True, temperatures=temps, validate=True)
    store = _np(full.all_logits).astype(np.float64)                                     # logits / T, as the policy sees them
    assert np.isfinite(store).all()
    np.testing.assert_allclose(store, oracle.temperature_scale(_np(raw.all_logits).astype(np.float64), temps), rtol=1e-6, atol=1e-7)
    conf = oracle.softmax64(store).max(-1)
    np.testing.assert_allclose(_np(full.all_crit), conf, rtol=0, atol=2e-6)             # criterion = max softmax of logits / T
    thr, margin = _gap_thresholds(conf, release=0.12)
    assert margin > 1e-6, margin
    ex_ref, _, _ = oracle.policy_scan(store, thr)
    out = eng.forward(*args, thresholds=thr, temperatures=temps, validate=True)
    ex = _np(out.exit_layer)
    assert np.array_equal(ex, ex_ref)
    assert len(np.unique(ex)) >= 10                                                     # many of the 24 stages really shrink
    got = _np(out.logits)
    assert np.array_equal(got, _np(full.all_logits)[ex, np.arange(B)])
    counts = eng.stage_counts()["docs"]
    assert counts == [int((ex >= e).sum()) for e in range(len(counts))]
    perm = np.random.default_rng(3).permutation(B)
    outp = eng.forward(*(a[perm] for a in args), thresholds=thr, temperatures=temps)
    assert np.array_equal(_np(outp.exit_layer), ex[perm]) and np.array_equal(_np(outp.logits), got[perm])
    # X-space probe at every one of the 23 exit layers: a re-association, so exits equal (thresholds sit in gaps) and logits within tolerance
    outx = eng.forward(*args, thresholds=thr, temperatures=temps, xprobe=True, probe_always=True, validate=True)
    exx = _np(outx.exit_layer)
    near = np.zeros(B, dtype=bool)
    for e in range(len(thr) - 1):
        near |= np.abs(conf[e] - thr[e]) < 1e-5
    assert (exx == ex)[~near].all()
    same = exx == ex
    np.testing.assert_allclose(_np(outx.logits)[same], got[same], rtol=0, atol=LOGIT_TOL)
    planx = eng.layer_plan()
    assert sum(planx["docs_probe"]) > 0 and planx["rows_qkv"][-1] == 0 and all(q <= m for q, m in zip(planx["rows_qkv"][1:], planx["rows_main"][:-1]))
    eng.close()


# ---- split precision off the happy path ---------------------------------------------------------------------------------
def _checkpoint_like(pkg, cfg, seed, gamma_outliers=(30.0, 60.0, 100.0), weight_tail=0.7, pos_gain=20.0):
    """Random-init weights re-shaped towards what trained checkpoints look like: log-normal LayerNorm gains with a few very
    large channels, log-normal (heavy-tailed) weight magnitudes with sparse x20 entries, large position embeddings."""
    rng = np.random.default_rng(seed)
    W = pkg.synth.make_weights(cfg, seed=seed, head_gain=3.0)
    for k in list(W):
        v = W[k]
        if k.endswith("LayerNorm.weight") or k.endswith("norm.weight"):
            g = np.exp(rng.normal(0.0, 0.5, v.shape)).astype(np.float32) * np.sign(rng.normal(size=v.shape)).astype(np.float32)
            ch = rng.choice(v.shape[0], size=len(gamma_outliers), replace=False)
            g[ch] = np.asarray(gamma_outliers, np.float32) * np.sign(rng.normal(size=len(ch))).astype(np.float32)
            W[k] = g
        elif v.ndim == 2 and ("dense" in k or "query" in k or "key" in k or "value" in k) and "early_exits" not in k and "classifier" not in k:
            m = np.exp(rng.normal(0.0, weight_tail, v.shape)).astype(np.float32)
            m[rng.random(v.shape) < 1e-3] *= 20.0
            W[k] = (v * m).astype(np.float32)
    W["layoutlmv3.pos_embed"] = (W["layoutlmv3.pos_embed"] * np.float32(pos_gain)).astype(np.float32)
    W["layoutlmv3.embeddings.position_embeddings.weight"] = (W["layoutlmv3.embeddings.position_embeddings.weight"] * np.float32(pos_gain)).astype(np.float32)
    return W


def _small_split_cfg(pkg):
    ee = dict(exits=["text_visual_concat", 1, 2], encoder_layer_strategy="ramp")
    return ee, pkg.ModelConfig.tiny(EE_config=ee, hidden_size=256, num_attention_heads=4, intermediate_size=512, num_hidden_layers=3,
                                    coordinate_size=48, shape_size=32)


def test_split_precision_with_checkpoint_like_statistics(pkg, oracle):
    """LayerNorm gains up to x100 on single channels, heavy-tailed weights, x20 position embeddings: inside the documented
    range of the split planes, so the split path must hold the same tolerance as the f32 MFMA path."""
    ee, cfg = _small_split_cfg(pkg)
    W = _checkpoint_like(pkg, cfg, seed=31)
    docs = pkg.synth.make_documents(cfg, 6, seed=32, text_len=40, min_words=2)
    ref = oracle.forward_all(cfg, W, docs, ee["exits"], return_hidden_cls=True)
    args = (docs["input_ids"], docs["attention_mask"], docs["bbox"], docs["pixel_values"])
    errs = {}
    for prec in ("fp32", "split"):
        eng = pkg.EarlyExitEngine(cfg, max_docs=8, max_text_len=40, precision=prec, xprobe=False)
        eng.load_weights(W)
        out = eng.forward(*args, dump_all=True, want_all=True, want_hidden_cls=True, validate=True)
        errs[prec] = (float(np.abs(_np(out.all_logits) - ref["logits_store"]).max()),
                      float(np.abs(_np(out.hidden_cls) - ref["hidden_cls"]).max()))
        eng.close()
    scale = float(np.abs(ref["logits_store"]).max())
    # hidden states now reach a few hundred: the tolerance is stated relative to the magnitudes involved (1e-4 abs at O(1..10)
    # logits = 1e-5 relative), and the split path must not be worse than twice the f32 MFMA path's own distance to the oracle
    assert errs["split"][0] <= max(LOGIT_TOL, 1e-5 * scale, 2.0 * errs["fp32"][0]), (errs, scale)
    assert errs["fp32"][0] <= max(LOGIT_TOL, 1e-5 * scale), (errs, scale)


def test_split_precision_overflow_fails_loudly(pkg, oracle):
    """A LayerNorm gain of 5000 on one channel pushes |LayerNorm out| past the split planes' range (3750): the split path
    must raise — through validate=True, through check(), and on the next forward call — while precision "fp32" runs the same
    weights within tolerance."""
    ee, cfg = _small_split_cfg(pkg)
    W = _checkpoint_like(pkg, cfg, seed=41, gamma_outliers=(5000.0,), weight_tail=0.0, pos_gain=1.0)
    docs = pkg.synth.make_documents(cfg, 4, seed=42, text_len=40, min_words=2)
    args = (docs["input_ids"], docs["attention_mask"], docs["bbox"], docs["pixel_values"])
    eng = pkg.EarlyExitEngine(cfg, max_docs=4, max_text_len=40, precision="split", xprobe=False)
    eng.load_weights(W)
    with pytest.raises(pkg.capi.MMEEError, match="overflow"):
        eng.forward(*args, dump_all=True, validate=True)
    eng.forward(*args, dump_all=True)                         # not validated ...
    with pytest.raises(pkg.capi.MMEEError, match="overflow"):
        eng.check()                                           # ... check() reports it
    eng.forward(*args, dump_all=True)
    import torch
    torch.cuda.synchronize()
    with pytest.raises(pkg.capi.MMEEError, match="PREVIOUS forward"):
        eng.forward(*args, dump_all=True)                     # ... and so does the next call
    eng.close()
    ref = oracle.forward_all(cfg, W, docs, ee["exits"])
    e32 = pkg.EarlyExitEngine(cfg, max_docs=4, max_text_len=40, precision="fp32", xprobe=False)
    e32.load_weights(W)
    out = e32.forward(*args, dump_all=True, want_all=True, validate=True)
    scale = float(np.abs(ref["logits_store"]).max())
    assert float(np.abs(_np(out.all_logits) - ref["logits_store"]).max()) <= max(LOGIT_TOL, 2e-5 * scale)
    e32.close()
