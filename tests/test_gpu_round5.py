"""Round-5 GPU tests: determinism of the DEFAULT product path, the micro-batched engine, the bench-size batch under pytest, the explicit
schedule suggestion, criterion overrides written into ``model.config`` (EE/utils.py:62-78), shader-clock stamps, and N3 / N4 / the
LayoutLMv3-large shape against fixtures minted from the reference's own code (tests/golden/make_golden.py)."""
import importlib
import os
import sys

import numpy as np
import pytest

from .conftest import (LARGE_GATE_EE, ROOT, load_golden, report_measured, sweep_ref_inputs, temperature_ref_inputs)

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-4
CONFIG2_EE = dict(exits=[2, 4, 6, 8, 10], encoder_layer_strategy="ramp")


def _np(t):
    return None if t is None else t.detach().cpu().numpy()


def _gap_thresholds(conf, release):
    """Per-exit thresholds in gaps of the confidences of the documents that reach each exit (strict '>' well-posed)."""
    E1, n = conf.shape
    thr = np.full(E1, 2.0)
    active = np.ones(n, dtype=bool)
    for e in range(E1 - 1):
        c = np.sort(conf[e, active])
        if len(c) < 2:
            break
        k = min(max(int(round((1.0 - release) * len(c))), 1), len(c) - 1)
        lo, hi = max(1, k - 3), min(len(c) - 1, k + 3)
        j = lo + int(np.argmax(c[lo:hi + 1] - c[lo - 1:hi]))
        thr[e] = 0.5 * (c[j - 1] + c[j])
        active &= ~(conf[e] > thr[e])
    return thr


@pytest.fixture(scope="module")
def base_model(pkg):
    cfg = pkg.ModelConfig.base(EE_config=CONFIG2_EE)
    W = pkg.synth.make_weights(cfg, seed=1234, head_gain=6.0)
    return cfg, W


def _args(docs):
    return docs["input_ids"], docs["attention_mask"], docs["bbox"], docs["pixel_values"]


def test_default_engine_is_bit_reproducible_with_interleaved_work(pkg, oracle, base_model):
    """VERDICT r04 item 3.  The DEFAULT engine (X-space probe on, no pinned schedule, nothing passed but inputs and thresholds) run 50 times on
    the same batch, with other batches, other thresholds, dump-all passes and synchronisation points in between: every run returns the SAME
    bits (logits, exit indices, confidences) -- the schedule is a function of the call, not of timing or of earlier forwards (rounds 2-4
    derived it from whichever earlier forward had finished).  north_star: "exit-layer indices are bit-exact"; the reference's policy is
    deterministic (EE/policy.py:28-45)."""
    import torch
    cfg, W = base_model
    B = 48
    docs = pkg.synth.make_documents(cfg, B, seed=501, text_len=512)
    other = pkg.synth.make_documents(cfg, 31, seed=502, text_len=512)
    eng = pkg.EarlyExitEngine(cfg, max_docs=B, max_text_len=512)          # every default
    assert eng.xprobe_default
    eng.load_weights(W)
    a = tuple(torch.from_numpy(x).cuda() for x in _args(docs))
    o_ = tuple(torch.from_numpy(x).cuda() for x in _args(other))
    full = eng.forward(*a, dump_all=True, want_all=True)
    conf = oracle.softmax64(_np(full.all_logits).astype(np.float64)).max(-1)
    thr = _gap_thresholds(conf, 0.25)
    first = eng.forward(*a, thresholds=thr)
    plan0 = eng.layer_plan()
    ref = tuple(_np(t).copy() for t in (first.logits, first.exit_layer, first.confidence))
    assert len(np.unique(ref[1])) >= 4
    rng = np.random.default_rng(3)
    for it in range(50):
        kind = it % 5
        if kind == 0:
            eng.forward(*o_, thresholds=float(rng.uniform(0.2, 0.9)))              # another batch, a global threshold: other stage populations
        elif kind == 1:
            eng.forward(*o_, dump_all=True)
            torch.cuda.synchronize()
        elif kind == 2:
            eng.forward(*a, thresholds=np.minimum(thr * rng.uniform(0.5, 1.0), 2.0))   # the same batch under other thresholds
        elif kind == 3:
            for _ in range(3):
                eng.forward(*o_, thresholds=0.0)                                       # everybody leaves at the first exit
        out = eng.forward(*a, thresholds=thr)
        if it % 7 == 0:
            torch.cuda.synchronize()
        got = tuple(_np(t) for t in (out.logits, out.exit_layer, out.confidence))
        for g_, r_ in zip(got, ref):
            assert np.array_equal(g_, r_), f"run {it}: the default path returned different bits"
        assert eng.layer_plan() == plan0
    eng.check()
    eng.close()


def test_suggested_schedule_is_explicit_and_results_do_not_depend_on_it(pkg, oracle, base_model):
    """ee_suggest_probe_mask / EarlyExitEngine.pin_schedule(): the cost model is an explicit query on the LAST forward's stage populations --
    nobody leaves -> no layer is worth probing; many leave -> every exit layer is; a dump-all forward cannot be priced (raises); a pinned
    subset gives the whole-layer bits (K | V probe); pin_schedule(False) returns to "every exit layer"."""
    import torch
    cfg, W = base_model
    B = 96
    docs = pkg.synth.make_documents(cfg, B, seed=77, text_len=512)
    eng = pkg.EarlyExitEngine(cfg, max_docs=B, max_text_len=512, xprobe=False)
    eng.load_weights(W)
    a = tuple(torch.from_numpy(x).cuda() for x in _args(docs))
    full = eng.forward(*a, dump_all=True, want_all=True)
    with pytest.raises(pkg.capi.MMEEError, match="dump"):
        eng.pin_schedule()
    conf = oracle.softmax64(_np(full.all_logits).astype(np.float64)).max(-1)
    exit_layers = [l - 1 for l in CONFIG2_EE["exits"]]
    # nobody leaves before the final classifier
    eng.forward(*a, thresholds=2.0)
    assert eng.pin_schedule() == []
    assert eng.pin_schedule(False) is None
    # half of the arrivals leave at every exit
    thr = _gap_thresholds(conf, 0.5)
    dflt = eng.forward(*a, thresholds=thr)
    plan = eng.layer_plan()
    assert [l for l, d in enumerate(plan["docs_probe"]) if d > 0] == exit_layers + [cfg.num_hidden_layers - 1]      # default: every decision layer
    sugg = eng.pin_schedule()          # the early, populous stages pay; a stage of a dozen documents does not cover the probe's fixed cost
    assert sugg and sugg[0] == 1 and set(sugg) <= set(exit_layers) and sugg == sorted(sugg), sugg
    # a hand-pinned subset: mixed probe-first / whole exit layers -- same bits as whole layers and as the default
    part = eng.pin_schedule([1, 7])
    assert part == [1, 7]
    mixed = eng.forward(*a, thresholds=thr)
    assert [l for l, d in enumerate(eng.layer_plan()["docs_probe"]) if d > 0] == [1, 7, cfg.num_hidden_layers - 1]
    whole = eng.forward(*a, thresholds=thr, whole_layers=True)
    for x, y, z in zip((dflt.logits, dflt.exit_layer, dflt.confidence), (mixed.logits, mixed.exit_layer, mixed.confidence),
                       (whole.logits, whole.exit_layer, whole.confidence)):
        assert np.array_equal(_np(x), _np(y)) and np.array_equal(_np(x), _np(z))
    # the X-space probe is priced differently (a leaving row also skips Q | K | V) but is the same kind of answer
    assert set(eng.pin_schedule(None, xprobe=True)) <= set(exit_layers)
    eng.pin_schedule(False)
    eng.close()


@pytest.mark.parametrize("xprobe", [False, True])
def test_micro_batched_engine_is_bit_identical(pkg, oracle, base_model, xprobe):
    """VERDICT r04 item 1(b).  MicroBatchedEngine (slices of the batch on two / three handles and HIP streams) against ONE handle on one stream:
    the same logits, exit indices and confidences bit for bit, under the default schedule, under a pinned mixed schedule, in dump-all mode
    and run serially; statistics are the sums over the slices."""
    import torch
    cfg, W = base_model
    B = 65                                                   # odd: slices of 33 + 32, and 22 + 22 + 21
    docs = pkg.synth.make_documents(cfg, B, seed=4242, text_len=512)
    one = pkg.EarlyExitEngine(cfg, max_docs=B, max_text_len=512, xprobe=xprobe)
    one.load_weights(W)
    a = tuple(torch.from_numpy(x).cuda() for x in _args(docs))
    full = one.forward(*a, dump_all=True, want_all=True)
    conf = oracle.softmax64(_np(full.all_logits).astype(np.float64)).max(-1)
    thr = _gap_thresholds(conf, 0.3)
    for n in (2, 3):
        mb = pkg.MicroBatchedEngine(cfg, max_docs=B, max_text_len=512, xprobe=xprobe, micro_batches=n)
        mb.load_weights(W)
        assert mb.split_sizes(B) == ([33, 32] if n == 2 else [22, 22, 21])
        for pin in (False, [3, 9]):
            one.pin_schedule(pin)
            mb.pin_schedule(pin)
            r = one.forward(*a, thresholds=thr)
            sc_one, fl_one, plan_one = one.stage_counts(), one.flops(), one.layer_plan()
            for kw in (dict(), dict(serial=True)):
                o = mb.forward(*a, thresholds=thr, **kw)
                for x, y in zip((r.logits, r.exit_layer, r.confidence), (o.logits, o.exit_layer, o.confidence)):
                    assert np.array_equal(_np(x), _np(y)), (n, pin, kw)
                assert mb.stage_counts() == sc_one
                assert mb.layer_plan()["rows_main"] == plan_one["rows_main"] and mb.layer_plan()["docs_probe"] == plan_one["docs_probe"]
                np.testing.assert_allclose(mb.flops()["total"], fl_one["total"], rtol=1e-12)
        d = mb.forward(*a, dump_all=True, want_all=True, want_head=True)
        assert np.array_equal(_np(d.all_logits), _np(full.all_logits))
        assert d.head_logits.shape == (len(CONFIG2_EE["exits"]), B, cfg.num_labels)
        small = mb.forward(*(t[:1] for t in a), thresholds=thr)          # one document: one slice
        assert np.array_equal(_np(small.logits), _np(r.logits)[:1]) and mb.split_sizes(1) == [1]
        mb.check()
        mb.close()
    one.close()


def test_bench_batch_1024_and_its_edges(pkg, oracle, base_model):
    """VERDICT r04 item 6.  The batch bench.py times (LayoutLMv3-base, exits 2/4/6/8/10 + final, B = 1024 ragged documents at T = 512: 22 GB of
    workspace, buffers beyond 2^31 bytes) under pytest, through properties no oracle run is needed for: an early-exit row is BIT-identical to
    the dump-all row at the exit the policy picks; permuting the batch permutes the outputs; the stage populations are the survivors; no
    error flag (validate=True).  Then the edges of the strong-scaling tail: B = 1023 (a partial last tile in every launch) and B = 1 equal the
    corresponding rows of the full batch bit for bit, the default X-space probe leaves at the same exits within the logit tolerance, and the
    micro-batched engine (what bench.py runs) returns the one-handle bits."""
    import torch
    cfg, W = base_model
    B = 1024
    docs = pkg.synth.make_documents(cfg, B, seed=2234, text_len=512)
    eng = pkg.EarlyExitEngine(cfg, max_docs=B, max_text_len=512, xprobe=False)
    eng.load_weights(W)
    a = tuple(torch.from_numpy(x).cuda() for x in _args(docs))
    full = eng.forward(*a, dump_all=True, want_all=True, validate=True)
    store = _np(full.all_logits).astype(np.float64)
    assert not np.isnan(store).any()
    conf = oracle.softmax64(store).max(-1)
    thr = _gap_thresholds(conf, 0.2)
    assert np.abs(conf[:-1] - thr[:-1, None]).min() > 1e-7
    ex_ref, pred_ref, _ = oracle.policy_scan(store, thr)
    out = eng.forward(*a, thresholds=thr, validate=True)
    ex = _np(out.exit_layer)
    assert np.array_equal(ex, ex_ref) and len(np.unique(ex)) == 6
    got = _np(out.logits)
    assert np.array_equal(got, _np(full.all_logits)[ex, np.arange(B)])
    sc = eng.stage_counts()
    assert sc["docs"] == [int((ex >= e).sum()) for e in range(len(sc["docs"]))]
    lens = docs["attention_mask"].sum(1) + 197
    assert sc["rows"][0] == int(lens.sum()) and sc["rows"][-1] == int(lens[ex >= len(sc["docs"]) - 1].sum())
    cf = _np(out.confidence)
    # permutation
    perm = np.random.default_rng(11).permutation(B)
    pt = torch.from_numpy(perm).cuda()
    outp = eng.forward(*(t[pt] for t in a), thresholds=thr)
    assert np.array_equal(_np(outp.exit_layer), ex[perm]) and np.array_equal(_np(outp.logits), got[perm])
    assert np.array_equal(_np(outp.confidence), cf[perm])
    # B = 1023 and B = 1: the same documents in smaller launches
    for n in (1023, 1):
        o = eng.forward(*(t[:n] for t in a), thresholds=thr, validate=True)
        assert np.array_equal(_np(o.exit_layer), ex[:n]) and np.array_equal(_np(o.logits), got[:n]) and np.array_equal(_np(o.confidence), cf[:n])
    last = eng.forward(*(t[B - 1:] for t in a), thresholds=thr)
    assert np.array_equal(_np(last.logits), got[B - 1:])
    # whole layers: the reference's order of operations
    w = eng.forward(*a, thresholds=thr, whole_layers=True)
    assert np.array_equal(_np(w.logits), got) and np.array_equal(_np(w.exit_layer), ex)
    # the engine's default probe (X space): same exits, logits within the bar
    xp = eng.forward(*a, thresholds=thr, xprobe=True, validate=True)
    assert np.array_equal(_np(xp.exit_layer), ex)
    err = float(np.abs(_np(xp.logits) - got).max())
    report_measured("test_bench_batch_1024_and_its_edges", "max |dlogit| X-space probe vs whole layers, B = 1024", err)
    assert err < LOGIT_TOL
    xp_bits = _np(xp.logits).copy()
    eng.close()
    del eng
    torch.cuda.empty_cache()
    mb = pkg.MicroBatchedEngine(cfg, max_docs=B, max_text_len=512, micro_batches=2)
    mb.load_weights(W)
    o = mb.forward(*a, thresholds=thr)
    assert np.array_equal(_np(o.exit_layer), ex) and np.array_equal(_np(o.logits), xp_bits)
    o = mb.forward(*a, thresholds=thr, xprobe=False)
    assert np.array_equal(_np(o.logits), got) and np.array_equal(_np(o.confidence), cf)
    o = mb.forward(*(t[:1023] for t in a), thresholds=thr, xprobe=False)
    assert np.array_equal(_np(o.logits), got[:1023])
    mb.check()
    mb.close()


def test_criterion_written_into_model_config_is_honoured(pkg):
    """VERDICT r04 item 8 / EE/utils.py:62-78: ``load_assets`` writes ``model.config.exit_config["inference_strategy"]`` and
    ``["global_threshold"]`` AFTER the model exists.  The mirror re-reads the dictionary at every call: the criterion values of ``forward`` and
    the decisions of ``early_exit`` follow the override, exactly as a model BUILT with that strategy computes them."""
    import torch
    ee_max = dict(exits=[1, 3], encoder_layer_strategy="ramp", inference_strategy="max_confidence", exit_head_num_layers=1)
    ee_ent = dict(ee_max, inference_strategy="entropy")
    cfg_max, cfg_ent = pkg.ModelConfig.tiny(EE_config=ee_max), pkg.ModelConfig.tiny(EE_config=ee_ent)
    W = pkg.synth.make_weights(cfg_max, seed=7)
    docs = pkg.synth.make_documents(cfg_max, 6, seed=11, text_len=48, min_words=3)
    t = {k: torch.from_numpy(v).cuda() for k, v in docs.items() if k != "labels"}
    m = pkg.LayoutLMv3EEForSequenceClassification(cfg_max, weights=W, max_docs=8, max_text_len=48)
    built = pkg.LayoutLMv3EEForSequenceClassification(cfg_ent, weights=W, max_docs=8, max_text_len=48)
    before = m.forward(**t)
    want = built.forward(**t)
    assert not np.array_equal(_np(before.exit_states[0][1]), _np(want.exit_states[0][1]))
    m.config.exit_config["inference_strategy"] = "entropy"              # EE/utils.py:74-76
    after = m.forward(**t)
    for j in range(2):
        assert np.array_equal(_np(after.exit_states[j][1]), _np(want.exit_states[j][1]))
        assert np.array_equal(_np(after.exit_states[j][0]), _np(before.exit_states[j][0]))      # the logits do not depend on the criterion
        lg = after.exit_states[j][0].double()
        e = torch.exp(lg)
        ent = torch.log(e.sum(1)) - (lg * e).sum(1) / e.sum(1)                                 # EE/models/EE_modules.py:155-160
        np.testing.assert_allclose(_np(after.exit_states[j][1]), _np(ent), rtol=0, atol=2e-5)
    assert np.array_equal(_np(after.exit_criteria[-1]), _np(want.exit_criteria[-1]))
    # the threshold of the same dictionary steers early_exit; entropy leaves when the criterion is BELOW it
    crit0 = np.sort(_np(after.exit_states[0][1]))
    m.config.exit_config["global_threshold"] = float(0.5 * (crit0[2] + crit0[3]))
    built.config.exit_config["global_threshold"] = m.config.exit_config["global_threshold"]
    ee1, ee2 = m.early_exit(**t), built.early_exit(**t)
    assert np.array_equal(_np(ee1.exit_layer), _np(ee2.exit_layer)) and np.array_equal(_np(ee1.logits), _np(ee2.logits))
    assert int((_np(ee1.exit_layer) == 0).sum()) == 3
    m.config.exit_config["inference_strategy"] = "patience"             # not built: the reference's get_sign raises for it too
    with pytest.raises(Exception):
        m.forward(**t)
    m.engine.close()
    built.engine.close()


def test_clock_stamps_and_one_term_refusals(pkg, base_model):
    import torch
    cfg, W = base_model
    docs = pkg.synth.make_documents(cfg, 16, seed=5, text_len=512)
    eng = pkg.EarlyExitEngine(cfg, max_docs=16, max_text_len=512)
    eng.load_weights(W)
    a = tuple(torch.from_numpy(x).cuda() for x in _args(docs))
    eng.forward(*a, dump_all=True)
    s0 = eng.clock_stamp()
    for _ in range(6):
        eng.forward(*a, dump_all=True)
    s1 = eng.clock_stamp()
    ghz, per = eng.clock_ghz(s0, s1)
    report_measured("test_clock_stamps_and_one_term_refusals", "shader clock over six small forwards (GHz)", ghz)
    filled = ((s0.cpu().numpy().reshape(-1, 2)[:, 1] > 0) & (s1.cpu().numpy().reshape(-1, 2)[:, 1] > 0)).sum()
    report_measured("test_clock_stamps_and_one_term_refusals", "CU slots stamped in both", float(filled))
    assert len(per) == 8 and filled >= 128, (per, filled)   # every XCD, and most of the 256 CUs, seen by both stamps
    assert 0.3 < ghz < 2.6 and max(per) - min(per) < 0.15 * ghz, per      # same-CU differences only: the XCDs agree
    eng.close()
    # MMEE_FLAG_ONE_TERM exists for split-precision LayoutLMv3 handles only: refused elsewhere instead of a silent mix
    f32 = pkg.EarlyExitEngine(cfg, max_docs=16, max_text_len=512, precision="fp32")
    f32.load_weights(W)
    with pytest.raises(pkg.capi.MMEEError, match="ONE_TERM"):
        f32.forward(*a, dump_all=True, one_term=True)
    f32.close()
    dcfg = pkg.ModelConfig.dit_base(EE_config=dict(exits=[2, 4], encoder_layer_strategy="ramp"))
    dW = pkg.synth.make_weights_beit(dcfg, seed=3)
    dit = pkg.EarlyExitEngine(dcfg, max_docs=4)
    dit.load_weights(dW)
    with pytest.raises(pkg.capi.MMEEError, match="ONE_TERM"):
        dit.forward(pixel_values=a[3][:4], thresholds=0.5, one_term=True)
    dit.forward(pixel_values=a[3][:4], thresholds=0.5, validate=True)
    dit.close()


# ---- N3 / N4 / large shape: fixtures minted from the reference's own code -----------------------------------------------------------------
def test_threshold_sweep_against_reference_vectors(pkg):
    """VERDICT r04 item 4.  ee_threshold_sweep on the CSF table, thresholds (generate_thresholds, its own np.random.seed(42)) and labels of
    tests/golden/sweep_ref.npz: exit histograms, accuracy and mean exit of every threshold vector equal what the reference's
    check_2D_threshold (EE/large_scale.py:49-50 == EE/thresh.py:184-185) and evaluate_exit_logits expressions (:88-92) gave, bit for bit, on
    both kernels (direct with histogram, integer ranks without), including the "no exit fires -> exit 0" rows; ee_msp_table agrees with
    scipy's softmax to the last bits."""
    g = load_golden("sweep_ref")
    logits, refs = sweep_ref_inputs()
    conf_d, corr_d = pkg.sweep.msp_table(logits, refs)
    np.testing.assert_allclose(_np(conf_d), g["conf"], rtol=1e-15, atol=0)
    assert np.array_equal(_np(corr_d), g["correct"])
    acc, mex, hist = pkg.sweep.threshold_sweep(g["conf"], g["correct"], g["thresholds"], want_hist=True)
    assert np.array_equal(_np(hist), g["hist"])
    assert np.array_equal(_np(acc), g["accuracy"]) and np.array_equal(_np(mex), g["mean_exit"])
    acc2, mex2, h2 = pkg.sweep.threshold_sweep(g["conf"], g["correct"], g["thresholds"])       # ranked kernels (many vectors, no histogram)
    assert h2 is None
    assert np.array_equal(_np(acc2), g["accuracy"]) and np.array_equal(_np(mex2), g["mean_exit"])
    V = int(g["n_generated"])
    assert float(_np(mex)[V]) == 0.0 and int(_np(hist)[V, 0]) == logits.shape[1]


def test_temperature_fit_against_reference_scaler(pkg):
    """VERDICT r04 item 4.  ee_temperature_fit against the temperatures the reference's TemperatureScaler (EE/generic_scaling.py:37-111, driven
    as EE/eval.py:298-329 drives it) found on the same seeded logits: <= 2e-4 relative (a Newton iteration on the exact gradient against
    L-BFGS-B's stopping rule), and never a worse NLL than the reference's optimum."""
    g = load_golden("temperature_ref")
    logits, refs = temperature_ref_inputs()
    res = pkg.calibration.fit_temperatures(logits, refs)
    rel = np.abs(res["temperature"] - g["temperature"]) / g["temperature"]
    report_measured("test_temperature_fit_against_reference_scaler", "max relative |T - T_reference|", float(rel.max()))
    assert (rel <= 2e-4).all(), (res["temperature"], g["temperature"])
    assert (res["nll"] <= g["nll_after"] + 1e-12).all()
    ts = pkg.calibration.TemperatureScaler()
    for e in range(logits.shape[0]):
        ts.fit(refs, logits[e])
        assert abs(ts.temperature[0] - g["temperature"][e]) <= 2e-4 * g["temperature"][e]


@pytest.mark.parametrize("precision", ["split", "fp32"])
def test_large_shape_gate_golden(pkg, oracle, precision):
    """VERDICT r04 item 5.  BASELINE configs[2] shape -- LayoutLMv3-large, 24 layers, 16 heads x 64, I = 4096, gate exits after layers 1..23 --
    against the composed reference (stock HF encoder + the reference's own LayoutLMv3Exit heads and classifier-on-gate-input wiring,
    EE/models/LayoutLMv3.py:764-792), in both precisions: dump-all logits of all 24 exits, the 2-way gate heads, the CLS row of every layer; then
    early exit under the fixture's thresholds with the reference Policy's exits -- whole layers, K | V probe and (split) the X-space probe,
    which runs the 16-head x 1024 kernels."""
    import torch
    g = load_golden("large_gate")
    cfg = pkg.ModelConfig.large(EE_config=LARGE_GATE_EE)
    W = pkg.synth.make_weights(cfg, seed=int(g["seed_w"]))
    docs = pkg.synth.make_documents(cfg, int(g["n_docs"]), seed=int(g["seed_docs"]), text_len=int(g["text_len"]))
    eng = pkg.EarlyExitEngine(cfg, max_docs=4, max_text_len=512, precision=precision, xprobe=False)
    eng.load_weights(W)
    a = tuple(torch.from_numpy(x).cuda() for x in _args(docs))
    out = eng.forward(*a, dump_all=True, want_all=True, want_head=True, want_hidden_cls=True, validate=True)
    e_store = float(np.abs(_np(out.all_logits) - g["logits_store"]).max())
    e_head = float(np.abs(_np(out.head_logits) - g["exit_logits"]).max())
    e_cls = float(np.abs(_np(out.hidden_cls) - g["hidden_cls"]).max())
    report_measured(f"test_large_shape_gate_golden[{precision}]", "max |d logits_store| vs composed reference", e_store)
    report_measured(f"test_large_shape_gate_golden[{precision}]", "max |d CLS row| over 25 layers", e_cls)
    assert e_store < LOGIT_TOL and e_head < LOGIT_TOL and e_cls < 1e-4
    for i in range(4):
        thr = float(g[f"pol_thr{i}"])
        store = g["logits_store"]
        c = oracle.softmax64(store).max(-1)
        if np.abs(c - thr).min() < 1e-4:                    # a confidence on the threshold: the exit index is not well-posed at 1e-4
            continue
        scheds = [dict(whole_layers=True), dict()] + ([dict(xprobe=True)] if precision == "split" else [])
        for kw in scheds:
            o = eng.forward(*a, thresholds=thr, validate=True, **kw)
            assert np.array_equal(_np(o.exit_layer), g[f"pol_exits{i}"]), (i, kw)
            np.testing.assert_allclose(_np(o.logits), g[f"pol_pred{i}"], rtol=0, atol=LOGIT_TOL)
    if precision == "split":
        eng.forward(*a, thresholds=2.0, xprobe=True)
        assert all(eng.layer_plan()["docs_probe"][l] == 2 for l in range(cfg.num_hidden_layers))      # every layer ends in a decision
    eng.close()


# ---- two ranks with DIFFERENT stage mixes run rank 0's pinned plan ---------------------------------------------------------------------------
_RANK_CODE = r'''
import importlib, json, os, sys
import numpy as np
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
pkg = importlib.import_module("multi-modal-early-exit_amd")
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + os.environ["MASTER_PORT"], rank=rank, world_size=world)
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
ee = dict(exits=[1, 2, 3], encoder_layer_strategy="ramp")
cfg = pkg.ModelConfig.tiny(EE_config=ee, hidden_size=256, num_attention_heads=4, intermediate_size=512, num_hidden_layers=4, coordinate_size=48, shape_size=32)
W = pkg.synth.make_weights(cfg, seed=9, head_gain=6.0)
N = 41
docs = pkg.synth.make_documents(cfg, N, seed=13, text_len=48, min_words=3)
idx = pkg.dist.shard_indices(N, rank, world)
eng = pkg.EarlyExitEngine(cfg, max_docs=N, max_text_len=48, precision="split", xprobe=False)
eng.load_weights(W)
mine = tuple(torch.from_numpy(docs[k][idx]).to(dev) for k in ("input_ids", "attention_mask", "bbox", "pixel_values"))
thr = pkg.dist.broadcast_array(np.array({thr!r}), 0, device=dev)
eng.forward(*mine, thresholds=thr)
own = eng.stage_counts()["docs"]
plan = np.full(8, -1, dtype=np.int64)
if rank == 0:
    pl = {plan!r}
    plan[:len(pl)] = pl
plan = pkg.dist.broadcast_array(plan, 0, device=dev)
pinned = eng.pin_schedule([int(x) for x in plan if x >= 0])
def run_local(ix):
    o = eng.forward(*mine, thresholds=thr)
    return pkg.dist.pack_results(o.logits, o.exit_layer, o.confidence)
rows = pkg.dist.run_sharded(run_local, N, rank, world)
lp = eng.layer_plan()
launches = [int(d > 0) for d in lp["docs_probe"]]
gathered = [None] * world
dist.all_gather_object(gathered, dict(rank=rank, own_stage_docs=own, launches=launches, pinned=pinned))
if rank == 0:
    lg, ex, cf = pkg.dist.unpack_results(rows)
    assert ex.dtype == torch.int32
    np.savez({out!r}, logits=lg.cpu().numpy(), exits=ex.cpu().numpy(), conf=cf.cpu().numpy(), meta=json.dumps(gathered))
eng.close()
dist.destroy_process_group()
'''


def test_two_ranks_with_different_mixes_run_rank0s_plan(pkg, oracle, tmp_path):
    """VERDICT r04 item 7(b).  Two gloo ranks share the one GPU of the box (the flow of bench.py --gpus N: thresholds and the pinned exit-layer
    plan are rank 0's, broadcast), their shards have DIFFERENT exit mixes, and still every rank issues the same launch list
    (ee_last_layer_plan: which layers were probed first) and the gathered (f32 logits, i32 exit, f32 confidence) rows equal the
    single-process run of all documents bit for bit."""
    import json
    import socket
    import subprocess
    import torch
    ee = dict(exits=[1, 2, 3], encoder_layer_strategy="ramp")
    cfg = pkg.ModelConfig.tiny(EE_config=ee, hidden_size=256, num_attention_heads=4, intermediate_size=512, num_hidden_layers=4,
                               coordinate_size=48, shape_size=32)
    W = pkg.synth.make_weights(cfg, seed=9, head_gain=6.0)
    N = 41
    docs = pkg.synth.make_documents(cfg, N, seed=13, text_len=48, min_words=3)
    eng = pkg.EarlyExitEngine(cfg, max_docs=N, max_text_len=48, precision="split", xprobe=False)
    eng.load_weights(W)
    a = _args(docs)
    full = eng.forward(*a, dump_all=True, want_all=True)
    conf = oracle.softmax64(_np(full.all_logits).astype(np.float64)).max(-1)
    thr = _gap_thresholds(conf, 0.3)
    plan = [0, 2]                                            # layer 0 and 2 probed first, layer 1 whole (layer 3, the last, is always the probe alone)
    eng.pin_schedule(plan)
    ref = eng.forward(*a, thresholds=thr)
    ref_launch = [int(d > 0) for d in eng.layer_plan()["docs_probe"]]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    outp = str(tmp_path / "gathered.npz")
    code = _RANK_CODE.format(root=ROOT, thr=[float(t) for t in thr], plan=plan, out=outp)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    g = np.load(outp)
    meta = json.loads(str(g["meta"]))
    assert meta[0]["own_stage_docs"] != meta[1]["own_stage_docs"]                    # the two shards really leave differently
    assert all(m["launches"] == ref_launch and m["pinned"] == plan for m in meta)      # ... and still issue the same launches
    assert np.array_equal(g["exits"], _np(ref.exit_layer)) and g["exits"].dtype == np.int32
    assert np.array_equal(g["logits"], _np(ref.logits)) and np.array_equal(g["conf"], _np(ref.confidence))
    eng.close()


@pytest.mark.parametrize("shape", ["tiny_fp32", "h256_split", "base_split"])
def test_output_attentions_and_head_mask(pkg, oracle, shape):
    """VERDICT r04 "missing" 5: the last two arguments of the reference signature (EE/models/LayoutLMv3.py:382-385, 631-641, 219-220).
    ``output_attentions=True`` returns L tensors (B, heads, S, S) of attention probabilities -- against the stock HF encoder's on the tiny
    configuration (tests/golden/tiny_attentions.npz) and against the oracle at a split-precision shape and at LayoutLMv3-base; ``head_mask``
    ((heads,) or (L, heads), as get_head_mask accepts) scales the probabilities and changes the logits as the oracle's restatement of the 4.26
    line does.  Both are side kernels of the dump-all forward (csrc/attention_maps.hip): `early_exit` refuses them."""
    import torch
    if shape == "tiny_fp32":
        g = load_golden("tiny_attentions")
        ee = dict(exits=[1, 3], encoder_layer_strategy="ramp", inference_strategy="max_confidence")
        cfg = pkg.ModelConfig.tiny(EE_config=ee)
        W = pkg.synth.make_weights(cfg, seed=int(g["seed_w"]))
        docs = {k[3:]: v for k, v in g.items() if k.startswith("in_")}
        T = 48
    elif shape == "h256_split":
        g = None
        ee = dict(exits=[1, 2], encoder_layer_strategy="ramp")
        cfg = pkg.ModelConfig.tiny(EE_config=ee, hidden_size=256, num_attention_heads=4, intermediate_size=512, num_hidden_layers=3,
                                   coordinate_size=48, shape_size=32)
        W = pkg.synth.make_weights(cfg, seed=5)
        docs = pkg.synth.make_documents(cfg, 4, seed=6, text_len=48, min_words=3)
        T = 48
    else:
        g = None
        ee = dict(exits=[2], encoder_layer_strategy="ramp")
        cfg = pkg.ModelConfig.base(EE_config=ee, num_hidden_layers=3)
        W = pkg.synth.make_weights(cfg, seed=8)
        docs = pkg.synth.make_documents(cfg, 2, seed=9, text_len=512)
        T = 512
    L, nh = cfg.num_hidden_layers, cfg.num_attention_heads
    m = pkg.LayoutLMv3EEForSequenceClassification(cfg, weights=W, max_docs=4, max_text_len=T)
    t = {k: torch.from_numpy(v).cuda() for k, v in docs.items() if k != "labels"}
    ref = oracle.forward_all(cfg, W, docs, ee["exits"], return_attentions=True)
    out = m.forward(**t, output_attentions=True)
    assert isinstance(out.attentions, tuple) and len(out.attentions) == L
    S = T + (cfg.input_size // cfg.patch_size) ** 2 + 1
    att = np.stack([_np(a) for a in out.attentions])
    assert att.shape == (L, docs["pixel_values"].shape[0], nh, S, S)
    err = float(np.abs(att - ref["attentions"]).max())
    report_measured(f"test_output_attentions_and_head_mask[{shape}]", "max |d attention probability| vs oracle", err)
    assert err < 2e-6 and abs(float(att.sum(-1).mean()) - 1.0) < 1e-6
    if g is not None:
        np.testing.assert_allclose(att, g["attentions"], rtol=0, atol=2e-6)                          # ... and vs the stock HF encoder itself
    pad = docs["attention_mask"][0] == 0
    if pad.any():
        assert (att[:, 0, :, :, :T][..., pad] == 0).all()                                           # masked keys: exactly 0
    np.testing.assert_allclose(_np(out.logits), ref["logits"], rtol=0, atol=LOGIT_TOL)
    plain = m.forward(**t)
    # asking for the maps changes nothing else (the maps are written in the padded layout, so that run keeps the pad rows: same values to rounding)
    assert plain.attentions is None
    np.testing.assert_allclose(_np(plain.logits), _np(out.logits), rtol=0, atol=2e-5)
    # head mask: per layer and head, and the (heads,) form broadcast over the layers
    rng = np.random.default_rng(4)
    hm = rng.choice([0.0, 1.0, 0.5], size=(L, nh)).astype(np.float32)
    hm[0, 0] = 0.0
    for mask in (hm, hm[1]):
        refm = oracle.forward_all(cfg, W, docs, ee["exits"], return_attentions=True, head_mask=mask)
        om = m.forward(**t, head_mask=torch.from_numpy(mask), output_attentions=True)
        np.testing.assert_allclose(np.stack([_np(a) for a in om.attentions]), refm["attentions"], rtol=0, atol=2e-6)
        e_l = float(np.abs(_np(om.logits) - refm["logits"]).max())
        report_measured(f"test_output_attentions_and_head_mask[{shape}]", "max |dlogit| under a head mask vs oracle", e_l)
        assert e_l < LOGIT_TOL
        for j in range(len(ee["exits"])):
            np.testing.assert_allclose(_np(om.exit_states[j][0]), refm["exit_logits"][j], rtol=0, atol=LOGIT_TOL)
        only = m.forward(**t, head_mask=mask)                                                       # numpy mask, no maps
        assert only.attentions is None
        np.testing.assert_allclose(_np(only.logits), _np(om.logits), rtol=0, atol=2e-5)
    assert float(np.abs(_np(om.logits) - _np(out.logits)).max()) > 1e-4                             # the mask really acts
    with pytest.raises(ValueError):
        m.engine.forward(**t, thresholds=0.5, head_mask=hm)                                         # not part of the early-exit path
    with pytest.raises(ValueError):
        m.forward(**t, head_mask=np.ones((L + 1, nh), np.float32))
    m.engine.close()


def test_model_wrapper_with_micro_batches(pkg):
    """``LayoutLMv3EEForSequenceClassification(..., micro_batches=2)``: the reference-signature ``forward`` (dump-all, every output field, hidden
    states, attention maps) and ``early_exit`` on two handles / streams return the bits of the one-handle model."""
    import torch
    ee = dict(exits=["text_visual_concat", 1, 2], encoder_layer_strategy="ramp")
    cfg = pkg.ModelConfig.tiny(EE_config=ee, hidden_size=256, num_attention_heads=4, intermediate_size=512, num_hidden_layers=3,
                               coordinate_size=48, shape_size=32)
    W = pkg.synth.make_weights(cfg, seed=7, head_gain=6.0)
    docs = pkg.synth.make_documents(cfg, 7, seed=11, text_len=48, min_words=3)
    t = {k: torch.from_numpy(v).cuda() for k, v in docs.items() if k != "labels"}
    one = pkg.LayoutLMv3EEForSequenceClassification(cfg, weights=W, max_docs=8, max_text_len=48)
    two = pkg.LayoutLMv3EEForSequenceClassification(cfg, weights=W, max_docs=8, max_text_len=48, micro_batches=2)
    assert type(two.engine).__name__ == "MicroBatchedEngine" and two.engine.split_sizes(7) == [4, 3]
    a, b = one.forward(**t, labels=torch.from_numpy(docs["labels"])), two.forward(**t, labels=torch.from_numpy(docs["labels"]))
    assert np.array_equal(_np(a.logits), _np(b.logits)) and float(a.loss) == float(b.loss)
    for j in range(3):
        assert np.array_equal(_np(a.exit_states[j][0]), _np(b.exit_states[j][0])) and np.array_equal(_np(a.exit_states[j][1]), _np(b.exit_states[j][1]))
    ha, hb = one.forward(**t, output_hidden_states=True, output_attentions=True), two.forward(**t, output_hidden_states=True, output_attentions=True)
    assert len(hb.hidden_states) == 4 and all(np.array_equal(_np(x), _np(y)) for x, y in zip(ha.hidden_states, hb.hidden_states))
    assert len(hb.attentions) == 3 and all(np.array_equal(_np(x), _np(y)) for x, y in zip(ha.attentions, hb.attentions))
    crit = np.sort(_np(a.exit_states[1][1]))
    thr = [2.0, float(0.5 * (crit[3] + crit[4])), 2.0, 2.0]            # three of the seven documents leave at the first encoder exit
    ea, eb = one.early_exit(**t, thresholds=thr), two.early_exit(**t, thresholds=thr)
    for x, y in zip((ea.logits, ea.exit_layer, ea.confidence), (eb.logits, eb.exit_layer, eb.confidence)):
        assert np.array_equal(_np(x), _np(y))
    assert len(np.unique(_np(eb.exit_layer))) >= 2
    two.config.exit_config["inference_strategy"] = "entropy"            # the override reaches both handles
    one.config.exit_config["inference_strategy"] = "entropy"
    assert np.array_equal(_np(one.forward(**t).exit_states[0][1]), _np(two.forward(**t).exit_states[0][1]))
    one.engine.close()
    two.engine.close()
