"""Round 6 (GPU): what VERDICT r05 / ADVICE r05 asked for -- the DiT wrapper on micro-batches, the reference's tuple order for
``return_dict=False``, micro-batch edge cases, the captured-graph form of ``ee_forward`` (replay == eager bits), the fused exit tail."""
import importlib

import numpy as np
import pytest

from .conftest import DIT_EE, TINY_CASES, load_golden

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


def test_dit_wrapper_with_micro_batches(pkg):
    """``DiTEEForImageClassification(micro_batches=2)`` used its own ``_run`` and raised for ``output_hidden_states``;
    it now shares the base class's routing (hidden states: first handle, a slice at a time) and returns the one-handle bits."""
    import torch
    g = load_golden("dit_tiny")
    cfg = pkg.ModelConfig.dit_tiny(EE_config=DIT_EE)
    W = pkg.synth.make_weights_beit(cfg, seed=int(g["seed_w"]))
    pix = torch.from_numpy(pkg.synth.make_documents(cfg, int(g["n_docs"]), seed=int(g["seed_docs"]), text_len=8)["pixel_values"])
    one = pkg.DiTEEForImageClassification(cfg, W, max_docs=4)
    two = pkg.DiTEEForImageClassification(cfg, W, max_docs=4, micro_batches=2)        # 6 documents: chunks of 4 + 2, each as two slices
    assert type(two.engine).__name__ == "MicroBatchedEngine"
    lab = torch.zeros(pix.shape[0], dtype=torch.int64)
    a, b = one(pixel_values=pix, labels=lab), two(pixel_values=pix, labels=lab)
    np.testing.assert_allclose(_np(b.logits), g["logits"], rtol=0, atol=1e-4)
    assert np.array_equal(_np(a.logits), _np(b.logits)) and float(a.loss) == float(b.loss)
    for j in range(4):
        assert np.array_equal(_np(a.exit_states[j][0]), _np(b.exit_states[j][0]))
    ha, hb = one(pixel_values=pix, output_hidden_states=True), two(pixel_values=pix, output_hidden_states=True)
    L = cfg.num_hidden_layers
    assert len(hb.hidden_states) == L + 1 and hb.attentions is None
    for x, y in zip(ha.hidden_states, hb.hidden_states):
        assert np.array_equal(_np(x), _np(y))
    for l in range(L + 1):
        np.testing.assert_allclose(_np(hb.hidden_states[l][:, 0]), g["hidden_cls"][l], rtol=0, atol=1e-4)
    ea, eb = one.early_exit(pixel_values=pix, thresholds=float(g["pol_thr1"])), two.early_exit(pixel_values=pix, thresholds=float(g["pol_thr1"]))
    assert np.array_equal(_np(eb.exit_layer), g["pol_exits1"]) and np.array_equal(_np(ea.logits), _np(eb.logits))
    with pytest.raises(NotImplementedError):
        two(pixel_values=pix, output_attentions=True)
    one.engine.close()
    two.engine.close()


def test_return_dict_false_follows_the_reference_tuple(pkg):
    """EE/models/LayoutLMv3.py:883-885: ``(logits,) + outputs[1:]`` with the loss in front when labels were passed, ``outputs`` being the
    backbone's tuple ``(sequence_output, [all_hidden_states], [all_attentions])`` (:287-296, 654-655)."""
    import torch
    name = "tiny_entropy_1layer_head"
    g = load_golden(name)
    cfg = pkg.ModelConfig.tiny(EE_config=TINY_CASES[name])
    m = pkg.LayoutLMv3EEForSequenceClassification(cfg, pkg.synth.make_weights(cfg, seed=int(g["seed_w"])), max_docs=8,
                                                  max_text_len=int(g["text_len"]))
    b = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("in_")}
    lab = b.pop("labels")
    d = m.forward(**b, labels=lab)
    t = m.forward(**b, labels=lab, return_dict=False)
    assert isinstance(t, tuple) and len(t) == 2 and float(t[0]) == float(d.loss) and np.array_equal(_np(t[1]), _np(d.logits))
    t = m.forward(**b, return_dict=False)
    assert len(t) == 1 and np.array_equal(_np(t[0]), _np(d.logits))
    t = m.forward(**b, return_dict=False, output_hidden_states=True, output_attentions=True)
    dh = m.forward(**b, output_hidden_states=True, output_attentions=True)      # (hidden states run on dense rows: equal to the ragged call to rounding only)
    L = cfg.num_hidden_layers
    assert len(t) == 3 and np.array_equal(_np(t[0]), _np(dh.logits)) and len(t[1]) == L + 1 and len(t[2]) == L
    np.testing.assert_allclose(_np(t[0]), _np(d.logits), rtol=0, atol=2e-5)
    assert all(np.array_equal(_np(x), _np(y)) for x, y in zip(t[1], dh.hidden_states))
    t = m.forward(**b, return_dict=False, output_attentions=True)
    assert len(t) == 2 and len(t[1]) == L and tuple(t[1][0].shape)[1] == cfg.num_attention_heads
    # the criterion override reaches exit_criterion() and the handle alike (they used to disagree after an override)
    m.config.exit_config["inference_strategy"] = "max_confidence"
    c = m.exit_criterion(d.logits)
    np.testing.assert_allclose(_np(c), _np(torch.softmax(d.logits, 1).max(1)[0]), rtol=0, atol=0)
    assert str(m.model_config.exit_config.inference_strategy) == "max_confidence" == str(m.engine.exit_config.inference_strategy)
    m.engine.close()


def test_micro_batch_edges(pkg):
    """``split_sizes(0)`` divided by zero; ``check()`` only visited the handles of the LAST call, so the device error of a
    slice that a later, smaller call did not use was never reported; ``validate=True`` serialised the slices."""
    import torch
    ee = dict(exits=[1, 2], encoder_layer_strategy="ramp")
    cfg = pkg.ModelConfig.tiny(EE_config=ee, hidden_size=256, num_attention_heads=4, intermediate_size=512, num_hidden_layers=3,
                               coordinate_size=48, shape_size=32)
    mb = pkg.MicroBatchedEngine(cfg, max_docs=8, max_text_len=32, micro_batches=2)
    mb.load_weights(pkg.synth.make_weights(cfg, seed=3, head_gain=6.0))
    with pytest.raises(ValueError):
        mb.split_sizes(0)
    docs = pkg.synth.make_documents(cfg, 8, seed=5, text_len=32, min_words=3)
    t = {k: torch.from_numpy(v).cuda() for k, v in docs.items() if k != "labels"}
    good = mb.forward(**t, thresholds=2.0, validate=True)           # validated after the join: both slices ran, nothing raised
    bad = dict(t)
    bad["input_ids"] = t["input_ids"].clone()
    bad["input_ids"][6, 1] = cfg.vocab_size + 5                     # a document of the SECOND slice
    mb.forward(**bad, thresholds=2.0)                               # only enqueues
    one = {k: v[:1] for k, v in t.items()}
    mb.forward(**one, thresholds=2.0)                               # one slice: the second handle is not used by this call
    with pytest.raises(pkg.capi.MMEEError):
        mb.check()                                                  # ... and its pending error is reported all the same
    again = mb.forward(**t, thresholds=2.0, validate=True)
    assert np.array_equal(_np(again.logits), _np(good.logits))
    with pytest.raises(pkg.capi.MMEEError):
        mb.forward(**bad, thresholds=2.0, validate=True)
    mb.close()


CONFIG2_EE = dict(exits=[2, 4, 6, 8, 10], encoder_layer_strategy="ramp")


@pytest.mark.parametrize("B", [1, 5])
def test_captured_graph_replays_the_eager_bits(pkg, B):
    """``ee_graph_capture`` / ``ee_graph_launch``.  One capture per (B, T, flags, outputs); replays over THREE different
    batches, with different threshold vectors (and temperatures) per launch, return the eager call's bits -- logits, exit indices,
    confidences, every evaluated exit's logits -- and the same stage populations.  LayoutLMv3-base, config 2's exit set, T = 512, B = 1 (the
    reference's eval_batch_size, EE/configs.py:36) and 5."""
    import torch
    cfg = pkg.ModelConfig.base(EE_config=CONFIG2_EE)
    eng = pkg.EarlyExitEngine(cfg, max_docs=B, max_text_len=512)
    eng.load_weights(pkg.synth.make_weights(cfg, seed=1234, head_gain=6.0))
    E = eng.E
    batches = [pkg.synth.make_documents(cfg, B, seed=900 + i, text_len=512) for i in range(3)]
    dev = lambda d: {k: torch.from_numpy(d[k]).cuda() for k in ("input_ids", "attention_mask", "bbox", "pixel_values")}
    thr_sets = [np.array([0.35, 0.4, 0.45, 0.5, 0.55, 2.0]), np.array([2.0, 2.0, 0.3, 0.3, 0.3, 2.0]), np.full(E + 1, 0.25)]
    temps = [None, np.array([1.5, 0.7, 1.0, 2.0, 1.1, 0.9]), None]
    first = dev(batches[0])
    # (device tensors are borrowed as they are: the graph's static inputs are the tensors handed to capture(), hence the clones)
    cap = eng.capture(**{k: v.clone() for k, v in first.items()}, thresholds=thr_sets[0], want_all=True)
    assert isinstance(cap, pkg.CapturedForward) and cap.graph_id >= 0
    seen = set()
    for rnd in range(2):                                   # every batch twice: a replay leaves nothing behind that the next one reads
        for i, b in enumerate(batches):
            t = dev(b)
            eager = eng.forward(**t, thresholds=thr_sets[i], temperatures=temps[i], want_all=True)
            sc_e = eng.stage_counts()
            lp_e = eng.layer_plan()
            ref = [_np(x).copy() for x in (eager.logits, eager.exit_layer, eager.confidence, eager.all_logits, eager.all_crit)]
            for k, v in t.items():
                cap.inputs[k].copy_(v)
            cap.outputs.all_logits.fill_(float("nan"))
            cap.outputs.all_crit.fill_(float("nan"))
            out = cap.launch(thresholds=thr_sets[i], temperatures=temps[i], validate=True)
            got = [_np(x) for x in (out.logits, out.exit_layer, out.confidence, out.all_logits, out.all_crit)]
            for a, g in zip(ref, got):
                assert np.array_equal(a, g, equal_nan=True)
            assert eng.stage_counts() == sc_e and eng.layer_plan() == lp_e
            seen.update(got[1].tolist())
    assert B == 1 or len(seen) >= 2                        # the threshold vectors really were this launch's (different exits across launches)
    # an out-of-range token id in a replay is reported like an eager call's
    cap.inputs["input_ids"][0, 1] = cfg.vocab_size + 3
    with pytest.raises(pkg.capi.MMEEError):
        cap.launch(thresholds=thr_sets[0], validate=True)
    cap.inputs["input_ids"].copy_(first["input_ids"])
    out = cap.launch(thresholds=thr_sets[0])
    assert np.array_equal(_np(out.exit_layer), _np(eng.forward(**first, thresholds=thr_sets[0]).exit_layer))
    # dump-all capture (what model.forward runs): no thresholds needed at launch
    cap2 = eng.capture(**{k: v.clone() for k, v in first.items()}, dump_all=True, want_all=True, want_head=True)
    ref = eng.forward(**first, dump_all=True, want_all=True, want_head=True)
    out2 = cap2.launch()
    assert np.array_equal(_np(out2.all_logits), _np(ref.all_logits)) and np.array_equal(_np(out2.head_crit), _np(ref.head_crit))
    cap.close()
    cap2.close()
    eng.close()


def test_result_rows_of_the_c_abi_equal_the_python_hosts(pkg):
    """VERDICT r05 "other": a non-Python host gets the row of the one all-gather from the C-ABI (ee_pack_results / ee_unpack_results): the same
    int32 words as dist.pack_results, denormal-valued exit indices and NaN payloads included."""
    import ctypes as C
    import torch
    lib = pkg.capi.load()
    n, K = 1000, 16
    lg = torch.randn(n, K, device="cuda")
    lg[3, 2] = float("nan")
    lg[4, 0] = 1e-42                                   # a float32 denormal
    ex = (torch.arange(n, device="cuda", dtype=torch.int32) % 7) - 1       # small and negative integers: denormals / NaN payloads as float bits
    cf = torch.rand(n, device="cuda")
    ref = pkg.dist.pack_results(lg, ex, cf)
    rows = torch.empty((n, K + 2), dtype=torch.int32, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    pkg.capi.check(lib.ee_pack_results(p(lg), p(ex), p(cf), n, K, p(rows), st), None, "ee_pack_results")
    assert torch.equal(rows, ref)
    lg2, ex2, cf2 = torch.empty_like(lg), torch.empty_like(ex), torch.empty_like(cf)
    pkg.capi.check(lib.ee_unpack_results(p(rows), n, K, p(lg2), p(ex2), p(cf2), st), None, "ee_unpack_results")
    assert torch.equal(lg2.view(torch.int32), lg.view(torch.int32)) and torch.equal(ex2, ex) and torch.equal(cf2, cf)
    a, b, c = pkg.dist.unpack_results(rows)
    assert torch.equal(a.view(torch.int32), lg.view(torch.int32)) and torch.equal(b, ex) and torch.equal(c, cf)


def test_holes_in_the_attention_mask_stay_rows(pkg, oracle):
    """Round 6: the kept text rows of a document are a PREFIX of its tokens (csrc/prep_embed.hip): a masked position before the last kept one --
    which the tokenizer never produces but the reference's signature accepts -- stays a row and is masked as a key, exactly as under dense rows.
    Both layouts, both schedules, against the oracle (which masks keys as the reference does, EE/models/LayoutLMv3.py:622-624)."""
    import torch
    from .conftest import H256_KW
    ee = dict(exits=["text_visual_concat", 1, 2], encoder_layer_strategy="ramp")
    cfg = pkg.ModelConfig.tiny(EE_config=ee, **H256_KW)
    W = pkg.synth.make_weights(cfg, seed=21, head_gain=6.0)
    docs = pkg.synth.make_documents(cfg, 6, seed=22, text_len=64, min_words=20)
    rng = np.random.default_rng(5)
    am = docs["attention_mask"]
    for b in range(6):                                   # punch holes into the valid range (never position 0), one document masked almost entirely
        n = int(am[b].sum())
        holes = rng.choice(np.arange(1, n - 1), size=(n - 3) if b == 5 else max(1, n // 4), replace=False)
        am[b, holes] = 0
    ref = oracle.forward_all(cfg, W, docs, ee["exits"], strategy="ramp")
    eng = pkg.EarlyExitEngine(cfg, max_docs=6, max_text_len=64)
    eng.load_weights(W)
    t = {k: torch.from_numpy(docs[k]).cuda() for k in ("input_ids", "attention_mask", "bbox", "pixel_values")}
    for kw in (dict(), dict(dense_rows=True), dict(whole_layers=True), dict(xprobe=False)):
        out = eng.forward(**t, dump_all=True, want_all=True, validate=True, **kw)
        np.testing.assert_allclose(_np(out.all_logits).astype(np.float64), ref["logits_store"], rtol=0, atol=1e-4)
    rows = eng.stage_counts()["rows"][0]
    last = [int(np.nonzero(am[b])[0].max()) + 1 for b in range(6)]
    assert rows == sum(last) + 6 * 17                     # prefix up to the last kept token + (64 / 16)^2 + 1 visual rows per document
    conf = np.sort(_np(out.all_crit)[1])
    thr = [2.0, float(0.5 * (conf[2] + conf[3])), 2.0, 2.0]
    a, b = eng.forward(**t, thresholds=thr, xprobe=False), eng.forward(**t, thresholds=thr, xprobe=False, dense_rows=True)
    assert np.array_equal(_np(a.exit_layer), _np(b.exit_layer)) and len(np.unique(_np(a.exit_layer))) == 2
    eng.close()


def test_idx16_attention_form_is_bit_identical_to_the_word_index(pkg):
    """Round 6 experiment kept in the DIAGNOSTIC library (MMEE_ATTN_IDX=16): the 16-bit pair index + delta table form of the attention kernel
    returns the bits of the shipped 32-bit form (tools/attn_xp_check.py: 24 base-shape documents through every layer)."""
    import os
    import subprocess
    import sys
    from .conftest import ROOT
    diag = os.path.join(ROOT, "multi-modal-early-exit_amd", "libmmee_hip_diag.so")
    if not os.path.exists(diag):
        pytest.skip("diagnostic library not built (make -C multi-modal-early-exit_amd/csrc diag)")
    tool = os.path.join(ROOT, "tools", "attn_xp_check.py")
    ref = os.path.join("/tmp", f"mmee_idx32_{os.getpid()}.pt")
    env = dict(os.environ, MMEE_LIB=diag)
    r = subprocess.run([sys.executable, tool, "save", ref], env=dict(env, MMEE_ATTN_IDX="32"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    r = subprocess.run([sys.executable, tool, "cmp", ref], env=dict(env, MMEE_ATTN_IDX="16"), capture_output=True, text=True, timeout=600)
    os.remove(ref)
    assert r.returncode == 0 and "0.000e+00" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_captured_graph_of_the_image_only_model(pkg):
    """``capture()`` on a BEiT / DiT handle (pixel_values is the whole input): replays equal the eager call, thresholds per launch."""
    import torch
    g = load_golden("dit_tiny")
    cfg = pkg.ModelConfig.dit_tiny(EE_config=DIT_EE)
    eng = pkg.EarlyExitEngine(cfg, max_docs=8, max_text_len=0)
    eng.load_weights(pkg.synth.make_weights_beit(cfg, seed=int(g["seed_w"])))
    pix = torch.from_numpy(pkg.synth.make_documents(cfg, 6, seed=int(g["seed_docs"]), text_len=8)["pixel_values"]).cuda()
    cap = eng.capture(pixel_values=pix.clone(), thresholds=2.0)
    for thr in (float(g["pol_thr1"]), 2.0, 0.0):
        e = eng.forward(pixel_values=pix, thresholds=thr)
        o = cap.launch(thresholds=thr, validate=True)
        assert np.array_equal(_np(o.exit_layer), _np(e.exit_layer)) and np.array_equal(_np(o.logits), _np(e.logits))
    assert np.array_equal(_np(cap.launch(thresholds=float(g["pol_thr1"])).exit_layer), g["pol_exits1"])
    cap.close()
    eng.close()


def test_model_surface_runs_small_batches_whole(pkg):
    """Round 6: ``model.early_exit`` on a batch of at most 16 documents (the reference's eval_batch_size = 1 included) runs whole layers -- a probe is
    13 launches and never pays there (bench.py ``small_batch``) -- unless the caller chose a schedule; larger batches keep the default (every decision
    layer probed first).  Same exits, logits within the bar, and the small-batch form is the one whose rows are bit-identical to the dump-all rows."""
    import torch
    from .conftest import H256_KW
    ee = dict(exits=[1, 2], encoder_layer_strategy="ramp")
    cfg = pkg.ModelConfig.tiny(EE_config=ee, **H256_KW)
    m = pkg.LayoutLMv3EEForSequenceClassification(cfg, weights=pkg.synth.make_weights(cfg, seed=9, head_gain=6.0), max_docs=24, max_text_len=48)
    docs = pkg.synth.make_documents(cfg, 24, seed=10, text_len=48, min_words=3)
    t = {k: torch.from_numpy(v).cuda() for k, v in docs.items() if k != "labels"}
    dump = m.forward(**t)
    conf = np.sort(_np(dump.exit_states[0][1]))
    thr = [float(0.5 * (conf[11] + conf[12])), 2.0, 2.0]
    big = m.early_exit(**t, thresholds=thr)
    plan_big = m.engine.layer_plan()
    assert sum(plan_big["docs_probe"]) > 0                                   # 24 documents: the default schedule
    small_t = {k: v[:8] for k, v in t.items()}
    small = m.early_exit(**small_t, thresholds=thr)
    assert sum(m.engine.layer_plan()["docs_probe"]) == 0                     # 8 documents: whole layers
    assert np.array_equal(_np(small.exit_layer), _np(big.exit_layer)[:8])
    np.testing.assert_allclose(_np(small.logits), _np(big.logits)[:8], rtol=0, atol=1e-4)
    forced = m.early_exit(**small_t, thresholds=thr, probe_always=True)      # the caller's choice wins
    assert sum(m.engine.layer_plan()["docs_probe"]) > 0 and np.array_equal(_np(forced.exit_layer), _np(small.exit_layer))
    m.engine.pin_schedule([0])                                               # ... and so does a pinned schedule
    m.early_exit(**small_t, thresholds=thr)
    assert m.engine.layer_plan()["docs_probe"][0] > 0
    m.engine.pin_schedule(False)
    m.early_exit(**small_t, thresholds=thr)
    assert sum(m.engine.layer_plan()["docs_probe"]) == 0
    m.engine.close()
