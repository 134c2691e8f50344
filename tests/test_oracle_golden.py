"""The oracle (oracle/ee_oracle.py) against the committed golden vectors minted from the composed reference
(tests/golden/make_golden.py: HF encoder + the reference's own exit-head / criterion / Policy classes)."""
import numpy as np
import pytest

from .conftest import BASE_EE, MATRIX_CASES, MATRIX_SEEDS, TINY_CASES, load_golden, matrix_config


def test_bucket_lut_matches_hf(oracle):
    g = load_golden("bucket_lut")
    d = g["delta"]
    assert np.array_equal(oracle.relative_position_bucket(d, 32, 128), g["lut_1d_32_128"])
    assert np.array_equal(oracle.relative_position_bucket(d, 64, 256), g["lut_2d_64_256"])
    assert np.array_equal(oracle.bucket_lut(1023, 32, 128), g["lut_1d_32_128"])


@pytest.mark.parametrize("name", list(TINY_CASES))
def test_tiny_forward_matches_reference(pkg, oracle, name):
    g = load_golden(name)
    ee = TINY_CASES[name]
    cfg = pkg.ModelConfig.tiny(EE_config=ee)
    W = pkg.synth.make_weights(cfg, seed=int(g["seed_w"]))
    docs = {k[3:]: v for k, v in g.items() if k.startswith("in_")}
    ec = cfg.exit_config
    out = oracle.forward_all(cfg, W, docs, ec.exits, strategy=str(ec.encoder_layer_strategy),
                             criterion=str(ec.inference_strategy), return_hidden_cls=True)
    tol = dict(rtol=0, atol=2e-5)
    np.testing.assert_allclose(out["hidden_cls"], g["hidden_cls"], **tol)
    np.testing.assert_allclose(out["exit_logits"], g["exit_logits"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(out["logits"], g["logits"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(out["exit_crit"], g["exit_crit"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(out["final_crit"], g["final_crit"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(out["logits_store"], g["logits_store"], rtol=0, atol=1e-4)
    if "gated_logits" in g:
        np.testing.assert_allclose(out["gated_logits"], g["gated_logits"], rtol=0, atol=1e-4)


@pytest.mark.parametrize("name", [n for n in MATRIX_CASES if n.startswith("h256")])
def test_matrix_h256_forward_matches_reference(pkg, oracle, name):
    """Entropy criterion, one-layer heads, gate strategy, every embedding-level exit at H = 256 (the smallest split-precision shape)."""
    g = load_golden(name)
    cfg, ee, n_docs, T = matrix_config(pkg, name)
    W = pkg.synth.make_weights(cfg, seed=MATRIX_SEEDS["seed_w"])
    docs = pkg.synth.make_documents(cfg, n_docs, seed=MATRIX_SEEDS["seed_docs"], text_len=T, min_words=3)
    ec = cfg.exit_config
    out = oracle.forward_all(cfg, W, docs, ec.exits, strategy=str(ec.encoder_layer_strategy), criterion=str(ec.inference_strategy),
                             return_hidden_cls=True)
    scale = max(1.0, float(np.abs(g["logits_store"]).max()))
    np.testing.assert_allclose(out["hidden_cls"], g["hidden_cls"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(out["exit_logits"], g["exit_logits"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(out["logits_store"], g["logits_store"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(out["exit_crit"], g["exit_crit"], rtol=0, atol=2e-5 * scale)
    if "gated_logits" in g:
        np.testing.assert_allclose(out["gated_logits"], g["gated_logits"], rtol=0, atol=1e-4)
    for i in range(4):
        ex, pred, _ = oracle.policy_scan(g["logits_store"], float(g[f"pol_thr{i}"]))
        assert np.array_equal(ex, g[f"pol_exits{i}"])


@pytest.mark.parametrize("name", list(TINY_CASES) + ["h256_gate"])
def test_c_openmp_restatement_matches_reference(pkg, oracle, name):
    """oracle/ee_oracle_c.c (the C / OpenMP restatement timed by bench.py's cpu_baseline beside torch-CPU) against the same golden vectors."""
    import importlib
    oc = importlib.import_module("oracle.ee_oracle_c")
    if not oc.available():
        pytest.skip("oracle/libee_oracle_c.so not built (make -C oracle)")
    g = load_golden(name)
    if name in TINY_CASES:
        cfg = pkg.ModelConfig.tiny(EE_config=TINY_CASES[name])
        W = pkg.synth.make_weights(cfg, seed=int(g["seed_w"]))
        docs = {k[3:]: v for k, v in g.items() if k.startswith("in_")}
    else:
        cfg, ee, n_docs, T = matrix_config(pkg, name)
        W = pkg.synth.make_weights(cfg, seed=MATRIX_SEEDS["seed_w"])
        docs = pkg.synth.make_documents(cfg, n_docs, seed=MATRIX_SEEDS["seed_docs"], text_len=T, min_words=3)
    ec = cfg.exit_config
    out = oc.COracle(cfg, W).forward_all(docs, ec.exits, strategy=str(ec.encoder_layer_strategy), return_hidden_cls=True)
    np.testing.assert_allclose(out["hidden_cls"], g["hidden_cls"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(out["logits_store"], g["logits_store"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(out["logits"], g["logits"], rtol=0, atol=1e-4)


def test_tiny_policy_matches_reference(oracle):
    g = load_golden("tiny_ramp")
    for i in range(4):
        ex, pred, conf = oracle.policy_scan(g["logits_store"], float(g[f"pol_thr{i}"]))
        assert np.array_equal(ex, g[f"pol_exits{i}"]) and ex.dtype == np.int32
        np.testing.assert_array_equal(pred, g[f"pol_pred{i}"])
        d = oracle.exit_distribution(ex, g["logits_store"].shape[0])
        np.testing.assert_allclose([d[k] for k in sorted(d)], g[f"pol_dist{i}"])
        ex2, pred2 = oracle.policy_loop(g["logits_store"], float(g[f"pol_thr{i}"]))
        assert np.array_equal(ex2, ex) and np.array_equal(pred2, pred)


def test_policy_random_matches_reference(oracle):
    g = load_golden("policy_random")
    for i in range(5):
        ex, pred, _ = oracle.policy_scan(g["logits_store"], float(g[f"pol_thr{i}"]))
        assert np.array_equal(ex, g[f"pol_exits{i}"])
        np.testing.assert_array_equal(pred, g[f"pol_pred{i}"])
    thr = oracle.heuristic_thresholds(g["heur_accuracy"], g["heur_ece"], float(g["heur_eps"]))
    ex, pred, _ = oracle.policy_scan(g["logits_store"], thr)
    assert np.array_equal(ex, g["heur_exits"])
    np.testing.assert_array_equal(pred, g["heur_pred"])


def test_base_shape_matches_reference(pkg, oracle):
    """S = 709 (512 text + 197 visual), LayoutLMv3-base, 2 documents: CLS rows of every layer and all exit logits."""
    from .conftest import load_golden as lg
    import hashlib
    g = lg("base_cls")
    cfg = pkg.ModelConfig.base(EE_config=BASE_EE)
    W = pkg.synth.make_weights(cfg, seed=int(g["seed_w"]))
    docs = pkg.synth.make_documents(cfg, int(g["n_docs"]), seed=int(g["seed_docs"]), text_len=int(g["text_len"]))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
    assert sha(docs["pixel_values"]) == str(g["sha_pixel_values"]) and sha(docs["bbox"]) == str(g["sha_bbox"])
    assert sha(W["layoutlmv3.embeddings.word_embeddings.weight"]) == str(g["sha_word_emb"])
    out = oracle.forward_all(cfg, W, docs, BASE_EE["exits"], return_hidden_cls=True)
    np.testing.assert_allclose(out["hidden_cls"], g["hidden_cls"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(out["logits_store"], g["logits_store"], rtol=0, atol=1e-4)
    for i in range(4):
        ex, pred, _ = oracle.policy_scan(out["logits_store"], float(g[f"pol_thr{i}"]))
        assert np.array_equal(ex, g[f"pol_exits{i}"])


def test_pil_resize_restatement_matches_pillow_golden(pkg, oracle):
    g = load_golden("preprocess")
    for i, (h, w, c) in enumerate(g["shapes"]):
        img = pkg.synth.make_page_image(100 + i, int(h), int(w), int(c))
        rgb = img if img.ndim == 3 else np.repeat(img[:, :, None], 3, axis=2)
        assert np.array_equal(oracle.pil_bilinear_resize_u8(rgb, 224, 224), g[f"res{i}"]), (h, w, c)
    lut = oracle.rescale_normalize_lut()
    assert lut[0] == -1.0 and lut[255] == 1.0 and lut.dtype == np.float32


def test_torch_oracle_matches_numpy_oracle(pkg, oracle):
    import importlib
    ot = importlib.import_module("oracle.ee_oracle_torch")
    for name in ("tiny_ramp", "tiny_gate"):
        g = load_golden(name)
        ee = TINY_CASES[name]
        cfg = pkg.ModelConfig.tiny(EE_config=ee)
        W = pkg.synth.make_weights(cfg, seed=int(g["seed_w"]))
        docs = {k[3:]: v for k, v in g.items() if k.startswith("in_")}
        out = ot.TorchOracle(cfg, W).forward_all(docs, ee["exits"], strategy=ee["encoder_layer_strategy"])
        np.testing.assert_allclose(out["logits_store"], g["logits_store"], rtol=0, atol=1e-4)


def test_dit_oracle_matches_hf_beit_golden(pkg, oracle):
    """BASELINE configs[4]: the BEiT restatement against stock HF BeitForImageClassification + the reference's exit-head class."""
    from .conftest import DIT_BASE_EE, DIT_EE
    for name, mk, ee, tol in (("dit_tiny", pkg.ModelConfig.dit_tiny, DIT_EE, 2e-5), ("dit_base_cls", pkg.ModelConfig.dit_base, DIT_BASE_EE, 5e-5)):
        g = load_golden(name)
        cfg = mk(EE_config=ee)
        W = pkg.synth.make_weights_beit(cfg, seed=int(g["seed_w"]))
        pix = pkg.synth.make_documents(cfg, int(g["n_docs"]), seed=int(g["seed_docs"]), text_len=8)["pixel_values"]
        out = oracle.forward_all_beit(cfg, W, pix, ee["exits"], return_hidden_cls=True)
        np.testing.assert_allclose(out["hidden_cls"], g["hidden_cls"], rtol=0, atol=tol)
        np.testing.assert_allclose(out["logits_store"], g["logits_store"], rtol=0, atol=1e-4)
        np.testing.assert_allclose(out["exit_crit"], g["exit_crit"], rtol=0, atol=1e-5)


def test_inputs_embeds_restatement(pkg, oracle):
    """HF:148-158, 171-186: word rows handed in as inputs_embeds with input_ids reproduce the input_ids result exactly (position ids come
    from input_ids); without input_ids the default position ids are pad_token_id + 1 + t."""
    ee = dict(exits=["text_avg", 1, 2], encoder_layer_strategy="ramp")
    cfg = pkg.ModelConfig.tiny(EE_config=ee)
    W = pkg.synth.make_weights(cfg, seed=3)
    docs = pkg.synth.make_documents(cfg, 3, seed=4, text_len=24, min_words=3)
    word = W["layoutlmv3.embeddings.word_embeddings.weight"]
    a = oracle.forward_all(cfg, W, docs, ee["exits"])["logits_store"]
    b = oracle.forward_all(cfg, W, dict(docs, inputs_embeds=word[docs["input_ids"]]), ee["exits"])["logits_store"]
    assert np.array_equal(a, b)
    no_ids = {k: v for k, v in docs.items() if k != "input_ids"}
    T = docs["input_ids"].shape[1]
    seq = np.broadcast_to(np.arange(cfg.pad_token_id + 1, T + cfg.pad_token_id + 1, dtype=np.int64)[None], docs["input_ids"].shape)
    c = oracle.forward_all(cfg, W, dict(no_ids, inputs_embeds=word[docs["input_ids"]]), ee["exits"])["logits_store"]
    d = oracle.forward_all(cfg, W, dict(docs, position_ids=seq), ee["exits"])["logits_store"]
    assert np.array_equal(c, d)
    assert np.abs(a - c).max() > 0          # padded documents: the pad-aware ids differ from the sequential ones


def test_hidden_states_restatement_matches_the_reference_rows(pkg, oracle):
    """``return_hidden_states`` (EE/models/LayoutLMv3.py:164, 182-183, 284-285): the state entering every layer and the last layer's output,
    pinned on the rows of the composed reference's own hidden states that the tiny golden holds."""
    g = load_golden("tiny_ramp")
    ee = TINY_CASES["tiny_ramp"]
    cfg = pkg.ModelConfig.tiny(EE_config=ee)
    W = pkg.synth.make_weights(cfg, seed=int(g["seed_w"]))
    batch = {k[3:]: g[k] for k in g if k.startswith("in_") and k != "in_labels"}
    out = oracle.forward_all(cfg, W, batch, ee["exits"], return_hidden_cls=True, return_hidden_states=True)
    hs = out["hidden_states"]
    assert hs.shape[0] == cfg.num_hidden_layers + 1 and np.array_equal(hs[:, :, 0], out["hidden_cls"])
    np.testing.assert_allclose(hs[:, :, 0], g["hidden_cls"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(hs[0][:, 1], g["emb_out_row1"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(hs[0][:, -1], g["emb_out_lastrow"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(hs[1][:, 1], g["layer1_row1"], rtol=0, atol=2e-5)


# ---- round 5: N3 / N4 and the LayoutLMv3-large shape pinned to the reference's own code (tests/golden/make_golden.py sweep / temperature / large)
def test_sweep_restatement_matches_reference_vectors(oracle):
    """EE/large_scale.py:42-96 run in the build container (generate_thresholds with its own np.random.seed(42), check_2D_threshold, the msp CSF)
    on seeded logits: the oracle's restatement reproduces every exit histogram, accuracy and mean exit bit for bit, including the rows where
    no exit fires and numpy's argmax returns exit 0 (EE/large_scale.py:50)."""
    import hashlib
    from .conftest import load_golden as lg, sweep_ref_inputs
    g = lg("sweep_ref")
    logits, refs = sweep_ref_inputs()
    assert hashlib.sha256(np.ascontiguousarray(logits).tobytes()).hexdigest()[:16] == str(g["sha_logits"])
    conf, corr = oracle.msp_table(logits, refs)
    np.testing.assert_allclose(conf, g["conf"], rtol=4e-16, atol=0)            # scipy's softmax against the oracle's: last-bit agreement
    assert np.array_equal(corr, g["correct"])
    acc, mex, hist = oracle.threshold_sweep(g["conf"], g["correct"], g["thresholds"])
    assert np.array_equal(hist, g["hist"])
    assert np.array_equal(acc, g["accuracy"]) and np.array_equal(mex, g["mean_exit"])
    V = int(g["n_generated"])
    assert (g["thresholds"][:V, -1] == 0).all()                                  # the reference leaves the last row at 0 (EE/large_scale.py:56)
    assert (g["exits_corner"][0] == 0).all() and g["hist"][V, 0] == logits.shape[1]      # nothing fires anywhere -> exit 0 for everybody
    ex64 = np.stack([(g["conf"] >= t[:, None]).argmax(0) for t in g["thresholds"][:64]])
    assert np.array_equal(ex64, g["exits_first64"])


def test_temperature_restatement_matches_reference_scaler(oracle):
    """EE/generic_scaling.py:37-111 driven as EE/eval.py:298-329 drives it (one scaler object over the exits): the oracle's L-BFGS-B restatement
    lands on the reference's temperatures (2e-4 relative: two optimisers' stopping rules, not two objectives)."""
    from .conftest import load_golden as lg, temperature_ref_inputs
    g = lg("temperature_ref")
    logits, refs = temperature_ref_inputs()
    for e in range(logits.shape[0]):
        t = oracle.fit_temperature(logits[e], refs)
        assert abs(t - g["temperature"][e]) <= 2e-4 * g["temperature"][e], (e, t, g["temperature"][e])
        np.testing.assert_allclose(oracle.nll_at_temperature(logits[e], refs, 1.0), g["nll_before"][e], rtol=1e-12)
        np.testing.assert_allclose(oracle.nll_at_temperature(logits[e], refs, float(g["temperature"][e])), g["nll_after"][e], rtol=1e-12)
    assert g["temperature"].min() < 0.5 and g["temperature"].max() > 2.0          # both sides of 1 are exercised


def test_large_shape_gate_matches_reference(pkg):
    """BASELINE configs[2] shape (LayoutLMv3-large: H = 1024, L = 24, 16 heads, I = 4096; gate exits after layers 1..23 + final, S = 709): the
    torch-CPU restatement the config-3 GPU tests lean on, against the composed reference (stock HF encoder + the reference's LayoutLMv3Exit /
    classifier wiring of EE/models/LayoutLMv3.py:764-792) on 2 documents."""
    import hashlib
    import importlib
    from .conftest import LARGE_GATE_EE, load_golden as lg
    g = lg("large_gate")
    otorch = importlib.import_module("oracle.ee_oracle_torch")
    cfg = pkg.ModelConfig.large(EE_config=LARGE_GATE_EE)
    W = pkg.synth.make_weights(cfg, seed=int(g["seed_w"]))
    docs = pkg.synth.make_documents(cfg, int(g["n_docs"]), seed=int(g["seed_docs"]), text_len=int(g["text_len"]))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
    assert sha(docs["pixel_values"]) == str(g["sha_pixel_values"]) and sha(docs["input_ids"]) == str(g["sha_input_ids"])
    out = otorch.TorchOracle(cfg, W).forward_all(docs, LARGE_GATE_EE["exits"], strategy="gate")
    assert out["logits_store"].shape == (24, 2, 16)
    np.testing.assert_allclose(out["logits_store"], g["logits_store"], rtol=0, atol=1e-4)          # gated logits: classifier(CLS of layer l)
    np.testing.assert_allclose(out["exit_logits"], g["exit_logits"], rtol=0, atol=1e-4)            # the 2-way gate heads
    np.testing.assert_allclose(out["gate_inputs"], g["hidden_cls"][1:24], rtol=0, atol=5e-5)       # CLS rows entering the heads
    oracle = importlib.import_module("oracle.ee_oracle")
    for i in range(4):
        ex, pred, _ = oracle.policy_scan(out["logits_store"], float(g[f"pol_thr{i}"]))
        assert np.array_equal(ex, g[f"pol_exits{i}"])


def test_attention_probabilities_match_hf(pkg, oracle):
    """``output_attentions=True``: the oracle's per-layer attention probabilities against the stock HF encoder's (tests/golden/tiny_attentions.npz),
    masked keys at exactly 0; a head mask (restated from transformers 4.26: probabilities times one factor per layer and head) scales them."""
    from .conftest import load_golden as lg
    g = lg("tiny_attentions")
    ee = dict(exits=[1, 3], encoder_layer_strategy="ramp", inference_strategy="max_confidence")
    cfg = pkg.ModelConfig.tiny(EE_config=ee)
    W = pkg.synth.make_weights(cfg, seed=int(g["seed_w"]))
    docs = {k[3:]: v for k, v in g.items() if k.startswith("in_")}
    out = oracle.forward_all(cfg, W, docs, ee["exits"], return_attentions=True)
    assert out["attentions"].shape == g["attentions"].shape
    np.testing.assert_allclose(out["attentions"], g["attentions"], rtol=0, atol=2e-6)
    pad = docs["attention_mask"][0] == 0
    assert pad.any() and (out["attentions"][:, 0, :, :, :48][..., pad] == 0).all()
    hm = np.array([[1.0, 0.0], [0.5, 1.0], [1.0, 1.0], [0.0, 2.0]], dtype=np.float32)
    o2 = oracle.forward_all(cfg, W, docs, ee["exits"], return_attentions=True, head_mask=hm)
    np.testing.assert_allclose(o2["attentions"][0], out["attentions"][0] * hm[0][None, :, None, None], rtol=0, atol=1e-7)      # layer 0 sees the same input
    assert (o2["attentions"][3][:, 0] == 0).all() and np.abs(o2["logits"] - out["logits"]).max() > 1e-3
