"""Golden-vector generator.  Runs ONLY in the build container (needs /root/reference and the installed
``transformers``); its outputs (``tests/golden/*.npz``) are committed and travel, this script's imports do not.

Composed reference, all of it code that already exists outside this repo:
  * encoder arithmetic: stock HF ``LayoutLMv3ForSequenceClassification`` (transformers 5.15.0, torch CPU fp32),
  * exit heads + criteria: the reference's own ``LayoutLMv3Exit`` / ``max_confidence`` / ``entropy`` / ``ExitConfig``
    (EE/models/LayoutLMv3.py:56-93, EE/models/EE_modules.py:149-195), imported from /root/reference,
  * policy: the reference's own ``Policy`` (EE/policy.py),
  * bucket LUTs: HF ``LayoutLMv3Encoder.relative_position_bucket``.
The reference's EE_modules imports ``fvcore`` (absent here) at module level without using it on this path; a dummy
module object is registered for that import only.

Round 5 (``sweep`` / ``temperature`` / ``large``): N3 and N4 pinned to the reference's own code, with these import-time-only stand-ins,
each an EMPTY module object registered for one ``import`` statement and never called:
  * ``seaborn``  -- EE/thresh.py:11 (plot helpers; ``check_2D_threshold`` / ``opt0_2D`` / ``opt1`` / ``CSF`` do not touch it),
  * ``evaluate`` -- EE/metrics.py (imported by EE/generic_scaling.py:5 for ``ece_logits``, which loads the REMOTE metric ``jordyvl/ece``; inside
    ``TemperatureScaler.set_temperature`` its two results only feed ``print`` lines, EE/generic_scaling.py:84-88, 104-110, so the name is
    rebound to a function returning NaN: the fitted temperature does not depend on it; ECE itself stays "parity unpinned").
EE/large_scale.py cannot be imported even so: its ``from utils import ...`` (line 7) executes EE/configs.py, whose module body needs a WORKING
``sacred.Experiment`` (decorators at import time) -- a functional stand-in, not a stub, so none is written.  ``generate_thresholds`` and
``check_2D_threshold`` (EE/large_scale.py:42-66) are pure numpy functions of that file: their ``def`` statements (and ``CSF_dict``) are
compiled FROM THE REFERENCE FILE, read where it lies, into a namespace that holds numpy + scipy's softmax, without executing the module's
imports -- the code that runs is the reference's text, nothing of it is stored here.

Weights and documents come from ``multi-modal-early-exit_amd.synth`` (numpy-seeded, rebuildable anywhere), so the
fixtures hold seeds + expected outputs (+ the small tiny-config inputs) instead of hundreds of MB of tensors.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz
"""
import hashlib
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/EE"
sys.path.insert(0, REF)

pkg = importlib.import_module("multi-modal-early-exit_amd")
from transformers import LayoutLMv3Config, LayoutLMv3ForSequenceClassification  # noqa: E402

_fv = types.ModuleType("fvcore"); _fvnn = types.ModuleType("fvcore.nn")
_fvnn.FlopCountAnalysis = object; _fvnn.parameter_count = object
sys.modules.setdefault("fvcore", _fv); sys.modules.setdefault("fvcore.nn", _fvnn)
from models.EE_modules import ExitConfig as RefExitConfig, max_confidence as ref_maxconf, entropy as ref_entropy  # noqa: E402
from models.LayoutLMv3 import LayoutLMv3Exit as RefExit  # noqa: E402
from policy import Policy as RefPolicy  # noqa: E402

torch.set_grad_enabled(False)
torch.manual_seed(0)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def hf_model(cfg, W, ee_cfg):
    hc = LayoutLMv3Config(
        vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
        num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
        max_position_embeddings=cfg.max_position_embeddings, type_vocab_size=cfg.type_vocab_size,
        coordinate_size=cfg.coordinate_size, shape_size=cfg.shape_size, input_size=cfg.input_size,
        patch_size=cfg.patch_size, num_labels=cfg.num_labels, max_2d_position_embeddings=cfg.max_2d_position_embeddings,
        rel_pos_bins=cfg.rel_pos_bins, max_rel_pos=cfg.max_rel_pos, rel_2d_pos_bins=cfg.rel_2d_pos_bins,
        max_rel_2d_pos=cfg.max_rel_2d_pos, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    m = LayoutLMv3ForSequenceClassification(hc).eval()
    sd = m.state_dict()
    own = {k: torch.from_numpy(v) for k, v in W.items() if k in sd}
    missing = [k for k in sd if k not in own and "position_ids" not in k and "visual_bbox" not in k]
    assert not missing, missing
    m.load_state_dict(own, strict=False)
    # the reference's heads
    hc.exit_config = RefExitConfig(**ee_cfg).__dict__
    heads = {}
    names = [k[:-len(".out_proj.weight")] for k in W if k.endswith(".out_proj.weight") and not k.startswith("classifier")]
    for n in names:
        h = RefExit(hc, cfg.hidden_size, n).eval()
        h.load_state_dict({k[len(n) + 1:]: torch.from_numpy(v) for k, v in W.items() if k.startswith(n + ".")})
        heads[n] = h
    return m, heads


def run_reference(cfg, W, docs, ee_cfg):
    """Composed reference forward; mirrors LayoutLMv3ModelEE.forward ordering (EE/models/LayoutLMv3.py:375-665)."""
    m, heads = hf_model(cfg, W, ee_cfg)
    ec = cfg.exit_config
    crit = ref_maxconf if str(ec.inference_strategy) == "max_confidence" else ref_entropy
    t = {k: torch.from_numpy(v) for k, v in docs.items() if k != "labels"}
    cap = {}
    h1 = m.layoutlmv3.embeddings.register_forward_hook(lambda mod, i, o: cap.__setitem__("text", o))
    layer_out = []
    hooks = [l.register_forward_hook(lambda mod, i, o: layer_out.append(o if torch.is_tensor(o) else o[0]))
             for l in m.layoutlmv3.encoder.layer]
    first_in = []
    h0 = m.layoutlmv3.encoder.layer[0].register_forward_pre_hook(lambda mod, a: first_in.append(a[0]))
    out = m(**t)
    h1.remove(); h0.remove(); [h.remove() for h in hooks]
    vis = m.layoutlmv3.forward_image(t["pixel_values"])
    emb_out = first_in[0]                    # LayerNorm(cat(text, visual)) — input of layer 0
    assert len(layer_out) == cfg.num_hidden_layers
    p = "layoutlmv3."
    nm = {"vision_avg": p + "vision_exit_embeddings", "text_avg": p + "text_exit_embeddings",
          "text_visual_concat": p + "concat_exit_embeddings"}
    ins, ex = [], []
    for e in ec.embedding_exits:
        x = {"vision_avg": vis.mean(1), "text_avg": cap["text"].mean(1), "text_visual_concat": emb_out.mean(1)}[e]
        ins.append(x); ex.append(heads[nm[e]](x))
    for k, l in enumerate(ec.encoder_exit_layers):
        x = layer_out[l - 1][:, 0, :]
        ins.append(x); ex.append(heads[f"{p}encoder.early_exits.{k}"](x))
    logits = out.logits
    res = {
        "exit_logits": torch.stack(ex).numpy() if ex else np.zeros((0,) + tuple(logits.shape), np.float32),
        "exit_crit": torch.stack([crit(z) for z in ex]).numpy() if ex else np.zeros((0, logits.shape[0]), np.float32),
        "logits": logits.numpy(), "final_crit": crit(logits).numpy(),
        "hidden_cls": torch.stack([emb_out[:, 0, :]] + [o[:, 0, :] for o in layer_out]).numpy(),
        "text_emb_mean": cap["text"].mean(1).numpy(), "vis_emb_mean": vis.mean(1).numpy(),
        "emb_out_mean": emb_out.mean(1).numpy(),
        "emb_out_row1": emb_out[:, 1, :].numpy(), "emb_out_lastrow": emb_out[:, -1, :].numpy(),
        "layer1_row1": layer_out[0][:, 1, :].numpy(),
    }
    E = len(ex)
    store = np.zeros((E + 1,) + tuple(logits.shape), np.float64)
    if str(ec.encoder_layer_strategy) == "gate":
        gl = torch.stack([m.classifier(x) for x in ins]).numpy() if ins else np.zeros((0,) + tuple(logits.shape))
        res["gated_logits"] = gl
        store[:E] = gl
    else:
        store[:E] = res["exit_logits"]
    store[-1] = res["logits"]
    res["logits_store"] = store
    return res


def policy_fixture(store, thresholds, prefix, res):
    for i, thr in enumerate(thresholds):
        pol = RefPolicy(store, {"exit_threshold": thr, "device": "cpu"})
        ex, pred, dist = pol.max_confidence_global_thresholding_policy()
        res[f"{prefix}_thr{i}"] = np.float64(thr)
        res[f"{prefix}_exits{i}"] = ex
        res[f"{prefix}_pred{i}"] = pred.numpy()
        res[f"{prefix}_dist{i}"] = np.array([dist[k] for k in range(store.shape[0])])


def main():
    ModelConfig, synth = pkg.ModelConfig, pkg.synth
    # ---- 1. bucket LUTs from HF's own function -----------------------------------------------------------------
    from transformers.models.layoutlmv3.modeling_layoutlmv3 import LayoutLMv3Encoder
    enc = LayoutLMv3Encoder.__new__(LayoutLMv3Encoder)
    d = torch.arange(-1023, 1024, dtype=torch.long)
    lut1 = LayoutLMv3Encoder.relative_position_bucket(enc, d, num_buckets=32, max_distance=128).numpy().astype(np.uint8)
    lut2 = LayoutLMv3Encoder.relative_position_bucket(enc, d, num_buckets=64, max_distance=256).numpy().astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, "bucket_lut.npz"), delta=d.numpy(), lut_1d_32_128=lut1, lut_2d_64_256=lut2)

    # ---- 2. tiny configs: every exit kind, ramp / gate / entropy -----------------------------------------------
    cases = {
        "tiny_ramp": dict(exits=["vision_avg", "text_avg", "text_visual_concat", 1, 2, 3, 4],
                          encoder_layer_strategy="ramp", inference_strategy="max_confidence"),
        "tiny_gate": dict(exits=["text_visual_concat", 2, 3], encoder_layer_strategy="gate",
                          inference_strategy="max_confidence"),
        "tiny_entropy_1layer_head": dict(exits=[1, 3], encoder_layer_strategy="ramp", inference_strategy="entropy",
                                         exit_head_num_layers=1),
    }
    for name, ee in cases.items():
        cfg = ModelConfig.tiny(EE_config=ee)
        W = synth.make_weights(cfg, seed=7)
        docs = synth.make_documents(cfg, 6, seed=11, text_len=48, min_words=3)
        docs["attention_mask"][5, :] = 1          # one document without padding
        docs["input_ids"][5, docs["input_ids"][5] == cfg.pad_token_id] = 5
        res = run_reference(cfg, W, docs, ee)
        if str(cfg.exit_config.inference_strategy) == "max_confidence":
            conf = np.sort(np.unique(np.round(
                np.exp(res["logits_store"] - res["logits_store"].max(-1, keepdims=True)).max(-1) /
                np.exp(res["logits_store"] - res["logits_store"].max(-1, keepdims=True)).sum(-1), 6)))
            mid = float(conf[len(conf) // 2]) + 1e-4
            policy_fixture(res["logits_store"], [0.0, mid, 0.9, 1.0 + 1e-6], "pol", res)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), seed_w=7, seed_docs=11, text_len=48, n_docs=6,
                            **{"in_" + k: v for k, v in docs.items()}, **res)
        print(name, "E =", res["exit_logits"].shape[0], "logits[0,:4] =", res["logits"][0, :4])

    # ---- 3. base shape (S = 512 + 197 = 709), CLS rows + exit logits only ---------------------------------------
    ee = dict(exits=["text_visual_concat", 2, 4, 6, 8, 10], encoder_layer_strategy="ramp")
    cfg = ModelConfig.base(EE_config=ee)
    W = synth.make_weights(cfg, seed=1234)
    docs = synth.make_documents(cfg, 2, seed=1234, text_len=512)
    res = run_reference(cfg, W, docs, ee)
    policy_fixture(res["logits_store"], [0.0, 0.5, 0.9, 1.0 + 1e-6], "pol", res)
    np.savez_compressed(os.path.join(HERE, "base_cls.npz"), seed_w=1234, seed_docs=1234, text_len=512, n_docs=2,
                        sha_pixel_values=sha(docs["pixel_values"]), sha_input_ids=sha(docs["input_ids"]),
                        sha_bbox=sha(docs["bbox"]), sha_word_emb=sha(W["layoutlmv3.embeddings.word_embeddings.weight"]),
                        **res)
    print("base_cls", res["logits_store"].shape, res["logits"][0, :4])

    # ---- 4. policy-only vectors at a larger N (reference Policy on random logits) --------------------------------
    rng = np.random.default_rng(5)
    store = rng.standard_normal((7, 512, 16)) * 3.0
    res = {"logits_store": store}
    policy_fixture(store, [0.0, 0.35, 0.6, 0.9, 1.0 + 1e-6], "pol", res)
    cm = {"accuracy": list(rng.uniform(0.3, 0.9, 7)), "ece": list(rng.uniform(0.02, 0.2, 7)),
          "average_confidence": list(rng.uniform(0.3, 0.9, 7))}
    pol = RefPolicy(store, {"exit_threshold": 0.5, "device": "cpu", "epsilon": 0.1, "calibration_metrics": cm})
    ex, pred, dist = pol.accuracy_calibration_heuristic()
    res.update(heur_accuracy=np.array(cm["accuracy"]), heur_ece=np.array(cm["ece"]),
               heur_avgconf=np.array(cm["average_confidence"]), heur_eps=0.1, heur_exits=ex, heur_pred=pred.numpy(),
               heur_dist=np.array([dist[k] for k in range(7)]))
    np.savez_compressed(os.path.join(HERE, "policy_random.npz"), **res)
    print("policy_random done")


PREPROCESS_SHAPES = [(1000, 762, 1), (100, 150, 1), (224, 224, 3), (300, 224, 1), (333, 1000, 3)]


def make_preprocess_golden():
    """Pillow's own Image.resize(BILINEAR) on synthetic pages (down-, up-scaling, identity, greyscale and RGB); the pages
    are rebuilt from their seeds by synth.make_page_image, the fixture holds Pillow's 224x224 outputs."""
    from PIL import Image
    out = {"shapes": np.array(PREPROCESS_SHAPES)}
    for i, (h, w, c) in enumerate(PREPROCESS_SHAPES):
        a = pkg.synth.make_page_image(100 + i, h, w, c)
        img = Image.fromarray(a).convert("RGB")
        out[f"res{i}"] = np.asarray(img.resize((224, 224), resample=Image.BILINEAR))
    np.savez_compressed(os.path.join(HERE, "preprocess.npz"), **out)
    print("preprocess golden done")


DIT_EE = dict(exits=[1, 2, 3, 4], encoder_layer_strategy="ramp", inference_strategy="max_confidence")
DIT_BASE_EE = dict(exits=[2, 4, 6, 8, 10], encoder_layer_strategy="ramp", inference_strategy="max_confidence")


def make_dit_golden():
    """BASELINE configs[4]: stock HF BeitForImageClassification (what the reference's "dit" branch loads) + the reference's
    LayoutLMv3Exit heads on the CLS row after each exit layer (this build's extrapolation — the reference has no DiT exits)."""
    from transformers import BeitConfig, BeitForImageClassification

    def run(cfg, ee, seed_w, n_docs, seed_docs):
        W = pkg.synth.make_weights_beit(cfg, seed=seed_w)
        bc = BeitConfig(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                        num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                        image_size=cfg.input_size, patch_size=cfg.patch_size, num_labels=cfg.num_labels,
                        layer_norm_eps=cfg.layer_norm_eps, use_absolute_position_embeddings=True, use_relative_position_bias=False,
                        use_shared_relative_position_bias=False, use_mean_pooling=True,
                        layer_scale_init_value=cfg.layer_scale_init_value, use_mask_token=False, hidden_dropout_prob=0.0,
                        attention_probs_dropout_prob=0.0, drop_path_rate=0.0)
        m = BeitForImageClassification(bc).eval()
        ren = {"attention.attention.query": "attention.q_proj", "attention.attention.key": "attention.k_proj",
               "attention.attention.value": "attention.v_proj", "attention.output.dense": "attention.o_proj",
               "intermediate.dense": "mlp.fc1", "output.dense": "mlp.fc2"}
        sd = {}
        for k, v in W.items():
            if "early_exits" in k:
                continue
            k2 = k.replace("beit.encoder.layer.", "beit.layers.")
            for a, b in ren.items():
                k2 = k2.replace(a, b)
            sd[k2] = torch.from_numpy(v)
        missing, unexpected = m.load_state_dict(sd, strict=False)
        assert not unexpected and all("mask_token" in x for x in missing), (missing, unexpected)
        bc.exit_config = RefExitConfig(**ee).__dict__
        bc.classifier_dropout = None
        heads = []
        for k in range(len(ee["exits"])):
            n = f"beit.encoder.early_exits.{k}"
            hd = RefExit(bc, cfg.hidden_size, n).eval()
            hd.load_state_dict({kk[len(n) + 1:]: torch.from_numpy(v) for kk, v in W.items() if kk.startswith(n + ".")})
            heads.append(hd)
        pix = pkg.synth.make_documents(cfg, n_docs, seed=seed_docs, text_len=8)["pixel_values"]
        layer_out = []
        hooks = [l.register_forward_hook(lambda mod, i, o: layer_out.append(o if torch.is_tensor(o) else o[0])) for l in m.beit.layers]
        emb = []
        h0 = m.beit.layers[0].register_forward_pre_hook(lambda mod, a: emb.append(a[0]))
        out = m(pixel_values=torch.from_numpy(pix))
        [h.remove() for h in hooks]; h0.remove()
        ex = [heads[k](layer_out[l - 1][:, 0, :]) for k, l in enumerate(sorted(ee["exits"]))]
        store = np.zeros((len(ex) + 1,) + tuple(out.logits.shape), np.float64)
        store[:-1] = torch.stack(ex).numpy()
        store[-1] = out.logits.numpy()
        return {"seed_w": seed_w, "seed_docs": seed_docs, "n_docs": n_docs,
                "hidden_cls": torch.stack([emb[0][:, 0, :]] + [o[:, 0, :] for o in layer_out]).numpy(),
                "exit_logits": torch.stack(ex).numpy(), "exit_crit": torch.stack([ref_maxconf(z) for z in ex]).numpy(),
                "logits": out.logits.numpy(), "logits_store": store}

    cfg = pkg.ModelConfig.dit_tiny(EE_config=DIT_EE)
    res = run(cfg, DIT_EE, 5, 6, 6)
    policy_fixture(res["logits_store"], [0.0, 0.5, 0.9, 1.0 + 1e-6], "pol", res)
    np.savez_compressed(os.path.join(HERE, "dit_tiny.npz"), **res)
    cfg = pkg.ModelConfig.dit_base(EE_config=DIT_BASE_EE)
    res = run(cfg, DIT_BASE_EE, 1234, 3, 77)
    np.savez_compressed(os.path.join(HERE, "dit_base_cls.npz"), **res)
    print("dit golden done", res["logits"][0, :4])


def _stub_module(name):
    import importlib.machinery
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    sys.modules.setdefault(name, m)
    return sys.modules[name]


def _reference_defs(path, names):
    """Compile the named top-level ``def`` / assignment statements of a reference source file (read in place) into a fresh namespace holding
    numpy, scipy's softmax and OrderedDict -- for pure functions of modules whose import-time machinery cannot run here."""
    import ast
    from collections import OrderedDict
    from scipy.special import softmax
    src = open(path).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if (isinstance(n, ast.FunctionDef) and n.name in names) or
            (isinstance(n, ast.Assign) and any(isinstance(t, ast.Name) and t.id in names for t in n.targets))]
    assert {getattr(n, "name", None) or n.targets[0].id for n in keep} == set(names), "reference file changed"
    ns = {"np": np, "softmax": softmax, "OrderedDict": OrderedDict}
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return ns


def make_sweep_golden():
    """N3 pinned to the reference: threshold vectors from ``generate_thresholds`` (EE/large_scale.py:53-75, its own
    ``np.random.seed(42)``, 10 percentiles per exit), exits from ``check_2D_threshold`` (EE/large_scale.py:49-50 and, identically,
    EE/thresh.py:184-185 -- both are run and must agree), the CSF table from ``CSF_dict["msp"]`` (EE/large_scale.py:12-18), accuracy / mean exit
    by the two expressions of ``evaluate_exit_logits`` (EE/large_scale.py:88-96).  Extra rows exercise the corner the reference's argmax
    has: a vector whose LAST threshold no confidence reaches leaves exit 0 for documents no exit fires for (EE/large_scale.py:50)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import sweep_ref_inputs
    _stub_module("seaborn")
    import thresh as ref_thresh                                   # EE/thresh.py (imports fine with the seaborn stand-in)
    ls = _reference_defs(os.path.join(REF, "large_scale.py"), ["CSF_dict", "entropy", "top12_margin_np", "check_2D_threshold", "generate_thresholds"])
    logits, refs = sweep_ref_inputs()
    E1, N, K = logits.shape
    V = 1500
    ls["num_per_exit"], ls["num_mixtures"] = 10, V              # module globals of the reference's __main__ (EE/large_scale.py:176-177: 10, 1 500 000)
    thr = np.asarray(ls["generate_thresholds"](logits, refs), dtype=np.float64)
    assert thr.shape == (V, E1) and (thr[:, -1] == 0).all()
    conf = np.apply_along_axis(ls["CSF_dict"]["msp"], -1, logits)   # EE/large_scale.py:86: (E1, N)
    # corner rows: nothing fires anywhere (-> exit 0 for everybody), nothing fires for most, everything fires at exit 0
    corner = np.array([[1.5] * E1, [0.999] * (E1 - 1) + [1.5], [0.0] * E1, list(np.linspace(0.9, 0.3, E1 - 1)) + [2.0]])
    thr = np.concatenate([thr, corner], axis=0)
    exits = np.stack([ls["check_2D_threshold"](conf, t) for t in thr])                      # (V + 4, N)
    exits2 = np.stack([ref_thresh.check_2D_threshold(conf, t) for t in thr])
    assert np.array_equal(exits, exits2)
    assert (exits[V] == 0).all() and (exits[V + 2] == 0).all()
    correct = (logits.argmax(-1) == refs[None, :])                                          # (E1, N)
    acc = np.array([np.mean(np.argmax(logits[e, np.arange(N)], axis=-1) == refs) for e in exits])      # EE/large_scale.py:88-91
    mean_exit = np.array([np.mean(e) for e in exits])                                                   # :92
    hist = np.stack([np.bincount(e, minlength=E1) for e in exits]).astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "sweep_ref.npz"), conf=conf, correct=correct.astype(np.uint8), thresholds=thr, accuracy=acc,
                        mean_exit=mean_exit, hist=hist, exits_first64=exits[:64].astype(np.int8), exits_corner=exits[V:].astype(np.int8),
                        n_generated=V, sha_logits=sha(logits))
    print("sweep_ref", thr.shape, "acc range", acc.min(), acc.max(), "mean exit range", mean_exit.min(), mean_exit.max())


def make_temperature_golden():
    """N4 pinned to the reference: ``TemperatureScaler`` (EE/generic_scaling.py:37-111) fitted per exit exactly as ``calibrate``
    drives it (EE/eval.py:298-329: ONE scaler object reused over the exits, so every fit is warm-started from the previous exit's
    temperature; L-BFGS-B on sklearn's log_loss)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import temperature_ref_inputs
    _stub_module("evaluate")
    import generic_scaling as gs
    gs.ece_logits = lambda *a, **k: float("nan")                  # feeds two print lines only (see the header)
    logits, refs = temperature_ref_inputs()
    T = gs.TemperatureScaler()
    temps, nll0, nll1 = [], [], []
    for e in range(logits.shape[0]):
        nll0.append(gs.manual_NLL(np.eye(logits.shape[2])[refs], logits[e]))
        T.fit(refs, logits[e])
        temps.append(float(T.temperature[0]))
        nll1.append(gs.manual_NLL(np.eye(logits.shape[2])[refs], T.temperature_scale(logits[e])))
    np.savez_compressed(os.path.join(HERE, "temperature_ref.npz"), temperature=np.array(temps), nll_before=np.array(nll0),
                        nll_after=np.array(nll1), sha_logits=sha(logits))
    print("temperature_ref", temps)


def make_large_golden():
    """BASELINE configs[2] shape from the composed reference: LayoutLMv3-large (H = 1024, L = 24, 16 heads, I = 4096), gate strategy, exits
    after layers 1..23 + final (EE/models/LayoutLMv3.py:764-792: the policy sees classifier(gate input)), 2 ragged documents at T = 512.
    CLS rows + exit logits + logits_store only."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import LARGE_GATE_EE, LARGE_GATE_SEEDS
    cfg = pkg.ModelConfig.large(EE_config=LARGE_GATE_EE)
    W = pkg.synth.make_weights(cfg, seed=LARGE_GATE_SEEDS["seed_w"])
    docs = pkg.synth.make_documents(cfg, 2, seed=LARGE_GATE_SEEDS["seed_docs"], text_len=512)
    res = run_reference(cfg, W, docs, LARGE_GATE_EE)
    policy_fixture(res["logits_store"], [0.0, 0.5, 0.9, 1.0 + 1e-6], "pol", res)
    keep = {k: v for k, v in res.items() if k not in ("emb_out_row1", "emb_out_lastrow", "layer1_row1")}
    np.savez_compressed(os.path.join(HERE, "large_gate.npz"), n_docs=2, text_len=512, sha_input_ids=sha(docs["input_ids"]),
                        sha_pixel_values=sha(docs["pixel_values"]), **LARGE_GATE_SEEDS, **keep)
    print("large_gate", res["logits_store"].shape, res["logits"][0, :4])


def make_attentions_golden():
    """``output_attentions=True`` (EE/models/LayoutLMv3.py:157, 219-220, 301): the attention probabilities of every layer from the stock HF encoder
    (``LayoutLMv3SelfAttention.forward`` returns ``attention_probs``; transformers 5.15 collects them through its output-capturing hooks) on
    the tiny configuration -- 3 documents, one of them without padding, S = 48 + 17.  ``head_mask`` has no counterpart in 5.x (the argument was
    dropped; the reference's 4.26 multiplies the probabilities by it) and therefore no fixture: parity unpinned for that argument."""
    ee = dict(exits=[1, 3], encoder_layer_strategy="ramp", inference_strategy="max_confidence")
    cfg = pkg.ModelConfig.tiny(EE_config=ee)
    W = pkg.synth.make_weights(cfg, seed=7)
    docs = pkg.synth.make_documents(cfg, 3, seed=11, text_len=48, min_words=3)
    docs["attention_mask"][2, :] = 1
    docs["input_ids"][2, docs["input_ids"][2] == cfg.pad_token_id] = 5
    m, _ = hf_model(cfg, W, ee)
    t = {k: torch.from_numpy(v) for k, v in docs.items() if k != "labels"}
    out = m.layoutlmv3(**t, output_attentions=True)
    att = torch.stack(list(out.attentions)).numpy()                       # (L, B, heads, S, S)
    assert att.shape == (cfg.num_hidden_layers, 3, cfg.num_attention_heads, 65, 65)
    np.savez_compressed(os.path.join(HERE, "tiny_attentions.npz"), seed_w=7, seed_docs=11, text_len=48, n_docs=3, attentions=att,
                        **{"in_" + k: v for k, v in docs.items()})
    print("tiny_attentions", att.shape, float(att.sum(-1).min()), float(att.sum(-1).max()))


def make_matrix_golden():
    """Criterion / head-depth / strategy matrix at the smallest split-precision shape (H = 256) and at base shape: entropy criterion,
    one-layer heads, gates, vision_avg / text_avg exits (EE/models/EE_modules.py:116-160, EE/models/LayoutLMv3.py:70-93, 465-605,
    764-792).  Same composed reference as main(); case table in tests/conftest.py (MATRIX_CASES)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import MATRIX_CASES, MATRIX_SEEDS, matrix_config
    for name in MATRIX_CASES:
        cfg, ee, n_docs, T = matrix_config(pkg, name)
        W = pkg.synth.make_weights(cfg, seed=MATRIX_SEEDS["seed_w"])
        docs = pkg.synth.make_documents(cfg, n_docs, seed=MATRIX_SEEDS["seed_docs"], text_len=T, min_words=3)
        res = run_reference(cfg, W, docs, ee)
        conf = np.exp(res["logits_store"] - res["logits_store"].max(-1, keepdims=True))
        conf = np.sort(np.unique(np.round((conf / conf.sum(-1, keepdims=True)).max(-1), 6)))
        mid = float(conf[len(conf) // 2]) + 1e-4
        policy_fixture(res["logits_store"], [0.0, mid, 0.9, 1.0 + 1e-6], "pol", res)
        keep = {k: v for k, v in res.items() if k not in ("emb_out_row1", "emb_out_lastrow", "layer1_row1")}
        np.savez_compressed(os.path.join(HERE, name + ".npz"), n_docs=n_docs, text_len=T, sha_input_ids=sha(docs["input_ids"]),
                            sha_pixel_values=sha(docs["pixel_values"]), **MATRIX_SEEDS, **keep)
        print(name, "E =", res["exit_logits"].shape[0], "store[:, 0, :3] =", res["logits_store"][:, 0, :3].ravel()[:6])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "matrix":
        make_matrix_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "preprocess":
        make_preprocess_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "dit":
        make_dit_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "sweep":
        make_sweep_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "temperature":
        make_temperature_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "large":
        make_large_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "attentions":
        make_attentions_golden()
    else:
        main()
        make_preprocess_golden()
        make_dit_golden()
        make_matrix_golden()
        make_sweep_golden()
        make_temperature_golden()
        make_large_golden()
        make_attentions_golden()
