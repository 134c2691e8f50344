import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("multi-modal-early-exit_amd")


@pytest.fixture(scope="session")
def oracle():
    return importlib.import_module("oracle.ee_oracle")


def report_measured(test, what, value):
    """Record a MEASURED parity error (not a tolerance): printed, and appended to gpurun_out/parity_measured.txt so that a GPU run
    leaves the numbers behind (copied to profiles/ per round)."""
    line = f"{test}: {what} = {value:.3e}"
    print(line)
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_measured.txt"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


TINY_CASES = {
    "tiny_ramp": dict(exits=["vision_avg", "text_avg", "text_visual_concat", 1, 2, 3, 4],
                      encoder_layer_strategy="ramp", inference_strategy="max_confidence"),
    "tiny_gate": dict(exits=["text_visual_concat", 2, 3], encoder_layer_strategy="gate",
                      inference_strategy="max_confidence"),
    "tiny_entropy_1layer_head": dict(exits=[1, 3], encoder_layer_strategy="ramp", inference_strategy="entropy",
                                     exit_head_num_layers=1),
}
BASE_EE = dict(exits=["text_visual_concat", 2, 4, 6, 8, 10], encoder_layer_strategy="ramp")
DIT_EE = dict(exits=[1, 2, 3, 4], encoder_layer_strategy="ramp", inference_strategy="max_confidence")
DIT_BASE_EE = dict(exits=[2, 4, 6, 8, 10], encoder_layer_strategy="ramp", inference_strategy="max_confidence")

# criterion / head / strategy matrix at the two shapes the split-precision kernels care about (tests/golden/make_golden.py "matrix"):
# name -> (shape, EE_config, n_docs, text_len).  "h256" = the smallest shape the split kernels accept; "base" = LayoutLMv3-base, S = 709.
H256_KW = dict(hidden_size=256, num_attention_heads=4, intermediate_size=512, num_hidden_layers=3, coordinate_size=48, shape_size=32)
MATRIX_CASES = {
    "h256_entropy_1layer": ("h256", dict(exits=["vision_avg", "text_avg", "text_visual_concat", 1, 2, 3], encoder_layer_strategy="ramp",
                                          inference_strategy="entropy", exit_head_num_layers=1), 6, 48),
    "h256_gate": ("h256", dict(exits=["text_avg", 1, 2], encoder_layer_strategy="gate", inference_strategy="max_confidence"), 6, 48),
    "base_emb_entropy_1layer": ("base", dict(exits=["vision_avg", "text_avg", 6], encoder_layer_strategy="ramp",
                                             inference_strategy="entropy", exit_head_num_layers=1), 2, 512),
    "base_gate": ("base", dict(exits=["vision_avg", "text_visual_concat", 6], encoder_layer_strategy="gate",
                               inference_strategy="max_confidence"), 2, 512),
}
MATRIX_SEEDS = dict(seed_w=31, seed_docs=32)


def matrix_config(pkg, name):
    shape, ee, n_docs, T = MATRIX_CASES[name]
    cfg = pkg.ModelConfig.base(EE_config=ee) if shape == "base" else pkg.ModelConfig.tiny(EE_config=ee, **H256_KW)
    return cfg, ee, n_docs, T


# ---- round 5: inputs of the fixtures minted from the reference's own sweep / temperature code (tests/golden/make_golden.py "sweep", "temperature")
def sweep_ref_inputs(seed=2024, E1=7, N=400, K=16):
    """Seeded synthetic dumped logits (E1, N, K) float64 and labels (N,): later exits are sharper and more often right."""
    rng = np.random.default_rng(seed)
    store = rng.standard_normal((E1, N, K)) * np.linspace(1.0, 3.0, E1)[:, None, None]
    refs = rng.integers(0, K, N)
    store[:, np.arange(N), refs] += np.linspace(0.5, 3.0, E1)[:, None]
    return store, refs.astype(np.int64)


def temperature_ref_inputs(seed=77, E1=5, N=3000, K=16):
    """Seeded validation logits (E1, N, K) float64 whose exits are mis-calibrated by different factors, and labels (N,)."""
    rng = np.random.default_rng(seed)
    refs = rng.integers(0, K, N)
    base = rng.standard_normal((E1, N, K))
    base[:, np.arange(N), refs] += np.linspace(1.0, 3.0, E1)[:, None]
    gain = np.array([0.4, 1.0, 2.5, 6.0, 0.8])[:E1]          # over- and under-confident exits: optimal temperatures on both sides of 1
    return base * gain[:, None, None], refs.astype(np.int64)


LARGE_GATE_EE = dict(exits=list(range(1, 24)), encoder_layer_strategy="gate", inference_strategy="max_confidence")
LARGE_GATE_SEEDS = dict(seed_w=41, seed_docs=42)
