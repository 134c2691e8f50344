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
