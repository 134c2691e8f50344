"""MI355X-native early-exit document-classification inference path (drop-in for the reference's eval hot path).

The directory name carries a hyphen, so import it with ``importlib.import_module("multi-modal-early-exit_amd")``
(or ``import mmee_amd`` — a two-line alias module at the repo root).
"""
from . import config, synth  # noqa: F401
from .config import (EarlyExitHead, EarlyExitInference, EarlyExitStrategy, ExitConfig, ModelConfig,  # noqa: F401
                     POSSIBLE_EXITS, parse_exits)
from . import capi  # noqa: F401,E402
from .engine import CapturedForward, EarlyExitEngine, EngineOutput, load_checkpoint_tensors, save_checkpoint  # noqa: F401,E402
from .microbatch import MicroBatchedEngine  # noqa: F401,E402
from .modeling import (DiTEEForImageClassification, EEModelOutput, EESequenceClassifierOutput,  # noqa: F401,E402
                       LayoutLMv3EEForSequenceClassification, load_local_processor)
from .policy import Policy, policy_scan_device  # noqa: F401,E402
from . import harness  # noqa: F401,E402
from . import dist  # noqa: F401,E402
from . import sweep  # noqa: F401,E402
from . import calibration  # noqa: F401,E402
from . import feed  # noqa: F401,E402
