"""Exit / model configuration for the MI355X early-exit path.

Mirrors the reference's configuration surface so that the same ``EE_config`` dictionaries work:

* ``ExitConfig``            <- EE/models/EE_modules.py:175-195 (same keys, same defaults, same validation)
* ``EarlyExitInference``    <- EE/models/EE_modules.py:116-146 (criterion + comparison sign)
* ``EarlyExitHead``         <- EE/models/EE_modules.py:167-172
* ``POSSIBLE_EXITS``        <- EE/models/LayoutLMv3.py:38
* exit-string parsing       <- EE/models/LayoutLMv3.py:100-108
* ``ModelConfig``           <- the HF ``LayoutLMv3Config`` fields the hot path reads (HF modeling_layoutlmv3.py)
"""
from __future__ import annotations

import json
import operator
import os
from dataclasses import dataclass, field, asdict
from enum import Enum
from typing import Any, Callable, Dict, List, Optional, Sequence, Union

EMBEDDING_EXITS = ("vision_avg", "text_avg", "text_visual_concat")  # fixed evaluation order, EE/models/LayoutLMv3.py:465-605
POSSIBLE_EXITS = list(EMBEDDING_EXITS) + list(range(1, 13))


class _StrEnum(str, Enum):
    def __str__(self) -> str:  # EE/models/EE_modules.py:49-57
        return self.value

    @classmethod
    def _missing_(cls, value):
        raise ValueError(f"{value} is not a valid {cls.__name__}, please select one from {cls.all()}")

    @classmethod
    def all(cls) -> List[str]:
        return [c.value for c in cls]


class EarlyExitInference(_StrEnum):
    MAX_CONFIDENCE = "max_confidence"
    ENTROPY = "entropy"
    PATIENCE = "patience"
    LTE = "lte"

    def get_sign(self) -> Callable:
        # EE/models/EE_modules.py:137-144: max_confidence exits when crit > thr, entropy when crit < thr
        if self == EarlyExitInference.MAX_CONFIDENCE:
            return operator.gt
        if self == EarlyExitInference.ENTROPY:
            return operator.lt
        raise NotImplementedError(f"{self} not implemented")

    @property
    def code(self) -> int:
        """Integer handed to the C-ABI (include/mmee.h ``ee_config.criterion``)."""
        if self == EarlyExitInference.MAX_CONFIDENCE:
            return 0
        if self == EarlyExitInference.ENTROPY:
            return 1
        raise NotImplementedError(f"{self} not implemented")


class EarlyExitHead(_StrEnum):
    GATE = "gate"
    RAMP = "ramp"
    EMBEXIT = "embexit"


class EarlyExitStrategy(_StrEnum):
    # training strategies are accepted (so reference configs validate) but unused on the inference path
    JOINT = "joint"
    JOINT_W_AVG = "joint_weighted_avg"
    JOINT_W = "joint_weighted"
    TWO_STAGE = "two-stage"
    ALTERNATING = "alternating"
    LAYERWISE = "layerwise"
    ONE_STAGE_SUBGRAPHS = "one_stage_subgraphs"
    ONE_STAGE_SUBGRAPHS_WEIGHTED = "one_stage_subgraphs_weighted"
    ONE_STAGE_SUBGRAPHS_ENTROPYREG = "one_stage_subgraphs_entropyreg"
    ONE_STAGE_SUBGRAPHS_WEIGHTED_ENTROPYREG = "one_stage_subgraphs_weighted_entropyreg"
    TWO_STAGE_SUBGRAPHS = "two_stage_subgraphs"
    TWO_STAGE_SUBGRAPHS_WEIGHTED = "two_stage_subgraphs_weighted"
    TWO_STAGE_SUBGRAPHS_ENTROPYREG = "two_stage_subgraphs_entropyreg"
    TWO_STAGE_SUBGRAPHS_WEIGHTED_ENTROPYREG = "two_stage_subgraphs_weighted_entropyreg"


def parse_exits(exits: Union[str, Sequence[Union[str, int]]]) -> List[Union[str, int]]:
    """Comma string -> ints (encoder layers) / strings (embedding exits); EE/models/LayoutLMv3.py:100-108."""
    if isinstance(exits, str):
        out: List[Union[str, int]] = []
        for e in exits.split(","):
            try:
                out.append(int(e))
            except ValueError:
                out.append(e)
        return out
    return list(exits)


class ExitConfig:
    """Same keys/defaults as EE/models/EE_modules.py:175-195."""

    def __init__(self, **kwargs):
        self.training_strategy = EarlyExitStrategy(kwargs.get("training_strategy", "joint_weighted_avg"))
        self.inference_strategy = EarlyExitInference(kwargs.get("inference_strategy", "max_confidence"))
        self.global_threshold = kwargs.get("global_threshold", 0.9)
        self.exits = parse_exits(kwargs.get("exits", ["text_avg", "vision_avg", 1, 4, 8]))
        self.encoder_layer_strategy = EarlyExitHead(kwargs.get("encoder_layer_strategy", "ramp"))
        self.exit_head_num_layers = kwargs.get("exit_head_num_layers", 2)

    # ---- derived views used by the hot path -------------------------------------------------
    @property
    def embedding_exits(self) -> List[str]:
        """Embedding-level exits in the order the reference evaluates them (vision, text, concat)."""
        return [e for e in EMBEDDING_EXITS if e in self.exits]

    @property
    def encoder_exit_layers(self) -> List[int]:
        """1-based encoder layers with an exit head, ascending: head k serves the k-th layer reached
        (EE/models/LayoutLMv3.py:222-227 picks ``early_exits[len(all_early_exits)]``)."""
        return sorted(int(e) for e in self.exits if isinstance(e, int))

    @property
    def num_exits(self) -> int:
        """Number of early exits E (the harness stores E+1 rows, EE/utils.py:160-164)."""
        return len(self.embedding_exits) + len(self.encoder_exit_layers)

    def as_dict(self) -> Dict[str, Any]:
        return {
            "training_strategy": str(self.training_strategy),
            "inference_strategy": str(self.inference_strategy),
            "global_threshold": self.global_threshold,
            "exits": list(self.exits),
            "encoder_layer_strategy": str(self.encoder_layer_strategy),
            "exit_head_num_layers": self.exit_head_num_layers,
        }


@dataclass
class ModelConfig:
    """The HF ``LayoutLMv3Config`` fields the inference path reads (defaults = layoutlmv3-base)."""

    vocab_size: int = 50265
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    max_position_embeddings: int = 514
    type_vocab_size: int = 1
    layer_norm_eps: float = 1e-5
    pad_token_id: int = 1
    max_2d_position_embeddings: int = 1024
    coordinate_size: int = 128
    shape_size: int = 128
    rel_pos_bins: int = 32
    max_rel_pos: int = 128
    rel_2d_pos_bins: int = 64
    max_rel_2d_pos: int = 256
    input_size: int = 224
    patch_size: int = 16
    num_channels: int = 3
    num_labels: int = 16
    hidden_act: str = "gelu"
    has_relative_attention_bias: bool = True
    has_spatial_attention_bias: bool = True
    EE_config: Dict[str, Any] = field(default_factory=dict)
    # model family: "layoutlmv3" (the reference's EE model) or "beit" (image-only DiT, BASELINE configs[4]; the
    # reference's "dit" branch loads a stock BEiT classifier, EE/configs.py:429-449)
    arch: str = "layoutlmv3"
    layer_scale_init_value: float = 0.1          # BEiT: lambda_1 / lambda_2 present when > 0
    use_mean_pooling: bool = True                # BEiT pooler: LayerNorm(mean of patch tokens)
    use_absolute_position_embeddings: bool = True
    use_relative_position_bias: bool = False
    use_shared_relative_position_bias: bool = False

    def __post_init__(self):
        if self.hidden_size % self.num_attention_heads:
            raise ValueError("hidden_size must be a multiple of num_attention_heads")
        if self.hidden_act != "gelu":
            raise ValueError("only the erf GELU of the reference checkpoints is implemented")
        if self.arch == "beit":
            if self.use_relative_position_bias or self.use_shared_relative_position_bias:
                raise ValueError("BEiT relative position bias is not built (DiT uses absolute position embeddings)")
            if not self.use_mean_pooling:
                raise ValueError("only use_mean_pooling=True (DiT) is built")
            return
        if self.arch != "layoutlmv3":
            raise ValueError(f"unknown arch {self.arch!r}")
        if 4 * self.coordinate_size + 2 * self.shape_size != self.hidden_size:
            raise ValueError("4*coordinate_size + 2*shape_size must equal hidden_size (HF:112-136 concat)")
        if not (self.has_relative_attention_bias and self.has_spatial_attention_bias):
            raise ValueError("the path implements the LayoutLMv3 relative + spatial attention bias")

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def num_patches(self) -> int:
        return (self.input_size // self.patch_size) ** 2

    @property
    def visual_len(self) -> int:
        return self.num_patches + 1

    @property
    def exit_config(self) -> ExitConfig:
        return ExitConfig(**self.EE_config)

    # ---- named shapes ---------------------------------------------------------------------------
    @classmethod
    def base(cls, **kw) -> "ModelConfig":
        return cls(**kw)

    @classmethod
    def large(cls, **kw) -> "ModelConfig":
        d = dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                 coordinate_size=171, shape_size=170)
        d.update(kw)
        return cls(**d)

    @classmethod
    def tiny(cls, **kw) -> "ModelConfig":
        """Small shape for fixtures: same structure, every dimension a legal kernel shape."""
        d = dict(vocab_size=300, hidden_size=128, num_hidden_layers=4, num_attention_heads=2, intermediate_size=256,
                 max_position_embeddings=66, coordinate_size=24, shape_size=16, input_size=64, patch_size=16)
        d.update(kw)
        return cls(**d)

    @classmethod
    def dit_base(cls, **kw) -> "ModelConfig":
        """DiT-base = BEiT-base/16 at 224 px with absolute position embeddings, layer scale 0.1, mean pooling."""
        d = dict(arch="beit", layer_norm_eps=1e-12)
        d.update(kw)
        return cls(**d)

    @classmethod
    def dit_tiny(cls, **kw) -> "ModelConfig":
        d = dict(arch="beit", layer_norm_eps=1e-12, hidden_size=128, num_hidden_layers=4, num_attention_heads=2,
                 intermediate_size=256, input_size=64, patch_size=16)
        d.update(kw)
        return cls(**d)

    # ---- HF checkpoint-dir config.json ----------------------------------------------------------
    @classmethod
    def from_hf_dict(cls, d: Dict[str, Any]) -> "ModelConfig":
        names = {f for f in cls.__dataclass_fields__}
        kw = {k: v for k, v in d.items() if k in names}
        if d.get("model_type") == "beit":
            kw["arch"] = "beit"
            if "image_size" in d:
                kw["input_size"] = d["image_size"]
        if "num_labels" not in kw and "id2label" in d:
            kw["num_labels"] = len(d["id2label"])
        ee = d.get("EE_config") or d.get("exit_config") or {}
        kw["EE_config"] = {k: (str(v) if isinstance(v, Enum) else v) for k, v in ee.items()}
        return cls(**kw)

    @classmethod
    def from_pretrained(cls, path: str) -> "ModelConfig":
        with open(os.path.join(path, "config.json")) as f:
            return cls.from_hf_dict(json.load(f))

    def to_hf_dict(self) -> Dict[str, Any]:
        d = asdict(self)
        d["model_type"] = self.arch
        if self.arch == "beit":
            d["image_size"] = self.input_size
        d["id2label"] = {str(i): f"LABEL_{i}" for i in range(self.num_labels)}
        return d
