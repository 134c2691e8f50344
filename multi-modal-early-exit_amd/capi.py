"""ctypes binding of libmmee_hip.so (include/mmee.h).  No torch types cross this boundary: pointers and sizes only.

The library is the product: if it cannot be loaded (not built, or no ROCm runtime) every entry point raises
``MMEEUnavailable`` — there is no CPU / PyTorch fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

ABI_VERSION = 3
MAX_ENCODER_EXITS = 64
EXIT_KIND = {"vision_avg": 0, "text_avg": 1, "text_visual_concat": 2}
FLAG_DENSE_ROWS = 1
FLAG_NO_EXIT = 2
FLAG_WHOLE_LAYERS = 4
FLAG_PROBE_ALWAYS = 8
FLAG_XPROBE = 16
FLAG_ONE_TERM = 32
DT_F32, DT_F16, DT_BF16 = 0, 1, 2
CLOCK_STAMP_WORDS = 4096       # MMEE_CLOCK_STAMP_WORDS

_LIB_NAME = "libmmee_hip.so"
_LIB_PATH = os.environ.get("MMEE_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), _LIB_NAME)


class MMEEUnavailable(RuntimeError):
    pass


class MMEEError(RuntimeError):
    pass


class EEConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("hidden_size", C.c_int32), ("num_hidden_layers", C.c_int32), ("num_attention_heads", C.c_int32),
        ("intermediate_size", C.c_int32),
        ("vocab_size", C.c_int32), ("max_position_embeddings", C.c_int32), ("type_vocab_size", C.c_int32),
        ("pad_token_id", C.c_int32),
        ("max_2d_position_embeddings", C.c_int32), ("coordinate_size", C.c_int32), ("shape_size", C.c_int32),
        ("rel_pos_bins", C.c_int32), ("max_rel_pos", C.c_int32), ("rel_2d_pos_bins", C.c_int32),
        ("max_rel_2d_pos", C.c_int32),
        ("input_size", C.c_int32), ("patch_size", C.c_int32), ("num_channels", C.c_int32), ("num_labels", C.c_int32),
        ("layer_norm_eps", C.c_float),
        ("n_embedding_exits", C.c_int32), ("embedding_exits", C.c_int32 * 3),
        ("n_encoder_exits", C.c_int32), ("encoder_exit_layers", C.c_int32 * MAX_ENCODER_EXITS),
        ("exit_head_num_layers", C.c_int32), ("strategy", C.c_int32), ("criterion", C.c_int32),
        ("max_docs", C.c_int32), ("max_text_len", C.c_int32), ("precision", C.c_int32),
        ("arch", C.c_int32), ("use_abs_pos", C.c_int32), ("layer_scale", C.c_int32), ("use_mean_pooling", C.c_int32),
    ]


# every symbol include/mmee.h declares: (restype, argtypes)
_vp, _i32, _u32 = C.c_void_p, C.c_int32, C.c_uint32
SYMBOLS = {
    "ee_create": (C.c_int, [C.POINTER(EEConfig), C.POINTER(_vp)]),
    "ee_destroy": (C.c_int, [_vp]),
    "ee_last_error": (C.c_char_p, [_vp]),
    "ee_load_tensor": (C.c_int, [_vp, C.c_char_p, _vp, C.POINTER(C.c_int64), _i32, _i32, _i32]),
    "ee_finalize": (C.c_int, [_vp]),
    "ee_num_expected_tensors": (_i32, [_vp]),
    "ee_expected_tensor_name": (C.c_char_p, [_vp, _i32]),
    "ee_forward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, C.POINTER(C.c_double),
                             C.POINTER(C.c_double), _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ee_graph_capture": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, C.POINTER(C.c_double),
                                   C.POINTER(C.c_double), _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_i32)]),
    "ee_graph_launch": (C.c_int, [_vp, _i32, C.POINTER(C.c_double), C.POINTER(C.c_double), _vp]),
    "ee_graph_destroy": (C.c_int, [_vp, _i32]),
    "ee_last_stage_counts": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32), _i32, C.POINTER(_i32), _vp]),
    "ee_last_flops": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double), _vp]),
    "ee_last_layer_plan": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), _i32, C.POINTER(C.c_double), _vp]),
    "ee_set_probe_mask": (C.c_int, [_vp, _i32, C.c_uint64]),
    "ee_set_criterion": (C.c_int, [_vp, _i32]),
    "ee_suggest_probe_mask": (C.c_int, [_vp, _u32, C.POINTER(C.c_uint64), _vp]),
    "ee_clock_stamp": (C.c_int, [_vp, _vp]),
    "ee_set_inputs_embeds": (C.c_int, [_vp, _vp]),
    "ee_set_hidden_states_out": (C.c_int, [_vp, _vp]),
    "ee_set_head_mask": (C.c_int, [_vp, _vp]),
    "ee_set_attentions_out": (C.c_int, [_vp, _vp]),
    "ee_policy_scan": (C.c_int, [_vp, _i32, _i32, _i32, C.POINTER(C.c_double), _vp, _vp, _vp, _vp, _vp]),
    "ee_pack_results": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "ee_unpack_results": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    "ee_threshold_sweep": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp]),
    "ee_msp_table": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "ee_debug_gemm": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "ee_debug_gemm_split": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, C.c_float, C.c_float, C.c_float, _vp,
                                      _i32, _i32, C.POINTER(C.c_float), _vp]),
    "ee_debug_attn_stamps": (C.c_int, [_vp]),
    "ee_temperature_fit": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ee_preprocess_images": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, C.c_size_t, _vp, _vp, _vp]),
    "ee_preprocess_workspace_bytes": (C.c_size_t, [_i32, _i32, _i32]),
    "ee_collate_pad": (C.c_int, [_vp, _vp, _vp, _i32, _i32, C.c_int64, _vp, _vp, _vp, _vp]),
    "ee_bucket_lut": (C.c_int, [_i32, _i32, _i32, _vp]),
    "ee_profile": (C.c_int, [_vp, _i32]),
    "ee_profile_read": (C.c_int, [_vp, _i32, C.c_char_p, _i32, C.POINTER(C.c_double), C.POINTER(_i32)]),
}

_lib: Optional[C.CDLL] = None


def lib_path() -> str:
    return _LIB_PATH


def load() -> C.CDLL:
    """Load the HIP library (once).  Raises MMEEUnavailable with the reason when it cannot be loaded."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise MMEEUnavailable(
            f"{_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C {os.path.dirname(_LIB_PATH)}/csrc`).  There is no CPU fallback.")
    try:
        lib = C.CDLL(_LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:  # e.g. ROCm runtime missing
        raise MMEEUnavailable(f"cannot load {_LIB_PATH}: {e}") from e
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise MMEEUnavailable(f"{_LIB_PATH} does not export {name} (stale build?)") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error(handle=None) -> str:
    msg = load().ee_last_error(handle)
    return msg.decode() if msg else ""


def check(rc: int, handle=None, what: str = ""):
    if rc != 0:
        raise MMEEError(f"{what}: {last_error(handle)}" if what else last_error(handle))
