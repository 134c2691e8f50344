"""Logit-harvesting harness: the build's counterpart of EE/utils.py:102-271 (``config_to_checkpoint``, ``get_logits``,
``dump_logits``) so that the reference's offline tooling (``eval.evaluate_checkpoint`` EE/eval.py:163-224,
``full_test_iteration`` :227-274, ``calibrate`` :277-346, ``large_scale.py``) finds the files it expects:

    results/<checkpoint>-<dataset>[-<n>i]/exit_logits-<name>.npz   (arr_0: float64 (E+1, N, K))
    results/<checkpoint>-<dataset>[-<n>i]/references-<name>.npz    (arr_0: labels (N,))
    results/<checkpoint>-<dataset>[-<n>i]/config.json

Differences from the reference, both deliberate: any batch size works (the reference indexes the store by *batch*
index and therefore assumes eval_batch_size = 1, EE/utils.py:188-193), and references are read from the batches as they
stream by instead of a second full pass over the loader (:136-138).
"""
from __future__ import annotations

import json
import os
from typing import Any, Dict, Iterable, Optional, Tuple

import numpy as np

from .engine import torch


def config_to_checkpoint(config: Dict[str, Any], root: str = "results") -> str:
    """EE/utils.py:114-122."""
    out = os.path.join(root, f"{str(config['checkpoint']).split('/')[-1]}-{str(config['test_dataset']).split('/')[-1]}")
    if config.get("downsampling"):
        out += f"-{config['downsampling']}i"
    return out


def dump_logits(model, logits, references, config: Dict[str, Any], name: str = "test", root: str = "results") -> str:
    """EE/utils.py:240-271 — same file names, same ``arr_0`` layout, same keys popped from config.json."""
    out = config_to_checkpoint(config, root)
    os.makedirs(out, exist_ok=True)
    if references is not None:
        np.savez_compressed(os.path.join(out, f"references-{name}.npz"), np.asarray(references))
    if torch is not None and torch.is_tensor(logits):
        logits = logits.detach().cpu().numpy()
    np.savez_compressed(os.path.join(out, f"exit_logits-{name}.npz"), np.asarray(logits))
    to_save = dict(config)
    ec = getattr(getattr(model, "config", None), "exit_config", None) or {}
    to_save.update({k: (str(v) if hasattr(v, "value") else v) for k, v in dict(ec).items()})
    for k in ("exit_threshold", "global_threshold", "inference_strategy", "exit_policy", "use_lte", "use_wandb",
              "calibrate", "full_test", "step", "epsilon"):
        to_save.pop(k, None)
    with open(os.path.join(out, "config.json"), "w+") as f:
        json.dump(to_save, f, indent=4, default=str)
    return out


def get_logits(model, config: Dict[str, Any], test_loader: Iterable[Dict[str, Any]], root: str = "results",
               use_cache: bool = True) -> Tuple[np.ndarray, np.ndarray, None]:
    """EE/utils.py:125-223: run ``model.forward`` over the loader and keep, per exit j, what the policy later sees —
    ``gated_logits[j]`` for gates, ``exit_states[j][0]`` for ramps — plus the final logits in the last row, as float64.
    Returns ``(logits_store (E+1,N,K), references (N,), None)`` and writes the npz/json triple."""
    label = config.get("labelset", "test")
    out = config_to_checkpoint(config, root)
    lp, rp = os.path.join(out, f"exit_logits-{label}.npz"), os.path.join(out, f"references-{label}.npz")
    if use_cache and os.path.exists(lp) and os.path.exists(rp):          # :148-158
        return np.load(lp)["arr_0"], np.load(rp)["arr_0"], None
    nr_exits = len(model.config.exit_config["exits"])
    rows, refs = [], []
    limit = config.get("downsampling") or None
    seen = 0
    for batch in test_loader:
        labels = batch.get("labels")
        fwd = {k: v for k, v in batch.items() if k in ("input_ids", "attention_mask", "bbox", "pixel_values",
                                                        "token_type_ids", "position_ids", "labels")}
        outputs = model.forward(**fwd)
        B = outputs.logits.shape[0]
        store = torch.empty((nr_exits + 1, B, outputs.logits.shape[1]), dtype=torch.float64, device=outputs.logits.device)
        for j in range(nr_exits):                                         # :182-192
            if outputs.gated_logits is not None and len(outputs.gated_logits) > 0:
                store[j] = outputs.gated_logits[j]
            else:
                store[j] = outputs.exit_states[j][0]
        store[-1] = outputs.logits
        rows.append(store.cpu().numpy())
        if labels is not None:
            refs.append(np.asarray(labels.cpu() if torch.is_tensor(labels) else labels).reshape(-1))
        seen += B
        if limit and seen >= limit:
            break
    logits_store = np.concatenate(rows, axis=1)
    references = np.concatenate(refs) if refs else None
    if limit:
        logits_store = logits_store[:, :limit]
        references = references[:limit] if references is not None else None
    cfg = dict(config)
    cfg["labelset"] = "test"                                              # :220
    dump_logits(model, logits_store, references, cfg, name="test" if label == "test" else "validation", root=root)
    return logits_store, references, None
