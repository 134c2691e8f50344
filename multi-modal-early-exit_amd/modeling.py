"""The reference's model surface on top of the HIP path.

``LayoutLMv3EEForSequenceClassification.forward`` keeps the signature and the return contract of
EE/models/LayoutLMv3.py:696-711, 887-896 (``EESequenceClassifierOutput`` fields of EE/models/EE_modules.py:231-273), so
``utils.get_logits`` (EE/utils.py:179-193), ``eval.py`` and ``analysis.py`` consume it unchanged:

* like the reference (whose in-forward exit is commented out, EE/models/LayoutLMv3.py:270-281) ``forward`` evaluates
  EVERY exit for every document — it runs the HIP path in dump-all mode;
* ``early_exit`` is the new fast path of the north star: the policy test runs on the device between layers and
  documents that satisfy it leave; it returns ``(logits, exit_layer, confidence)``.

No arithmetic of the path happens here: tensors in, pointers across the C-ABI, tensors out.  The cross-entropy values
reported when ``labels`` are passed (``loss`` / ``exit_losses``, EE/models/LayoutLMv3.py:756-869) are evaluation
by-products computed from the returned logits.
"""
from __future__ import annotations

from collections import OrderedDict
from types import SimpleNamespace
from typing import Any, Dict, Iterator, Mapping, Optional, Sequence, Tuple, Union

import numpy as np

from . import capi
from .config import ExitConfig, ModelConfig
from .engine import EarlyExitEngine, EngineOutput, load_checkpoint_tensors, torch


class _ModelOutput(OrderedDict):
    """Minimal stand-in for ``transformers.ModelOutput``: attribute, key and integer access; ``None`` fields are
    skipped by integer indexing / ``to_tuple`` exactly as HF does."""

    _fields: Tuple[str, ...] = ()

    def __init__(self, **kw):
        super().__init__()
        for f in self._fields:
            v = kw.get(f)
            object.__setattr__(self, f, v)
            if v is not None:
                self[f] = v

    def __getitem__(self, k):
        if isinstance(k, str):
            return super().__getitem__(k)
        return self.to_tuple()[k]

    def to_tuple(self):
        return tuple(self[k] for k in self.keys())


class EEModelOutput(_ModelOutput):  # EE/models/EE_modules.py:198-228
    _fields = ("last_hidden_state", "hidden_states", "attentions", "exit_states", "gate_inputs", "lte_output")


class EESequenceClassifierOutput(_ModelOutput):  # EE/models/EE_modules.py:231-273
    _fields = ("logits", "loss", "hidden_states", "attentions", "exit_losses", "exit_criteria", "exit_states",
               "gated_logits", "lte_output")


def load_local_processor(path: str):
    """The ``AutoProcessor`` the reference attaches to the model (``AutoProcessor.from_pretrained(model_weights,
    apply_ocr=False)``, EE/models/LayoutLMv3.py:674-677; read by ``load_dataset`` through ``model.processor`` /
    ``.tokenizer``, EE/utils.py:83-98) — from a LOCAL directory only (there is no hub access).  Returns None when the directory
    holds no processor / tokenizer files or transformers cannot build one from them."""
    import os
    if not path or not os.path.isdir(path):
        return None
    if not any(os.path.exists(os.path.join(path, f)) for f in ("preprocessor_config.json", "processor_config.json", "tokenizer_config.json")):
        return None
    try:
        from transformers import AutoProcessor
        return AutoProcessor.from_pretrained(path, apply_ocr=False, local_files_only=True)
    except Exception as e:  # noqa: BLE001 - a broken processor directory must not stop the model from loading
        import warnings
        warnings.warn(f"processor files under {path} could not be loaded ({type(e).__name__}: {e}); model.processor stays None")
        return None


class LayoutLMv3EEForSequenceClassification:
    """Drop-in for the reference class of the same name (inference only)."""

    def __init__(self, config: Union[ModelConfig, Mapping[str, Any]], weights: Optional[Mapping[str, Any]] = None,
                 max_docs: int = 64, max_text_len: int = 512, precision: str = "auto", device=None, micro_batches: int = 1):
        """``micro_batches`` > 1: batches run as that many slices on as many handles and HIP streams (``MicroBatchedEngine``: bit-identical results,
        +0.9 % docs/s at two slices of >= 512 documents; each handle holds a copy of the weights and ``ceil(max_docs / micro_batches)`` documents of
        workspace).  ``output_hidden_states=True`` is a per-handle feature and runs on the first handle, a slice at a time."""
        if not isinstance(config, ModelConfig):
            config = ModelConfig.from_hf_dict(dict(config))
        else:
            # the model owns its configuration: _sync_exit_config writes the criterion override into EE_config, which must not be the CALLER's dictionary
            # (round 6: a test's override leaked into the module-level dictionary another test built its model from)
            import dataclasses
            config = dataclasses.replace(config, EE_config=dict(config.EE_config))
        self.model_config = config
        if micro_batches > 1:
            from .microbatch import MicroBatchedEngine
            self.engine = MicroBatchedEngine(config, max_docs=max_docs, max_text_len=max_text_len, precision=precision, device=device,
                                             micro_batches=micro_batches)
        else:
            self.engine = EarlyExitEngine(config, max_docs=max_docs, max_text_len=max_text_len, precision=precision,
                                          device=device)
        ec = config.exit_config
        # what callers read: .config.exit_config[...] / .num_labels / .id2label (EE/utils.py:62-78, 142-144, 161)
        self.config = SimpleNamespace(
            exit_config={"training_strategy": ec.training_strategy, "inference_strategy": ec.inference_strategy,
                         "global_threshold": ec.global_threshold, "exits": list(ec.exits),
                         "encoder_layer_strategy": ec.encoder_layer_strategy,
                         "exit_head_num_layers": ec.exit_head_num_layers},
            EE_config=dict(config.EE_config), num_labels=config.num_labels,
            id2label={i: f"LABEL_{i}" for i in range(config.num_labels)}, use_return_dict=True,
            hidden_size=config.hidden_size, num_hidden_layers=config.num_hidden_layers)
        self.num_labels = config.num_labels
        self.apply_gating = str(ec.encoder_layer_strategy) == "gate"
        # the reference fetches AutoProcessor(model_weights) from the hub (EE/models/LayoutLMv3.py:674-677); here
        # from_pretrained attaches one when the checkpoint directory (or EE_config["model_weights"], if that is a local
        # directory) holds processor files, else it stays None
        self.processor = load_local_processor(str(config.EE_config.get("model_weights", "")))
        self.training = False
        self._weights = None
        if weights is not None:
            self.load_weights(weights)

    # ---- loading ----------------------------------------------------------------------------------------------------
    def load_weights(self, weights: Mapping[str, Any]):
        self.engine.load_weights(weights)
        self._weights = weights

    @classmethod
    def from_pretrained(cls, path: str, config: Optional[Any] = None, **kw) -> "LayoutLMv3EEForSequenceClassification":
        """Local HF-format checkpoint directory only (EE/configs.py:389-411); hub names cannot be fetched offline."""
        cfg = ModelConfig.from_pretrained(path)
        ee = getattr(config, "EE_config", None) if config is not None else None
        if ee:
            cfg.EE_config.update({k: (str(v) if hasattr(v, "value") else v) for k, v in dict(ee).items()})
        m = cls(cfg, **kw)
        m.load_weights(load_checkpoint_tensors(path))
        m.processor = load_local_processor(path) or m.processor
        return m

    # ---- nn.Module-ish surface the harness touches -------------------------------------------------------------------
    @property
    def device(self):
        return self.engine.device

    def eval(self):
        self.training = False
        return self

    def to(self, *_a, **_k):
        return self

    def named_parameters(self) -> Iterator[Tuple[str, Any]]:
        if self._weights is None:
            return iter(())
        return ((k, v) for k, v in self._weights.items())

    def exit_criterion(self, logits):
        """``max_confidence`` / ``entropy`` on a logits tensor (EE/models/EE_modules.py:149-160)."""
        self._sync_exit_config()
        if str(self.model_config.exit_config.inference_strategy) == "max_confidence":
            return torch.softmax(logits, dim=1).max(dim=1)[0]
        e = torch.exp(logits)
        return torch.log(e.sum(1)) - (logits * e).sum(1) / e.sum(1)

    def _sync_exit_config(self):
        """A deliberate EXTENSION of the reference, not its behaviour: the reference binds ``exit_criterion`` / ``threshold_sign`` once in
        ``__init__`` (EE/models/LayoutLMv3.py:125-131, 342-348), so ``load_assets``' later write of
        ``model.config.exit_config["inference_strategy"]`` (EE/utils.py:62-78) never reaches its forward -- results agree there only because
        ``build_model`` merges the CLI configuration into ``EE_config`` BEFORE construction (EE/configs.py:389-393).  Here the dictionary is
        re-read at every call: a changed criterion is pushed down to the kernels (ee_set_criterion) AND into ``model_config.EE_config``, so that
        ``exit_criterion()``, ``to_hf_dict()`` and the handle agree; ``early_exit`` reads the threshold from the same dictionary.  A strategy
        the kernels do not implement (patience / lte) raises, as ``EarlyExitInference.get_sign`` does in the reference."""
        want = self.config.exit_config["inference_strategy"]
        want = str(getattr(want, "value", want))
        if want != str(self.engine.exit_config.inference_strategy):
            self.engine.set_criterion(want)
        if str(self.model_config.EE_config.get("inference_strategy")) != want:
            self.model_config.EE_config["inference_strategy"] = want
            self.config.EE_config["inference_strategy"] = want

    # ---- chunked engine call -------------------------------------------------------------------------------------------
    def _run(self, tensors: Dict[str, Any], **kw) -> EngineOutput:
        self._sync_exit_config()
        B = tensors["pixel_values"].shape[0]
        eng = self.engine
        if kw.get("want_hidden_states") and hasattr(eng, "engines"):       # per-handle feature: the first handle, a slice at a time
            eng = eng.engines[0]
        mb = eng.max_docs
        if B <= mb:
            return eng.forward(**tensors, **kw)
        parts = []
        for s in range(0, B, mb):
            sl = {k: (v[s:s + mb] if v is not None else None) for k, v in tensors.items()}
            parts.append(eng.forward(**sl, **kw))
        cat = lambda xs, d: None if xs[0] is None else torch.cat(xs, dim=d)
        return EngineOutput(cat([p.logits for p in parts], 0), cat([p.exit_layer for p in parts], 0),
                            cat([p.confidence for p in parts], 0), cat([p.all_logits for p in parts], 1),
                            cat([p.all_crit for p in parts], 1), cat([p.head_logits for p in parts], 1),
                            cat([p.head_crit for p in parts], 1), cat([p.hidden_cls for p in parts], 1),
                            cat([p.hidden_states for p in parts], 1), cat([p.attentions for p in parts], 1))

    # ---- the reference signature -------------------------------------------------------------------------------------
    def forward(self, input_ids=None, attention_mask=None, bbox=None, pixel_values=None, labels=None,
                token_type_ids=None, position_ids=None, head_mask=None, inputs_embeds=None, output_attentions=None,
                output_hidden_states=None, return_dict=None, **kwargs) -> EESequenceClassifierOutput:
        if (input_ids is None and inputs_embeds is None) or pixel_values is None:
            # (the reference's own forward cannot run these either: with pixel_values=None `visual_embeddings` is unbound at
            # EE/models/LayoutLMv3.py:550, with neither input_ids nor inputs_embeds `embedding_output` is at :565)
            raise ValueError("the HIP path implements the multimodal evaluation input: input_ids (or inputs_embeds) AND pixel_values "
                             "(EE/utils.py:93-98); text-only / image-only calls are not built")
        # head_mask / output_attentions (EE/models/LayoutLMv3.py:382-385, 631-641, 219-220): served by side kernels of this dump-all forward
        # (csrc/attention_maps.hip); the fused attention kernels of the fast path never materialise a map
        out = self._run(dict(input_ids=input_ids, attention_mask=attention_mask, bbox=bbox, pixel_values=pixel_values,
                             token_type_ids=token_type_ids, position_ids=position_ids, inputs_embeds=inputs_embeds),
                        dump_all=True, want_all=True, want_head=True, validate=True, want_hidden_states=bool(output_hidden_states),
                        head_mask=head_mask, want_attentions=bool(output_attentions))
        return self._pack(out, labels, return_dict)

    def _pack(self, out: EngineOutput, labels, return_dict):
        E = self.engine.E
        logits = out.all_logits[E]
        exit_states = tuple((out.head_logits[j], out.head_crit[j]) for j in range(E))
        gated, exit_criteria, exit_losses, loss = (), [], [], None
        if labels is not None:
            lab = labels.to(logits.device).view(-1) if torch.is_tensor(labels) else torch.as_tensor(labels, device=logits.device).view(-1)
            F = torch.nn.functional
            loss = F.cross_entropy(logits, lab)
            if self.apply_gating:                              # EE/models/LayoutLMv3.py:764-792
                gated = tuple(out.all_logits[j] for j in range(E))
                for j in range(E):
                    y = torch.zeros(lab.shape[0], 2, device=logits.device)
                    y[torch.arange(lab.shape[0]), (gated[j].argmax(-1) == lab).long()] = 1
                    exit_losses.append(F.binary_cross_entropy_with_logits(out.head_logits[j], y))
                    exit_criteria.append(out.head_crit[j])
            else:                                              # :860-869
                for j in range(E):
                    exit_losses.append(F.cross_entropy(out.head_logits[j], lab))
                    exit_criteria.append(out.head_crit[j])
        exit_criteria.append(out.all_crit[E])                  # :871-872
        # output_hidden_states (EE/models/LayoutLMv3.py:887-896 passes the encoder's tuple through): L + 1 tensors of (B, T + Pv, H)
        hidden_states = None if out.hidden_states is None else tuple(out.hidden_states[l] for l in range(out.hidden_states.shape[0]))
        attentions = None if out.attentions is None else tuple(out.attentions[l] for l in range(out.attentions.shape[0]))      # L x (B, heads, S, S)
        res = EESequenceClassifierOutput(loss=loss, logits=logits, hidden_states=hidden_states, attentions=attentions,
                                         exit_losses=exit_losses, exit_criteria=exit_criteria, exit_states=exit_states,
                                         gated_logits=gated)
        if return_dict is False:
            # EE/models/LayoutLMv3.py:883-885: `(logits,) + outputs[1:]`, the loss in front when labels were passed, where `outputs` is the
            # backbone's own tuple `(sequence_output, [all_hidden_states], [all_attentions])` (:287-296, 654-655) -- the exit fields exist
            # in the dictionary form only.  (With embedding-level exits or gates the reference's tuple path raises AttributeError at
            # :648-651, `encoder_outputs.exit_states` on a tuple; the same tuple is returned here for every configuration.)
            output = (logits,) + tuple(v for v in (hidden_states, attentions) if v is not None)
            return ((loss,) + output) if loss is not None else output
        return res


    __call__ = forward

    # ---- the fast path ------------------------------------------------------------------------------------------------
    def early_exit(self, input_ids, attention_mask=None, bbox=None, pixel_values=None, token_type_ids=None,
                   position_ids=None, thresholds: Optional[Union[float, Sequence[float]]] = None,
                   temperatures: Optional[Sequence[float]] = None, **kw) -> EngineOutput:
        """(logits, exit_layer, confidence) with the policy test on the device: identical to running ``forward`` on
        everything and then ``Policy(...)`` (EE/eval.py:87-98), but deeper layers only see the surviving documents.
        ``thresholds`` defaults to ``config.exit_config["global_threshold"]``."""
        if thresholds is None:
            thresholds = self.config.exit_config["global_threshold"]
        self._small_batch_schedule(pixel_values, kw)
        return self._run(dict(input_ids=input_ids, attention_mask=attention_mask, bbox=bbox, pixel_values=pixel_values,
                              token_type_ids=token_type_ids, position_ids=position_ids),
                         thresholds=thresholds, temperatures=temperatures, **kw)

    SMALL_BATCH_WHOLE_LAYERS = 16

    def _small_batch_schedule(self, pixel_values, kw):
        """Round 6: a probe in front of an exit layer is 13 launches; at the reference's operating point (``eval_batch_size = 1``, EE/configs.py:36)
        it never pays -- one document's forward takes 2.5 ms probe-first and 1.95 ms with whole layers, eight documents 5.2 / 4.2 ms, 64 the same
        (``bench.py`` ``small_batch``).  For batches of at most ``SMALL_BATCH_WHOLE_LAYERS`` documents ``early_exit`` therefore runs whole layers unless the
        caller chose a schedule (``whole_layers`` / ``probe_always`` / ``xprobe`` arguments, or ``engine.pin_schedule``): a function of the call's batch
        size, never of timing; exit indices are the same and whole layers are the form whose rows are bit-identical to the dump-all rows."""
        if pixel_values is None or any(k in kw for k in ("whole_layers", "probe_always", "xprobe")):
            return
        if getattr(self.engine, "_pinned", False):
            return
        if int(pixel_values.shape[0]) <= self.SMALL_BATCH_WHOLE_LAYERS:
            kw["whole_layers"] = True


class DiTEEForImageClassification(LayoutLMv3EEForSequenceClassification):
    """Image-only DiT / BEiT classifier with per-layer exit heads on the same HIP kernels (BASELINE configs[4]).

    The reference's "dit" branch loads a stock ``AutoModelForImageClassification`` (EE/configs.py:429-449) and defines no exit
    heads for it; heads here are ``LayoutLMv3Exit`` (EE/models/LayoutLMv3.py:56-93) on the CLS row after each exit layer
    (``beit.encoder.early_exits.k``) — a documented extrapolation.  ``forward(pixel_values, labels=None)`` returns
    the same ``EESequenceClassifierOutput`` fields; ``early_exit(pixel_values=...)`` is the fast path.

    What this wrapper refuses: ``head_mask`` and ``output_attentions`` (``NotImplementedError``: the side kernels of
    csrc/attention_maps.hip are built for the LayoutLMv3 layers, whose bias tables they read; BEiT relative-position bias is not
    built either), a call without ``pixel_values`` (``ValueError``), and every text argument of the LayoutLMv3 signature (there is no
    text stream).  ``output_hidden_states`` and ``micro_batches`` work as in the base class (hidden states are a per-handle feature and
    run on the first handle, a slice at a time)."""

    def forward(self, pixel_values=None, labels=None, head_mask=None, output_attentions=None, output_hidden_states=None,
                return_dict=None, **kwargs) -> EESequenceClassifierOutput:
        if pixel_values is None:
            raise ValueError("pixel_values is required")
        if head_mask is not None or output_attentions:
            raise NotImplementedError("head_mask / attention maps are not part of the evaluation hot path")
        out = self._run(dict(pixel_values=pixel_values), dump_all=True, want_all=True, want_head=True, validate=True,
                        want_hidden_states=bool(output_hidden_states))      # BeitEncoder: embedding output + every layer's output, (B, Pv, H)
        return self._pack(out, labels, return_dict)

    __call__ = forward

    def early_exit(self, pixel_values=None, thresholds=None, temperatures=None, **kw) -> EngineOutput:
        if thresholds is None:
            thresholds = self.config.exit_config["global_threshold"]
        self._small_batch_schedule(pixel_values, kw)
        return self._run(dict(pixel_values=pixel_values), thresholds=thresholds, temperatures=temperatures, **kw)
