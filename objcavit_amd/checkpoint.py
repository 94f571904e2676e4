"""Checkpoint ingestion (row N3 of SURVEY.md section 8f): a reference run's weights into the drop-in model.

The reference trains ``GraphBinsLM`` (a LightningModule whose ``self.model`` is ``GraphBins`` / ``AdaBins``,
modules/GraphBinsLM.py:79-85) and resumes / evaluates through Lightning's ``load_from_checkpoint`` (main.py:26-28).
A Lightning ``.ckpt`` is a pickled dict; the weights sit under ``"state_dict"`` with every model key prefixed by the
attribute name -- ``model.`` -- next to the states of the loss and of the 16 torchmetrics objects
(``abs_rel.normed_abs_diff_total`` ...).  The drop-in modules expose the reference's keys and shapes exactly
(tests/test_host_logic.py), including the prototype-layer keys ``saca_1.image_encoder_layers.*`` that
``nn.TransformerEncoder`` leaves behind, so ingestion is: unwrap, strip the prefix, drop what is not the
model's, load strictly.  Nothing here needs Lightning or OmegaConf.
"""
from __future__ import annotations

from typing import Dict, Iterable, Mapping, Tuple

import torch

MODEL_PREFIX = "model."


def extract_model_state(ckpt: Mapping, prefix: str = MODEL_PREFIX) -> Dict[str, torch.Tensor]:
    """The model's ``state_dict`` out of a Lightning checkpoint dict, a bare LightningModule ``state_dict`` (keys
    ``model.*`` + metric / loss states) or an already bare model ``state_dict`` (returned unchanged)."""
    sd = ckpt["state_dict"] if "state_dict" in ckpt and isinstance(ckpt["state_dict"], Mapping) else ckpt
    if any(k.startswith(prefix) for k in sd):
        return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    return dict(sd)


def check_compatible(model: torch.nn.Module, state: Mapping[str, torch.Tensor]) -> Tuple[Iterable[str], Iterable[str], Iterable[str]]:
    """(missing keys, unexpected keys, shape mismatches) of ``state`` against ``model`` -- nothing is modified."""
    own = model.state_dict()
    missing = sorted(k for k in own if k not in state)
    unexpected = sorted(k for k in state if k not in own)
    shapes = sorted(f"{k}: checkpoint {tuple(state[k].shape)} vs model {tuple(own[k].shape)}"
                    for k in own if k in state and tuple(state[k].shape) != tuple(own[k].shape))
    return missing, unexpected, shapes


def load_reference_checkpoint(model: torch.nn.Module, path_or_ckpt, strict: bool = True, trust_pickle: bool = False):
    """Load a reference checkpoint (path to a Lightning ``.ckpt`` / ``torch.save``d state dict, or the loaded mapping)
    into a drop-in ``GraphBins`` / ``AdaBins``.  Returns the (missing, unexpected) key lists; with ``strict`` any
    difference raises, listing the keys.

    A Lightning checkpoint also pickles ``hyper_parameters`` (an OmegaConf tree): ``torch.load(weights_only=True)``
    refuses those classes.  ``trust_pickle=True`` falls back to a full unpickle -- only for files you trust."""
    ckpt = path_or_ckpt
    if not isinstance(ckpt, Mapping):
        try:
            ckpt = torch.load(path_or_ckpt, map_location="cpu", weights_only=True)
        except Exception as e:                                    # noqa: BLE001 -- re-raised with the remedy below
            if not trust_pickle:
                raise RuntimeError(f"{path_or_ckpt}: not loadable with weights_only=True ({type(e).__name__}: {e}); "
                                   "Lightning checkpoints embed OmegaConf hyper-parameters -- pass trust_pickle=True for "
                                   "a file you trust, or re-save its ['state_dict'] alone") from e
            ckpt = torch.load(path_or_ckpt, map_location="cpu", weights_only=False)
    state = extract_model_state(ckpt)
    missing, unexpected, shapes = check_compatible(model, state)
    if shapes or (strict and (missing or unexpected)):
        raise RuntimeError("checkpoint does not fit the model:\n  missing: %s\n  unexpected: %s\n  shapes: %s"
                           % (missing[:8], unexpected[:8], shapes[:8]))
    model.load_state_dict({k: v for k, v in state.items() if k not in unexpected}, strict=False)
    return missing, unexpected
