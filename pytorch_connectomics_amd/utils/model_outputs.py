"""Pick the tensor to blend from a model output: Tensor | {"output": Tensor, "ds_*": ...} |
{"output": {head: Tensor}}  (contract of the reference's connectomics/utils/model_outputs.py:244-300)."""
from __future__ import annotations

from collections.abc import Mapping
from typing import Any, Optional

import torch


def _cfg_value(obj: Any, key: str, default: Any = None) -> Any:
    if obj is None:
        return default
    if isinstance(obj, Mapping):
        return obj.get(key, default)
    return getattr(obj, key, default)


def get_inference_model_value(cfg: Any, key: str, default: Any = None) -> Any:
    return _cfg_value(_cfg_value(_cfg_value(cfg, "inference", None), "model", None), key, default)


def get_inference_select_channel(cfg: Any) -> Any:
    return get_inference_model_value(cfg, "select_channel", None)


def get_inference_channel_activations(cfg: Any) -> list:
    value = get_inference_model_value(cfg, "channel_activations", None)
    return list(value) if isinstance(value, (list, tuple)) else []


def get_model_head_names(cfg: Any) -> list[str]:
    heads = _cfg_value(_cfg_value(cfg, "model", None), "heads", None) or {}
    return list(heads.keys()) if isinstance(heads, Mapping) else []


def unwrap_main_output(outputs: Any) -> Any:
    if isinstance(outputs, Mapping) and "output" in outputs:
        return outputs["output"]
    return outputs


def select_output_tensor(outputs: Any, *, requested_head: Optional[str] = None, primary_head: Optional[str] = None,
                         purpose: str = "output selection") -> tuple[torch.Tensor, Optional[str]]:
    out = unwrap_main_output(outputs)
    if isinstance(out, torch.Tensor):
        if requested_head is not None:
            raise ValueError(f"{purpose} requested head '{requested_head}', but the model output is a single tensor.")
        return out, None
    if not isinstance(out, Mapping):
        raise TypeError(f"{purpose} expected a tensor or mapping, got {type(out).__name__}.")
    if not out:
        raise ValueError(f"{purpose} received an empty output mapping.")
    head = requested_head
    if head is None:
        if primary_head is not None and primary_head in out:
            head = primary_head
        elif len(out) == 1:
            head = next(iter(out.keys()))
        else:
            raise ValueError(f"{purpose} requires an explicit head because available output heads are "
                             f"{sorted(out.keys())}.")
    if head not in out:
        raise ValueError(f"{purpose} requested head '{head}', but available output heads are {sorted(out.keys())}.")
    sel = out[head]
    if not isinstance(sel, torch.Tensor):
        raise TypeError(f"{purpose} requires head '{head}' to be a tensor, got {type(sel).__name__}.")
    return sel, head


__all__ = ["select_output_tensor", "unwrap_main_output", "get_inference_select_channel",
           "get_inference_channel_activations", "get_inference_model_value", "get_model_head_names"]
