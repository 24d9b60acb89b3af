"""Pick the tensor to blend from a model output: Tensor | {"output": Tensor, "ds_*": ...} |
{"output": {head: Tensor}}  (contract of the reference's connectomics/utils/model_outputs.py:244-300)."""
from __future__ import annotations

from collections.abc import Mapping
from typing import Any, Optional

import torch


_ABSENT = object()


def _dig(node: Any, *path: str, default: Any = None) -> Any:
    """node[path[0]][path[1]]... through mappings and attribute namespaces alike; `default` as soon as a step is missing."""
    for key in path:
        if node is None:
            return default
        node = node.get(key, _ABSENT) if isinstance(node, Mapping) else getattr(node, key, _ABSENT)
        if node is _ABSENT:
            return default
    return node


def _cfg_value(obj: Any, key: str, default: Any = None) -> Any:
    return _dig(obj, key, default=default)


def get_inference_model_config(cfg: Any) -> Any:
    """The `inference.model` node (head / channel selection / activations at inference time), None when absent."""
    return _dig(cfg, "inference", "model")


def get_inference_model_value(cfg: Any, key: str, default: Any = None) -> Any:
    return _dig(cfg, "inference", "model", key, default=default)


def get_inference_select_channel(cfg: Any) -> Any:
    return get_inference_model_value(cfg, "select_channel")


def get_inference_channel_activations(cfg: Any) -> list:
    specs = get_inference_model_value(cfg, "channel_activations")
    return specs if isinstance(specs, list) else []                # like the reference: anything but a list (a tuple too) means "none"


def _named_heads(cfg: Any) -> Mapping:
    heads = _dig(cfg, "model", "heads")
    return heads if isinstance(heads, Mapping) else {}


def get_model_head_names(cfg: Any) -> list[str]:
    return list(_named_heads(cfg))


def _checked_head(name: Any, heads: Mapping, what: str, purpose: str) -> str:
    if not isinstance(name, str) or not name.strip():
        raise ValueError(f"{what} for {purpose} must be a non-empty string.")
    name = name.strip()
    if name not in heads:
        lead = f"Requested output head '{name}'" if what.startswith("Requested") else f"{what}='{name}'"
        raise ValueError(f"{lead} for {purpose} is not present in model.heads ({sorted(heads.keys())}).")
    return name


def resolve_output_head(cfg: Any, *, requested_head: Optional[str] = None, purpose: str = "output selection",
                        allow_none: bool = True) -> Optional[str]:
    """Which named head a caller means (reference utils/model_outputs.py:61-131): the explicit request, else
    `inference.model.head` (a comma-separated list there is a merged-inference spec, not a single head), else
    `model.primary_head`, else the only head; None for models without named heads."""
    heads = _named_heads(cfg)
    if not heads:
        return None
    if requested_head is not None:
        return _checked_head(requested_head, heads, "Requested output head", purpose)
    configured = get_inference_model_value(cfg, "head", None)
    if configured is not None and not (isinstance(configured, str) and "," in configured):
        return _checked_head(configured, heads, "Requested output head", purpose)
    primary = _dig(cfg, "model", "primary_head")
    if primary is not None:
        return _checked_head(primary, heads, "model.primary_head", purpose)
    if len(heads) == 1:
        return next(iter(heads))
    if allow_none:
        return None
    raise ValueError(f"{purpose} requires inference.model.head or model.primary_head when model.heads has "
                     f"multiple entries ({sorted(heads.keys())}).")


def resolve_output_heads(cfg: Any, *, purpose: str = "output selection") -> list[str]:
    """One or more heads for merged inference: `inference.model.head: "a,b,c"` in the order written, else the single
    resolved head (reference utils/model_outputs.py:146-172)."""
    heads = _named_heads(cfg)
    if not heads:
        return []
    configured = get_inference_model_value(cfg, "head", None)
    if isinstance(configured, str) and "," in configured:
        names = [h.strip() for h in configured.split(",") if h.strip()]
        if not names:
            raise ValueError(f"inference.model.head for {purpose} is an empty list.")
        unknown = [n for n in names if n not in heads]
        if unknown:
            raise ValueError(f"inference.model.head for {purpose} references unknown heads {unknown}; "
                             f"available: {sorted(heads.keys())}.")
        return names
    one = resolve_output_head(cfg, purpose=purpose, allow_none=True)
    return [one] if one else []


def resolve_configured_output_head(cfg: Any, *, purpose: str = "output selection", allow_none: bool = True) -> Optional[str]:
    """`resolve_output_head` for what the CONFIG asks for (no explicit request)."""
    return resolve_output_head(cfg, requested_head=None, purpose=purpose, allow_none=allow_none)


def _head_width(heads: Mapping, name: str) -> int:
    return int(_dig(heads[name], "out_channels", default=0))


def get_total_model_head_channels(cfg: Any) -> int:
    """Sum of the `out_channels` of every named head (0 without named heads)."""
    heads = _named_heads(cfg)
    return sum(_head_width(heads, name) for name in heads)


def resolve_output_channels(cfg: Any, *, requested_head: Optional[str] = None, purpose: str = "output selection",
                            allow_ambiguous: bool = True) -> Optional[int]:
    """How many channels the selected output carries (reference utils/model_outputs.py:175-221): the width of the requested /
    configured head, the SUM over a comma-separated head list (merged inference), `model.out_channels` for a model without named
    heads; None when several heads exist, none is selected and `allow_ambiguous`."""
    heads = _named_heads(cfg)
    if not heads:
        width = _dig(cfg, "model", "out_channels")
        return None if width is None else int(width)
    if isinstance(requested_head, str) and "," in requested_head:
        wanted = [part.strip() for part in requested_head.split(",") if part.strip()]
        unknown = [name for name in wanted if name not in heads]
        if unknown:
            raise ValueError(f"Requested output heads {unknown} for {purpose} not in model.heads ({sorted(heads.keys())}).")
        return sum(_head_width(heads, name) for name in wanted)
    if requested_head is None:
        merged = resolve_output_heads(cfg, purpose=purpose)
        if len(merged) > 1:
            return sum(_head_width(heads, name) for name in merged)
    chosen = resolve_output_head(cfg, requested_head=requested_head, purpose=purpose, allow_none=allow_ambiguous)
    return None if chosen is None else _head_width(heads, chosen)


def resolve_configured_output_channels(cfg: Any, *, purpose: str = "output selection", allow_ambiguous: bool = True) -> Optional[int]:
    return resolve_output_channels(cfg, requested_head=None, purpose=purpose, allow_ambiguous=allow_ambiguous)


def resolve_head_target_slice(cfg: Any, head_name: str):
    """The label `target_slice` configured for a named head; None for an unknown head or a head without one."""
    heads = _named_heads(cfg)
    return _dig(heads[head_name], "target_slice") if head_name in heads else None


def unwrap_main_output(outputs: Any) -> Any:
    return outputs["output"] if isinstance(outputs, Mapping) and "output" in outputs else outputs


def select_output_tensor(outputs: Any, *, requested_head: Optional[str] = None, primary_head: Optional[str] = None,
                         purpose: str = "output selection") -> tuple[torch.Tensor, Optional[str]]:
    """-> (tensor, head name | None).  A plain tensor has no head to ask for; in a {head: tensor} mapping the head is the
    request, else `primary_head` when present, else the only entry (messages: reference utils/model_outputs.py:244-300)."""
    main = unwrap_main_output(outputs)
    if torch.is_tensor(main):
        if requested_head is None:
            return main, None
        raise ValueError(f"{purpose} requested head '{requested_head}', but the model output is a single tensor.")
    if not isinstance(main, Mapping):
        raise TypeError(f"{purpose} expected a tensor or mapping, got {type(main).__name__}.")
    names = sorted(main)
    if not names:
        raise ValueError(f"{purpose} received an empty output mapping.")
    head = requested_head
    if head is None:
        head = primary_head if primary_head in main and primary_head is not None else (names[0] if len(names) == 1 else None)
        if head is None:
            raise ValueError(f"{purpose} requires an explicit head because available output heads are {names}.")
    elif head not in main:
        raise ValueError(f"{purpose} requested head '{head}', but available output heads are {names}.")
    if not torch.is_tensor(main[head]):
        raise TypeError(f"{purpose} requires head '{head}' to be a tensor, got {type(main[head]).__name__}.")
    return main[head], head


__all__ = ["select_output_tensor", "resolve_output_head", "resolve_output_heads", "unwrap_main_output", "get_inference_select_channel",
           "get_inference_channel_activations", "get_inference_model_value", "get_inference_model_config", "get_model_head_names",
           "get_total_model_head_channels", "resolve_configured_output_head", "resolve_output_channels",
           "resolve_configured_output_channels", "resolve_head_target_slice"]
