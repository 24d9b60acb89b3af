"""Architecture registry -- same contract as the reference's
connectomics/models/architectures/registry.py:14-120 (names, warning on overwrite, ValueError
listing the available names on a miss), re-implemented."""
from __future__ import annotations

import warnings
from typing import Callable, Dict, List

_ARCHITECTURE_REGISTRY: Dict[str, Callable] = {}


def register_architecture(name: str):
    """Decorator: ``@register_architecture("my_model") def build(cfg) -> nn.Module``."""
    def decorator(builder_fn: Callable) -> Callable:
        if name in _ARCHITECTURE_REGISTRY:
            warnings.warn(f"Architecture '{name}' already registered. Overwriting previous registration.",
                          UserWarning)
        _ARCHITECTURE_REGISTRY[name] = builder_fn
        return builder_fn
    return decorator


def get_architecture_builder(name: str) -> Callable:
    if name not in _ARCHITECTURE_REGISTRY:
        raise ValueError(f"Architecture '{name}' not found.\n"
                         f"Available architectures: {list_architectures()}\n"
                         f"Register new architectures with @register_architecture decorator.")
    return _ARCHITECTURE_REGISTRY[name]


def list_architectures() -> List[str]:
    return sorted(_ARCHITECTURE_REGISTRY.keys())


def is_architecture_available(name: str) -> bool:
    return name in _ARCHITECTURE_REGISTRY


def unregister_architecture(name: str) -> None:
    if name not in _ARCHITECTURE_REGISTRY:
        raise ValueError(f"Architecture '{name}' not registered.")
    del _ARCHITECTURE_REGISTRY[name]


def get_architecture_info() -> Dict[str, Dict[str, str]]:
    return {name: {"name": name, "module": fn.__module__,
                   "doc": fn.__doc__.strip() if fn.__doc__ else "No documentation"}
            for name, fn in _ARCHITECTURE_REGISTRY.items()}


__all__ = ["register_architecture", "get_architecture_builder", "list_architectures",
           "is_architecture_available", "unregister_architecture", "get_architecture_info"]
