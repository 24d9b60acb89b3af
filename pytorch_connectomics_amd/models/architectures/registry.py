"""Architecture registry -- same contract as the reference's connectomics/models/architectures/registry.py:14-120 (the six public
functions, UserWarning on overwrite, ValueError listing the available names on a miss), as one table object."""
from __future__ import annotations

import warnings
from typing import Callable, Dict, List


class _BuilderTable(dict):
    """name -> builder(cfg) -> nn.Module."""

    def require(self, name: str, miss: str) -> Callable:
        try:
            return self[name]
        except KeyError:
            raise ValueError(miss.format(name=name, names=sorted(self))) from None

    def describe(self) -> Dict[str, Dict[str, str]]:
        return {name: dict(name=name, module=fn.__module__, doc=(fn.__doc__ or "").strip() or "No documentation")
                for name, fn in self.items()}


_ARCHITECTURE_REGISTRY = _BuilderTable()
_MISS_LOOKUP = ("Architecture '{name}' not found.\nAvailable architectures: {names}\n"
                "Register new architectures with @register_architecture decorator.")
_MISS_REMOVE = "Architecture '{name}' not registered."


def register_architecture(name: str):
    """Decorator: ``@register_architecture("my_model") def build(cfg) -> nn.Module``."""
    def decorator(builder_fn: Callable) -> Callable:
        replaced = name in _ARCHITECTURE_REGISTRY
        _ARCHITECTURE_REGISTRY[name] = builder_fn                     # an overwritten name keeps its place in the table
        if replaced:
            warnings.warn(f"Architecture '{name}' already registered. Overwriting previous registration.", UserWarning)
        return builder_fn
    return decorator


def get_architecture_builder(name: str) -> Callable:
    return _ARCHITECTURE_REGISTRY.require(name, _MISS_LOOKUP)


def list_architectures() -> List[str]:
    return sorted(_ARCHITECTURE_REGISTRY)


def is_architecture_available(name: str) -> bool:
    return name in _ARCHITECTURE_REGISTRY


def unregister_architecture(name: str) -> None:
    _ARCHITECTURE_REGISTRY.require(name, _MISS_REMOVE)
    del _ARCHITECTURE_REGISTRY[name]


def get_architecture_info() -> Dict[str, Dict[str, str]]:
    return _ARCHITECTURE_REGISTRY.describe()


__all__ = ["register_architecture", "get_architecture_builder", "list_architectures",
           "is_architecture_available", "unregister_architecture", "get_architecture_info"]
