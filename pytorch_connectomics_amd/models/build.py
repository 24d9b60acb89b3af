"""build_model(cfg): look up cfg.model.arch.type in the registry and call the builder
(contract of the reference's connectomics/models/build.py:24-72: same log lines, the registry's ValueError on a miss)."""
from __future__ import annotations

import logging

from .architectures import get_architecture_builder

logger = logging.getLogger(__name__)

# get_model_info() key -> log label, in the order the reference prints them; the last one only with deep supervision
_INFO_LINES = (("parameters", "Parameters", True), ("trainable_parameters", "Trainable", True),
               ("deep_supervision", "Deep Supervision", False), ("output_scales", "Output Scales", False))


def _describe(model, arch: str):
    yield f"Model: {type(model).__name__} (architecture: {arch})"
    info = model.get_model_info() if hasattr(model, "get_model_info") else None
    for key, label, thousands in _INFO_LINES if info else ():
        if key == "output_scales" and not info["deep_supervision"]:
            continue
        yield f"  {label}: {info[key]:,}" if thousands else f"  {label}: {info[key]}"


def build_model(cfg):
    arch = cfg.model.arch.type
    model = get_architecture_builder(arch)(cfg)          # ValueError lists the registered names
    for line in _describe(model, arch):
        logger.info(line)
    return model


__all__ = ["build_model"]
