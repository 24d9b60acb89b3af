"""build_model(cfg): look up cfg.model.arch.type in the registry and call the builder
(contract of the reference's connectomics/models/build.py:24-72)."""
from __future__ import annotations

import logging

from .architectures import get_architecture_builder

logger = logging.getLogger(__name__)


def build_model(cfg):
    model_arch = cfg.model.arch.type
    builder = get_architecture_builder(model_arch)   # ValueError lists the registered names
    model = builder(cfg)
    logger.info("Model: %s (architecture: %s)", model.__class__.__name__, model_arch)
    if hasattr(model, "get_model_info"):
        info = model.get_model_info()
        logger.info("  Parameters: %s", f"{info['parameters']:,}")
        logger.info("  Trainable: %s", f"{info['trainable_parameters']:,}")
        logger.info("  Deep Supervision: %s", info["deep_supervision"])
        if info["deep_supervision"]:
            logger.info("  Output Scales: %s", info["output_scales"])
    return model


__all__ = ["build_model"]
