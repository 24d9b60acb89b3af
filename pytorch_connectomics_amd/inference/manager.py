"""InferenceManager: wires the device sliding-window engine and TTA (contract of the reference's
connectomics/inference/manager.py:24-119)."""
from __future__ import annotations

import logging
from typing import Optional

import torch
import torch.nn as nn

from .tta import TTAPredictor
from .window import build_sliding_inferer, is_2d_inference_mode

logger = logging.getLogger(__name__)


class InferenceManager:
    def __init__(self, cfg, model: nn.Module, forward_fn: callable):
        self.cfg = cfg
        self.model = model
        self.forward_fn = forward_fn
        if is_2d_inference_mode(cfg):
            self.sliding_inferer = None
            logger.warning("Sliding-window inference disabled for 2D models with do_2d=True. "
                           "Using direct inference instead.")
        else:
            self.sliding_inferer = build_sliding_inferer(cfg)
        self.tta = TTAPredictor(cfg=cfg, sliding_inferer=self.sliding_inferer, forward_fn=self.forward_fn,
                                model=model)

    def predict_with_tta(self, images: torch.Tensor, mask: Optional[torch.Tensor] = None,
                         mask_align_to_image: bool = False, requested_head: Optional[str] = None) -> torch.Tensor:
        return self.tta.predict(images, mask=mask, mask_align_to_image=mask_align_to_image,
                                requested_head=requested_head)

    def predict_named_heads_with_tta(self, images: torch.Tensor, heads: list[str], mask=None,
                                     mask_align_to_image: bool = False) -> dict[str, torch.Tensor]:
        return {h: self.predict_with_tta(images, mask=mask, mask_align_to_image=mask_align_to_image,
                                         requested_head=h) for h in heads}

    def is_distributed_tta_sharding_enabled(self) -> bool:
        return self.tta.is_distributed_sharding_enabled()

    def is_distributed_window_sharding_enabled(self) -> bool:
        sw = getattr(getattr(self.cfg, "inference", None), "sliding_window", None)
        if sw is None:
            return False
        dl = getattr(getattr(self.cfg, "data", None), "dataloader", None)
        lazy = bool(getattr(dl, "use_lazy_zarr", False) or getattr(dl, "use_lazy_h5", False))
        is_dist = torch.distributed.is_available() and torch.distributed.is_initialized()
        world = torch.distributed.get_world_size() if is_dist else 1
        return bool(lazy and getattr(sw, "distributed_sharding", False) and is_dist and world > 1)

    def should_skip_postprocess_on_rank(self) -> bool:
        if self.tta.should_skip_postprocess_on_rank():
            return True
        if self.is_distributed_window_sharding_enabled():
            return torch.distributed.get_rank() != 0
        return False


__all__ = ["InferenceManager"]
