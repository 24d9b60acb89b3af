"""InferenceManager: wires the device sliding-window engine and TTA (contract of the reference's
connectomics/inference/manager.py:24-119: constructor, the two predict entries, the three sharding queries)."""
from __future__ import annotations

import logging
from typing import Dict, List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from .tta import TTAPredictor
from .window import build_sliding_inferer, is_2d_inference_mode

logger = logging.getLogger(__name__)


def _section(cfg, *names):
    """cfg.a.b.c with None for anything missing on the way."""
    for name in names:
        cfg = getattr(cfg, name, None)
    return cfg


class InferenceManager:
    def __init__(self, cfg, model: nn.Module, forward_fn: callable):
        self.cfg, self.model, self.forward_fn = cfg, model, forward_fn
        self.sliding_inferer = None
        if not is_2d_inference_mode(cfg):
            self.sliding_inferer = build_sliding_inferer(cfg)
        else:
            logger.warning("Sliding-window inference disabled for 2D models with do_2d=True. "
                           "Using direct inference instead.")
        self.tta = TTAPredictor(cfg=cfg, sliding_inferer=self.sliding_inferer, forward_fn=forward_fn, model=model)

    # ---- prediction: everything happens in TTAPredictor (device engine, device ensembles)
    def predict_with_tta(self, images: torch.Tensor, mask: Optional[torch.Tensor] = None,
                         mask_align_to_image: bool = False, requested_head: Optional[str] = None) -> torch.Tensor:
        return self.tta.predict(images, mask=mask, mask_align_to_image=mask_align_to_image, requested_head=requested_head)

    def predict_named_heads_with_tta(self, images: torch.Tensor, heads: List[str], mask=None,
                                     mask_align_to_image: bool = False) -> Dict[str, torch.Tensor]:
        out: Dict[str, torch.Tensor] = {}
        for head in heads:
            out[head] = self.predict_with_tta(images, mask, mask_align_to_image, head)
        return out

    # ---- which rank holds the result of a sharded run
    def is_distributed_tta_sharding_enabled(self) -> bool:
        return self.tta.is_distributed_sharding_enabled()

    def is_distributed_window_sharding_enabled(self) -> bool:
        """Lazy (zarr / HDF5) volumes + `sliding_window.distributed_sharding` + an initialised process group of > 1 ranks."""
        window = _section(self.cfg, "inference", "sliding_window")
        loader = _section(self.cfg, "data", "dataloader")
        if window is None or not getattr(window, "distributed_sharding", False):
            return False
        if not (getattr(loader, "use_lazy_zarr", False) or getattr(loader, "use_lazy_h5", False)):
            return False
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def should_skip_postprocess_on_rank(self) -> bool:
        if self.tta.should_skip_postprocess_on_rank():
            return True
        return self.is_distributed_window_sharding_enabled() and dist.get_rank() != 0


__all__ = ["InferenceManager"]
