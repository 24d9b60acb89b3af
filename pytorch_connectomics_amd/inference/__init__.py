"""connectomics.inference counterpart (sliding-window engine on the device)."""
from .manager import InferenceManager
from .tta import TTAPredictor
from .window import (EagerSlidingWindowEngine, build_sliding_inferer, compute_importance_map,
                     compute_scan_interval, dense_patch_slices, build_sliding_importance_map,
                     normalize_weighted_accumulator)

__all__ = ["InferenceManager", "TTAPredictor", "EagerSlidingWindowEngine", "build_sliding_inferer", "compute_importance_map", "compute_scan_interval",
           "dense_patch_slices", "build_sliding_importance_map", "normalize_weighted_accumulator"]
