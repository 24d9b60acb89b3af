"""connectomics.inference counterpart (sliding-window engine on the device).  Same public names as the reference package
(connectomics/inference/__init__.py:1-45)."""
from .artifact import (PredictionArtifactMetadata, build_prediction_artifact_metadata, read_prediction_artifact,
                       write_prediction_artifact)
from .chunked import is_chunked_inference_enabled, run_chunked_prediction_inference
from .lazy import lazy_predict_region, lazy_predict_volume
from .manager import InferenceManager
from .output import apply_prediction_transform, apply_storage_dtype_transform
from .stage import run_prediction_inference
from .tta import TTAPredictor
from .tta_affinity import invert_view
from .tta_ensemble import TTAEnsembleAccumulator
from .window import (EagerSlidingWindowEngine, build_sliding_inferer, compute_importance_map,
                     compute_scan_interval, dense_patch_slices, build_sliding_importance_map,
                     normalize_weighted_accumulator)

__all__ = ["InferenceManager", "TTAPredictor", "TTAEnsembleAccumulator", "invert_view", "EagerSlidingWindowEngine", "build_sliding_inferer", "compute_importance_map", "compute_scan_interval",
           "dense_patch_slices", "build_sliding_importance_map", "normalize_weighted_accumulator", "run_prediction_inference",
           "lazy_predict_region", "lazy_predict_volume", "run_chunked_prediction_inference", "is_chunked_inference_enabled",
           "PredictionArtifactMetadata", "build_prediction_artifact_metadata", "read_prediction_artifact",
           "write_prediction_artifact", "apply_prediction_transform", "apply_storage_dtype_transform"]
