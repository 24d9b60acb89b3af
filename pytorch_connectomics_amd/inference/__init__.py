"""connectomics.inference counterpart (sliding-window engine on the device).  Same public names as the reference package
(connectomics/inference/__init__.py:1-45)."""
from .artifact import (PredictionArtifactMetadata, build_prediction_artifact_metadata, read_prediction_artifact,
                       write_prediction_artifact, write_prediction_artifact_attrs)
from .chunked import is_chunked_inference_enabled, is_external_chunk_sharding_enabled, run_chunked_prediction_inference
from .lazy import lazy_predict_region, lazy_predict_volume
from .manager import InferenceManager
from .output import apply_prediction_transform, apply_storage_dtype_transform, resolve_output_filenames, write_outputs
from .stage import run_prediction_inference
from .tta import TTAPredictor
from .tta_affinity import invert_view
from .tta_ensemble import TTAEnsembleAccumulator
from .window import (EagerSlidingWindowEngine, build_sliding_inferer, compute_importance_map,
                     compute_scan_interval, dense_patch_slices, build_sliding_importance_map, is_2d_inference_mode,
                     normalize_weighted_accumulator, resolve_inferer_overlap, resolve_inferer_roi_size)

__all__ = ["InferenceManager", "TTAPredictor", "TTAEnsembleAccumulator", "invert_view", "EagerSlidingWindowEngine", "build_sliding_inferer", "compute_importance_map", "compute_scan_interval",
           "dense_patch_slices", "build_sliding_importance_map", "normalize_weighted_accumulator", "run_prediction_inference",
           "lazy_predict_region", "lazy_predict_volume", "run_chunked_prediction_inference", "is_chunked_inference_enabled",
           "PredictionArtifactMetadata", "build_prediction_artifact_metadata", "read_prediction_artifact",
           "write_prediction_artifact", "write_prediction_artifact_attrs", "apply_prediction_transform",
           "apply_storage_dtype_transform", "resolve_output_filenames", "write_outputs", "is_2d_inference_mode", "is_external_chunk_sharding_enabled", "resolve_inferer_overlap",
           "resolve_inferer_roi_size"]
