"""Prediction stage: predict one volume and (optionally) write the canonical raw-prediction artifact -- counterpart of the
reference's connectomics/inference/stage.py (:16-87, `run_prediction_inference`).

MI355X design: the reference moves the float32 prediction to the host first and applies the semantic transform
(intensity scale / dtype) and the storage-dtype transform there with numpy.  Here both run on the device on the
HBM-resident prediction, so the ONE device -> host copy moves the stored representation (uint8 / float16 when configured:
4x / 2x fewer PCIe bytes), and only when an artifact is requested.  The returned tensor is the untransformed prediction
on the device, as in the reference."""
from __future__ import annotations

from pathlib import Path
from typing import Any, Optional, Sequence

import numpy as np
import torch

from .artifact import build_prediction_artifact_metadata, write_prediction_artifact
from .output import apply_prediction_transform, apply_storage_dtype_transform

__all__ = ["run_prediction_inference"]


def _prediction_tensor_to_czyx(predictions: torch.Tensor) -> torch.Tensor:
    """(1, C, Z, Y, X) or (C, Z, Y, X) -> CZYX, still on the device (stage.py:16-28, same messages)."""
    shape = tuple(predictions.shape)
    if len(shape) == 5 and shape[0] != 1:
        raise ValueError(f"run_prediction_inference can write one artifact per call; got batch size {shape[0]}.")
    if len(shape) not in (4, 5):
        raise ValueError(f"Prediction artifact expects CZYX data, got shape {shape}.")
    return predictions.detach().reshape(shape[-4:])


def _normalize_compression(value: Any) -> Optional[str]:
    return str(value) if value not in (None, "", "none") else None


def _stored_representation(cfg, czyx: torch.Tensor) -> np.ndarray:
    """Semantic transform (intensity scale / dtype) and storage dtype ON THE DEVICE, then the one device -> host copy."""
    stored = apply_storage_dtype_transform(cfg, apply_prediction_transform(cfg, czyx))
    if isinstance(stored, torch.Tensor):
        if stored.is_cuda:
            torch.cuda.current_stream(stored.device).synchronize()
        stored = stored.cpu().numpy()
    return np.asarray(stored)


def run_prediction_inference(manager, images: torch.Tensor, *, mask: Optional[torch.Tensor] = None,
                             mask_align_to_image: bool = False, requested_head: Optional[str] = None,
                             output_path=None, image_path: Optional[str] = None, checkpoint_path=None,
                             input_shape: Optional[Sequence[int]] = None,
                             crop_pad: Optional[Sequence[Sequence[int]]] = None) -> torch.Tensor:
    """Model prediction without decoding or evaluation; with `output_path` the single-volume prediction is also written
    as the raw-prediction artifact (HDF5 `main` dataset + metadata attributes).  On the contributing ranks of a sharded
    run (`manager.should_skip_postprocess_on_rank()`) nothing is written and the empty tensor is returned."""
    predictions = manager.predict_with_tta(images, mask=mask, mask_align_to_image=mask_align_to_image,
                                           requested_head=requested_head)
    contributes_only = getattr(manager, "should_skip_postprocess_on_rank", lambda: False)
    if output_path is None or contributes_only():
        return predictions
    cfg = manager.cfg
    stored = _stored_representation(cfg, _prediction_tensor_to_czyx(predictions))
    compression = _normalize_compression(getattr(cfg.inference, "save_compression", "gzip"))
    described = build_prediction_artifact_metadata(
        cfg, image_path=image_path, checkpoint_path=None if checkpoint_path is None else str(checkpoint_path),
        output_head=requested_head, input_shape=input_shape, final_shape=stored.shape[-3:], crop_pad=crop_pad,
        intensity_dtype=str(stored.dtype), extra={"compression": str(compression)})
    write_prediction_artifact(output_path, stored, metadata=described, compression=compression)
    return predictions
