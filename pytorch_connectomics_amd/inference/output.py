"""Prediction / storage dtype transforms (reference: connectomics/inference/output.py:150-243).

`apply_prediction_transform` (semantic: `inference.prediction_transform.{enabled,intensity_scale,intensity_dtype}`)
and `apply_storage_dtype_transform` (`inference.save_dtype`) keep the reference's config keys, order of operations and
numpy semantics (scale in fp32 -> clip to the integer range -> truncating cast; unknown dtype names warn and keep the
data).  Device tensors are transformed by one HIP kernel (`pytc_scale_cast`) BEFORE the device->host copy, so a uint8
artifact crosses PCIe at a quarter of the fp32 bytes; numpy arrays (results already on the host, e.g. stitched chunk
files) go through numpy exactly like the reference.
"""
from __future__ import annotations

import logging
from typing import Any, Optional, Union

import numpy as np
import torch

logger = logging.getLogger(__name__)

_DTYPE_MAP = {"uint8": np.uint8, "int8": np.int8, "uint16": np.uint16, "int16": np.int16, "uint32": np.uint32,
              "int32": np.int32, "float16": np.float16, "float32": np.float32, "float64": np.float64}
_DEVICE_TARGETS = ("uint8", "int8", "uint16", "int16", "int32", "float16", "float32")

ArrayLike = Union[np.ndarray, torch.Tensor]


def _convert_intensity_dtype(data: ArrayLike, target: Optional[str], *, config_name: str, scale: float = 1.0) -> ArrayLike:
    """output.py:147-185 (+ the scale of :195-201 folded in for the device path)."""
    if target is None:
        if scale != 1.0:
            data = data * float(scale)
        return data
    if target not in _DTYPE_MAP:
        logger.warning("Unknown dtype '%s' in %s. Supported: %s. Keeping current dtype.", target, config_name,
                       list(_DTYPE_MAP))
        return data * float(scale) if scale != 1.0 else data
    if isinstance(data, torch.Tensor) and data.is_cuda:
        if target not in _DEVICE_TARGETS:
            raise NotImplementedError(f"{config_name}: dtype '{target}' has no device cast; convert on the host array")
        from .. import hip_ops as ops
        return ops.scale_cast(data.float().contiguous(), scale=scale, target=target)
    arr = data.detach().cpu().numpy() if isinstance(data, torch.Tensor) else data
    if scale != 1.0:
        arr = arr.astype(np.float32, copy=False) * np.float32(scale)
    tdt = _DTYPE_MAP[target]
    if np.issubdtype(tdt, np.integer):
        info = np.iinfo(tdt)
        arr = np.clip(arr, info.min, info.max)
    return arr.astype(tdt, copy=False)


def _apply_intensity_transform(data: ArrayLike, *, intensity_scale, intensity_dtype, config_name: str) -> ArrayLike:
    """output.py:188-211: scale >= 0 -> fp32 (* scale when != 1); negative scale = disabled; then the dtype cast."""
    scale = 1.0
    if intensity_scale is not None and intensity_scale >= 0:
        if isinstance(data, np.ndarray):
            data = data.astype(np.float32, copy=False)
        elif data.dtype != torch.float32:
            data = data.float()
        scale = float(intensity_scale)
    else:
        logger.info("Intensity scaling disabled for %s (scale=%s < 0), keeping raw predictions", config_name,
                    intensity_scale)
    return _convert_intensity_dtype(data, intensity_dtype, config_name=config_name, scale=scale)


def apply_prediction_transform(cfg: Any, data: ArrayLike) -> ArrayLike:
    """output.py:213-227."""
    inf = getattr(cfg, "inference", None)
    if inf is None:
        return data
    tc = getattr(inf, "prediction_transform", None)
    if tc is None or not getattr(tc, "enabled", False):
        return data
    return _apply_intensity_transform(data, intensity_scale=getattr(tc, "intensity_scale", -1.0),
                                      intensity_dtype=getattr(tc, "intensity_dtype", None),
                                      config_name="inference.prediction_transform")


def apply_storage_dtype_transform(cfg: Any, data: ArrayLike) -> ArrayLike:
    """output.py:230-243: save/cache-only dtype conversion; skipped when `inference.save_dtype` is unset."""
    inf = getattr(cfg, "inference", None)
    if inf is None:
        return data
    sd = getattr(inf, "save_dtype", None)
    if sd is None:
        return data
    return _convert_intensity_dtype(data, sd, config_name="inference.save_dtype")


__all__ = ["apply_prediction_transform", "apply_storage_dtype_transform"]
