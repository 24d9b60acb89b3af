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


# ------------------------------------------------------------------------------------------------ per-volume output files
# file stems that say nothing about the volume ("img.h5", "data.zarr/main"), and container directories to climb out of
_PLAIN_STEMS = frozenset({"img", "image", "raw", "em", "main", "data"})
_CONTAINER_SUFFIXES = (".zarr", ".n5", ".ome.zarr")


def volume_stem_from_path(path) -> str:
    """The per-volume directory name of an image path (reference runtime/output_naming.py:54-94): the file stem when it is
    informative; else the nearest ancestor directory that is neither a container (`*.zarr`, `*.n5`) nor itself uninformative
    (`/data/seed101/data.zarr/img` -> `seed101`, `/data/seed101/img.h5` -> `seed101`); else "volume"."""
    from pathlib import Path
    here = Path(str(path))
    stem = here.stem.strip()
    if stem and stem.lower() not in _PLAIN_STEMS:
        return stem
    for ancestor in here.parents:
        name = ancestor.name.strip()
        if not name:
            break
        low = name.lower()
        if low in _PLAIN_STEMS or any(low.endswith(sfx) for sfx in _CONTAINER_SUFFIXES):
            continue
        return name
    return "volume"


def resolve_output_filenames(cfg: Any, batch, global_step: int = 0) -> list:
    """One directory stem per batch item from the batch's metadata (reference output.py:19-83): the file names of
    `image_meta_dict` (a list of per-item dicts or one dict of lists, key `filename_or_obj`), else path-like `image` entries;
    items without a name become `volume_<global_step>_<index>`."""
    import os
    images = batch.get("image")
    if images is None or isinstance(images, (str, os.PathLike)):
        count = 1
    else:
        count = len(images) if isinstance(images, (list, tuple)) else int(images.shape[0])
    names: list = []
    meta = batch.get("image_meta_dict")
    if isinstance(meta, list):
        names = [item["filename_or_obj"] for item in meta if isinstance(item, dict) and item.get("filename_or_obj") is not None]
        count = max(count, len(names))
    elif isinstance(meta, dict):
        listed = meta.get("filename_or_obj")
        if isinstance(listed, (list, tuple)):
            names = [name for name in listed if name is not None]
        elif listed is not None:
            names = [listed]
        count = max(count, len(names))
    if not names:
        if isinstance(images, (str, os.PathLike)):
            names = [str(images)]
        elif isinstance(images, (list, tuple)):
            names = [str(item) for item in images if isinstance(item, (str, os.PathLike))]
            count = max(count, len(names))
    return [volume_stem_from_path(names[i]) if i < len(names) and names[i] else f"volume_{global_step}_{i}" for i in range(count)]


def write_outputs(cfg: Any, predictions, filenames, suffix: str = "prediction.h5", mode: str = "test", batch_meta: Any = None) -> None:
    """Persist a batch of predictions as `<inference.save_path>/<volume stem>/<suffix stem>.h5`, dataset `main`, after the storage
    dtype transform (reference output.py:252-356).  The MI355X package writes the HDF5 backend (through h5py or the in-repo libhdf5
    shim); the reference's other backends (tiff / nii.gz / png through its data.io package) and its nnU-Net restore step belong to
    the data pipeline, which stays with the reference: they are refused by name instead of being skipped silently."""
    from pathlib import Path
    inf = getattr(cfg, "inference", None)
    root = getattr(inf, "save_path", None) if inf is not None else None
    if inf is None or not root:
        return
    data_cfg = getattr(cfg, "data", None)
    pre = getattr(data_cfg, "nnunet_preprocessing", None)
    if pre is not None and getattr(pre, "enabled", False) and getattr(pre, "restore_to_input_space", False) and \
            suffix in ("prediction.h5", "prediction"):
        raise NotImplementedError("write_outputs: restoring predictions to the nnU-Net input space is part of the reference's data "
                                  "pipeline (data.processing.nnunet_preprocess), not of pytorch_connectomics_amd")
    backend = str(getattr(inf, "save_backend", "h5")).lower()
    if backend != "h5":
        raise NotImplementedError(f"write_outputs: inference.save_backend={backend!r} is written by the reference's data.io package; "
                                  "pytorch_connectomics_amd writes 'h5'")
    from ..utils.h5lite import get_h5_backend
    h5 = get_h5_backend()
    if h5 is None:
        raise RuntimeError("write_outputs needs h5py or the in-repo libpytc_h5.so (csrc/host/h5io.c); neither loads")
    preds = np.asarray(predictions.detach().cpu().numpy() if hasattr(predictions, "detach") else predictions)
    names = list(filenames)
    if preds.ndim < 4 and not (preds.ndim == 3 and names and preds.shape[0] == len(names)):
        preds = preds[np.newaxis]                                  # one volume without a batch axis
    if len(names) != preds.shape[0]:
        logger.warning("write_outputs - filename count (%d) does not match batch size (%d). Using first %d filenames.", len(names),
                       preds.shape[0], min(len(names), preds.shape[0]))
    name = suffix[:-len(".nii.gz")] if suffix.endswith(".nii.gz") else (Path(suffix).stem if "." in Path(suffix).name else suffix)
    for index in range(preds.shape[0]):
        if index >= len(names):
            logger.warning("write_outputs - no filename for batch index %d, skipping", index)
            continue
        sample = apply_storage_dtype_transform(cfg, np.squeeze(preds[index]))
        folder = Path(root) / str(names[index])
        folder.mkdir(parents=True, exist_ok=True)
        with h5.File(str(folder / f"{name}.h5"), "w") as fh:
            fh.create_dataset("main", data=np.ascontiguousarray(sample), compression="gzip")
        logger.info("Saved HDF5: %s/%s.h5", names[index], name)


__all__ = ["apply_prediction_transform", "apply_storage_dtype_transform", "resolve_output_filenames", "write_outputs",
           "volume_stem_from_path"]
