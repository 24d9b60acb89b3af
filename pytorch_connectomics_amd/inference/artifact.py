"""Raw prediction artifact (reference: connectomics/inference/artifact.py:15-260).

Canonical per-volume layout `(C, Z, Y, X)`, dataset `main`, gzip, metadata as dataset attributes (tuples / lists /
dicts JSON-encoded, `None` skipped).  The container is real HDF5: h5py when importable, else utils/h5lite.py (a ctypes
layer over a C shim on the image's libhdf5; the files validate with h5dump and carry h5py's attribute types), so the
reference's decoders read what this engine writes.  Only when neither exists is the same content written as
`<path>.npy` + `<path>.attrs.json` (a warning says so).
"""
from __future__ import annotations

import json
from dataclasses import asdict, dataclass, field
from pathlib import Path
from typing import Any, Callable, Mapping, Optional, Sequence, Tuple

import numpy as np

from ..utils.h5lite import get_h5_backend
from ..utils.model_outputs import get_inference_select_channel

h5py = get_h5_backend()                # h5py, or the in-repo h5lite over libhdf5, or None


@dataclass(frozen=True)
class PredictionArtifactMetadata:
    """artifact.py:15-38 (same field names, so attrs interchange)."""
    kind: str = "raw_prediction"
    layout: str = "CZYX"
    image_path: Optional[str] = None
    checkpoint_path: Optional[str] = None
    output_head: Optional[str] = None
    input_shape: Optional[Tuple[int, ...]] = None
    final_shape: Optional[Tuple[int, ...]] = None
    crop_pad: Optional[Tuple[Tuple[int, int], ...]] = None
    transpose: Optional[Tuple[int, ...]] = None
    model_architecture: Optional[str] = None
    model_output_identity: Optional[str] = None
    decode_after_inference: Optional[bool] = None
    chunk_shape: Optional[Tuple[int, ...]] = None
    halo: Optional[Tuple[int, ...]] = None
    channel_order: Optional[Tuple[str, ...]] = None
    activation: Optional[str] = None
    intensity_scale: Optional[float] = None
    intensity_dtype: Optional[str] = None
    extra: Mapping[str, Any] = field(default_factory=dict)


def _cfg_get(obj: Any, path: str, default: Any = None) -> Any:
    node = obj
    for part in path.split("."):
        if node is None:
            return default
        node = node.get(part, default) if isinstance(node, Mapping) else getattr(node, part, default)
    return node


def _tuple_or_none(value: Optional[Sequence[Any]]):
    if value is None or len(value) == 0:
        return None
    return tuple(int(v) for v in value)


def build_prediction_artifact_metadata(cfg: Any, *, image_path=None, checkpoint_path=None, output_head=None,
                                       input_shape=None, final_shape=None, crop_pad=None, chunk_shape=None, halo=None,
                                       intensity_scale=None, intensity_dtype=None, extra=None) -> PredictionArtifactMetadata:
    """artifact.py:78-121: intensity scale / dtype default to the enabled prediction transform's; the output identity
    string is `head=<h>` (or `primary_head=<p>`) + `select_channel=<sel>` joined by ';' (artifact.py:55-71)."""
    tc = _cfg_get(cfg, "inference.prediction_transform")
    if bool(getattr(tc, "enabled", False)):
        if intensity_scale is None:
            intensity_scale = float(getattr(tc, "intensity_scale", -1.0))
        if intensity_dtype is None:
            intensity_dtype = getattr(tc, "intensity_dtype", None)
    ident = []
    if output_head:
        ident.append(f"head={output_head}")
    elif _cfg_get(cfg, "model.primary_head"):
        ident.append(f"primary_head={_cfg_get(cfg, 'model.primary_head')}")
    sel = get_inference_select_channel(cfg)
    if sel is not None:
        ident.append(f"select_channel={sel}")
    return PredictionArtifactMetadata(
        image_path=None if image_path is None else str(image_path),
        checkpoint_path=None if checkpoint_path is None else str(checkpoint_path),
        output_head=output_head, input_shape=_tuple_or_none(input_shape), final_shape=_tuple_or_none(final_shape),
        crop_pad=tuple((int(p[0]), int(p[1])) for p in crop_pad) if crop_pad is not None else None,
        transpose=_tuple_or_none(_cfg_get(cfg, "data.data_transform.val_transpose")),
        model_architecture=_cfg_get(cfg, "model.arch.type"),
        model_output_identity=";".join(ident) if ident else None,
        decode_after_inference=bool(_cfg_get(cfg, "decoding.enabled", True)),
        chunk_shape=_tuple_or_none(chunk_shape), halo=_tuple_or_none(halo),
        intensity_scale=intensity_scale, intensity_dtype=intensity_dtype, extra=extra or {})


def _json_attr(value: Any) -> Any:
    if value is None or isinstance(value, (str, int, float, bool)):
        return value
    if isinstance(value, (tuple, list, dict)):
        return json.dumps(value)
    return str(value)


def artifact_attrs(metadata: PredictionArtifactMetadata) -> dict:
    """The attribute dictionary `write_prediction_artifact_attrs` (artifact.py:133-139) puts on the dataset."""
    attrs = asdict(metadata)
    extra = attrs.pop("extra", {}) or {}
    return {k: _json_attr(v) for k, v in {**attrs, **dict(extra)}.items() if v is not None}


def write_prediction_artifact_attrs(dataset: Any, metadata: PredictionArtifactMetadata) -> None:
    """The metadata attributes on an open HDF5 dataset (h5py or the in-repo shim: anything with `.attrs[key] = value`)."""
    for key, value in artifact_attrs(metadata).items():
        dataset.attrs[key] = value


def write_prediction_artifact(path, data: Optional[np.ndarray] = None, *, metadata: Optional[PredictionArtifactMetadata] = None,
                              dataset: str = "main", compression: Optional[str] = "gzip", shape=None, dtype=None,
                              chunks=None, writer: Optional[Callable[[Any], None]] = None) -> Path:
    """artifact.py:141-203.  `data` is CZYX; streaming mode (`data=None`, `shape`, `dtype`, `writer(dset)`) is kept: the
    writer receives an array-like it fills in place."""
    arr = None if data is None else np.asarray(data)
    if arr is None:
        if shape is None or dtype is None:
            raise ValueError("Streaming prediction artifacts require shape and dtype.")
        a_shape, a_dtype = tuple(int(v) for v in shape), np.dtype(dtype)
    else:
        if arr.ndim != 4:
            raise ValueError(f"Prediction artifacts must use CZYX layout, got shape {arr.shape}")
        a_shape, a_dtype = tuple(int(v) for v in arr.shape), arr.dtype
    if len(a_shape) != 4:
        raise ValueError(f"Prediction artifacts must use CZYX layout, got shape {a_shape}")
    md = metadata or PredictionArtifactMetadata(final_shape=tuple(a_shape[-3:]), intensity_dtype=str(a_dtype))
    out = Path(path)
    out.parent.mkdir(parents=True, exist_ok=True)
    if h5py is not None:
        with h5py.File(out, "w") as handle:
            dset = (handle.create_dataset(dataset, shape=a_shape, dtype=a_dtype, chunks=chunks, compression=compression)
                    if arr is None else handle.create_dataset(dataset, data=arr, chunks=chunks, compression=compression))
            for k, v in artifact_attrs(md).items():
                dset.attrs[k] = v
            if writer is not None:
                writer(dset)
        return out
    import warnings
    warnings.warn("no HDF5 backend (h5py / libpytc_h5.so): prediction artifact written as .npy + .attrs.json")
    npy = Path(str(out) + ".npy")
    if arr is None:
        mm = np.lib.format.open_memmap(npy, mode="w+", dtype=a_dtype, shape=a_shape)
        if writer is not None:
            writer(mm)
        mm.flush()
        del mm
    else:
        np.save(npy, arr)
    Path(str(out) + ".attrs.json").write_text(json.dumps({"dataset": dataset, "attrs": artifact_attrs(md)}, indent=1))
    return out


def read_prediction_artifact(path, *, dataset: str = "main", return_metadata: bool = False):
    """artifact.py:206-240 counterpart: array (CZYX) and, optionally, the attribute dictionary."""
    p = Path(path)
    if h5py is not None and p.exists():
        with h5py.File(p, "r") as handle:
            arr = handle[dataset][...]
            attrs = {k: (v.decode() if isinstance(v, bytes) else v) for k, v in handle[dataset].attrs.items()}
    else:
        arr = np.load(str(p) + ".npy")
        attrs = json.loads(Path(str(p) + ".attrs.json").read_text())["attrs"]
    return (arr, attrs) if return_metadata else arr


__all__ = ["PredictionArtifactMetadata", "build_prediction_artifact_metadata", "artifact_attrs", "write_prediction_artifact_attrs",
           "write_prediction_artifact", "read_prediction_artifact"]
