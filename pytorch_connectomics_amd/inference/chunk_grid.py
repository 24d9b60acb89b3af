"""Chunked-inference geometry helpers under the reference's module name (connectomics/inference/chunk_grid.py:22-111):
the crop helpers live in `.crop` (normalize_crop_pad, resolve_selected_affinity_offsets, resolve_global_prediction_crop);
this module adds the chunk-shape / HDF5-layout / mode resolvers.  Host integer logic, pinned by tests/golden/crops.json and
tests/test_host_chunked.py."""
from __future__ import annotations

from typing import Any, Sequence

from .crop import normalize_crop_pad, resolve_global_prediction_crop, resolve_selected_affinity_offsets


_HDF5_BACKENDS = ("h5", "hdf5")
_OUTPUT_MODES = ("decoded", "raw_prediction")
_H5_CHUNK_EDGE = 64


def _chunking(cfg: Any):
    return getattr(getattr(cfg, "inference", None), "chunking", None)


def validate_chunked_output_format(cfg: Any) -> None:
    """chunk_grid.py:78-86: chunked inference streams ONE HDF5 output; any other save backend is a configuration error."""
    chosen = str(getattr(getattr(cfg, "inference", None), "save_backend", _HDF5_BACKENDS[0])).lower()
    if chosen in _HDF5_BACKENDS:
        return
    raise ValueError(f"Chunked inference writes a single streamed HDF5 output only; unsupported inference.save_backend={chosen!r}.")


def resolve_chunk_shape(cfg: Any, final_shape: Sequence[int]) -> tuple[int, int, int]:
    """chunk_grid.py:89-97: `chunking.chunk_size`, `axes: z` keeps full YX, every axis clipped to the (cropped) volume."""
    section = _chunking(cfg)
    wanted = [int(v) for v in (getattr(section, "chunk_size", None) or ())]
    if not wanted:
        raise ValueError("inference.chunking.chunk_size must be set for chunked inference")
    extent = [int(v) for v in final_shape[:3]]
    split = str(getattr(section, "axes", "all")).lower()
    if split not in ("all", "z"):
        raise ValueError("inference.chunking.axes must be 'all' or 'z'")
    used = wanted[:1] if split == "z" else wanted[:3]          # like the reference: z slabs read the first entry only
    if len(used) < (1 if split == "z" else 3) or min(used) <= 0:
        # (the reference lets a zero / negative / short chunk_size through and fails later with an empty grid or an IndexError)
        raise ValueError(f"inference.chunking.chunk_size must be 3 positive ints, got {wanted}")
    if split == "z":
        return (used[0], extent[1], extent[2])
    return tuple(min(w, e) for w, e in zip(used, extent))


def resolve_h5_spatial_chunks(spatial_shape: Sequence[int]) -> tuple[int, int, int]:
    """chunk_grid.py:100-102: HDF5 chunk = min(64, extent) per spatial axis (the channel axis is chunked whole)."""
    z, y, x = (min(_H5_CHUNK_EDGE, int(v)) for v in spatial_shape[:3])
    return (z, y, x)


def resolve_chunk_output_mode(cfg: Any) -> str:
    """chunk_grid.py:105-110."""
    mode = str(getattr(cfg.inference.chunking, "output_mode", _OUTPUT_MODES[0])).lower()
    if mode in _OUTPUT_MODES:
        return mode
    raise ValueError("inference.chunking.output_mode must be 'decoded' or 'raw_prediction'.")


__all__ = ["normalize_crop_pad", "resolve_selected_affinity_offsets", "resolve_global_prediction_crop",
           "validate_chunked_output_format", "resolve_chunk_shape", "resolve_h5_spatial_chunks", "resolve_chunk_output_mode"]
