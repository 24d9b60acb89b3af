"""Chunked-inference geometry helpers under the reference's module name (connectomics/inference/chunk_grid.py:22-111):
the crop helpers live in `.crop` (normalize_crop_pad, resolve_selected_affinity_offsets, resolve_global_prediction_crop);
this module adds the chunk-shape / HDF5-layout / mode resolvers.  Host integer logic, pinned by tests/golden/crops.json and
tests/test_host_chunked.py."""
from __future__ import annotations

from typing import Any, Sequence

from .crop import normalize_crop_pad, resolve_global_prediction_crop, resolve_selected_affinity_offsets


def validate_chunked_output_format(cfg: Any) -> None:
    """chunk_grid.py:78-86: chunked inference streams ONE HDF5 output; any other save backend is a configuration error."""
    backend = str(getattr(getattr(cfg, "inference", None), "save_backend", "h5")).lower()
    if backend not in {"h5", "hdf5"}:
        raise ValueError("Chunked inference writes a single streamed HDF5 output only; "
                         f"unsupported inference.save_backend={backend!r}.")


def resolve_chunk_shape(cfg: Any, final_shape: Sequence[int]) -> tuple[int, int, int]:
    """chunk_grid.py:89-97: `chunking.chunk_size`, `axes: z` keeps full YX, every axis clipped to the (cropped) volume."""
    ch = getattr(getattr(cfg, "inference", None), "chunking", None)
    size = getattr(ch, "chunk_size", None)
    if not size:
        raise ValueError("inference.chunking.chunk_size must be set for chunked inference")
    size = [int(v) for v in size]
    if len(size) != 3 or any(v <= 0 for v in size):
        raise ValueError(f"inference.chunking.chunk_size must be 3 positive ints, got {size}")
    axes = str(getattr(ch, "axes", "all")).lower()
    if axes == "z":
        return (size[0], int(final_shape[1]), int(final_shape[2]))
    if axes != "all":
        raise ValueError("inference.chunking.axes must be 'all' or 'z'")
    return tuple(min(size[a], int(final_shape[a])) for a in range(3))


def resolve_h5_spatial_chunks(spatial_shape: Sequence[int]) -> tuple[int, int, int]:
    """chunk_grid.py:100-102: HDF5 chunk = min(64, extent) per spatial axis (the channel axis is chunked whole)."""
    return tuple(min(int(spatial_shape[a]), 64) for a in range(3))


def resolve_chunk_output_mode(cfg: Any) -> str:
    """chunk_grid.py:105-110."""
    mode = str(getattr(cfg.inference.chunking, "output_mode", "decoded")).lower()
    if mode not in {"decoded", "raw_prediction"}:
        raise ValueError("inference.chunking.output_mode must be 'decoded' or 'raw_prediction'.")
    return mode


__all__ = ["normalize_crop_pad", "resolve_selected_affinity_offsets", "resolve_global_prediction_crop",
           "validate_chunked_output_format", "resolve_chunk_shape", "resolve_h5_spatial_chunks", "resolve_chunk_output_mode"]
