"""Prediction-space crops (host integer logic): the user's `inference.model.crop_pad` plus, for DeepEM-style
affinities, the border where an offset's partner voxel lies outside the volume.

Contracts followed: connectomics/inference/chunk_grid.py:22-77 (normalize_crop_pad, resolve_selected_affinity_offsets,
resolve_global_prediction_crop) and connectomics/data/processing/affinity.py:291-358 (compute_affinity_crop_pad,
crop_spatial_by_pad, crop_spatial_by_offsets).  A crop is ((before, after),) * 3 in (z, y, x) order.
"""
from __future__ import annotations

from typing import Any, List, Optional, Sequence, Tuple

from ..utils.channel_slices import resolve_channel_indices
from ..utils.model_outputs import get_inference_select_channel
from .tta_affinity import resolve_affinity_channel_groups_from_cfg, resolve_affinity_mode_from_cfg

Pad = Tuple[Tuple[int, int], ...]
ZERO_CROP: Pad = ((0, 0), (0, 0), (0, 0))


def normalize_crop_pad(value: Any) -> Pad:
    """None / [] -> no crop; [z, y, x] -> symmetric; [z0, z1, y0, y1, x0, x1] -> asymmetric."""
    if value is None or (hasattr(value, "__len__") and len(value) == 0):
        return ZERO_CROP
    v = [int(a) for a in value]
    if len(v) == 3:
        return tuple((a, a) for a in v)
    if len(v) == 6:
        return tuple((v[2 * i], v[2 * i + 1]) for i in range(3))
    raise ValueError(f"inference.model.crop_pad must have length 3 or 6, got {value!r}")


def compute_affinity_crop_pad(offsets: Sequence[Sequence[int]], *, affinity_mode: str = "deepem") -> Pad:
    """Per axis, the largest reach of any offset: DeepEM stores edge (v - o, v) at v, so positive offsets invalidate the
    leading border and negative ones the trailing border; the other ("banis") convention mirrors that."""
    if not offsets:
        return tuple()
    mode = str(affinity_mode).strip().lower()
    if mode not in ("deepem", "banis"):
        raise ValueError(f"Unknown affinity_mode {affinity_mode!r}")
    nd = len(offsets[0])
    if any(len(o) != nd for o in offsets):
        raise ValueError(f"Mixed affinity offset dimensions are not supported: {offsets!r}")
    pos = [max(max(int(o[a]), 0) for o in offsets) for a in range(nd)]
    neg = [max(max(-int(o[a]), 0) for o in offsets) for a in range(nd)]
    return tuple((pos[a], neg[a]) if mode == "deepem" else (neg[a], pos[a]) for a in range(nd))


def crop_spatial_by_pad(data, crop_pad: Sequence[Tuple[int, int]], *, item_name: str = "data"):
    """Drop (before, after) voxels from the trailing len(crop_pad) axes of a numpy array or tensor (a view)."""
    if not crop_pad:
        return data
    k = len(crop_pad)
    if data.ndim < k:
        raise ValueError(f"Cannot crop {item_name}: rank {data.ndim} is smaller than crop rank {k}")
    index = [slice(None)] * data.ndim
    for i, (lo, hi) in enumerate(crop_pad):
        n = int(data.shape[data.ndim - k + i])
        if lo < 0 or hi < 0:
            raise ValueError(f"Crop pad must be non-negative for {item_name}, got {crop_pad}")
        if lo + hi >= n:
            raise ValueError(f"Cannot crop {item_name}: crop pad {tuple(crop_pad)} is too large for shape {tuple(data.shape)}")
        index[data.ndim - k + i] = slice(lo, n - hi)
    return data[tuple(index)]


def crop_spatial_by_offsets(data, offsets, *, affinity_mode: str = "deepem", item_name: str = "data"):
    return crop_spatial_by_pad(data, compute_affinity_crop_pad(offsets, affinity_mode=affinity_mode), item_name=item_name)


def resolve_selected_affinity_offsets(cfg: Any) -> List[Tuple[int, int, int]]:
    """Offsets of the affinity channels that survive `inference.model.select_channel`, in output order."""
    groups = resolve_affinity_channel_groups_from_cfg(cfg)
    if not groups:
        return []
    per_channel: List[Optional[Tuple[int, int, int]]] = [None] * max(hi for (_, hi), _ in groups)
    for (lo, hi), offs in groups:
        for ch, off in zip(range(lo, hi), offs):
            per_channel[ch] = off
    sel = get_inference_select_channel(cfg)
    if sel is not None:
        keep = resolve_channel_indices(sel, num_channels=len(per_channel), context="inference.model.select_channel")
        per_channel = [per_channel[i] for i in keep]
    return [o for o in per_channel if o is not None]


def resolve_global_prediction_crop(cfg: Any) -> Pad:
    """user crop_pad + (DeepEM mode only) the affinity validity border, per axis and side."""
    model_cfg = getattr(getattr(cfg, "inference", None), "model", None)
    user = normalize_crop_pad(getattr(model_cfg, "crop_pad", None))
    aff = ZERO_CROP
    if resolve_affinity_mode_from_cfg(cfg) == "deepem":
        offs = resolve_selected_affinity_offsets(cfg)
        if offs:
            aff = compute_affinity_crop_pad(offs, affinity_mode="deepem")
    return tuple((int(user[a][0]) + int(aff[a][0]), int(user[a][1]) + int(aff[a][1])) for a in range(3))


def cropped_shape(shape: Sequence[int], crop_pad: Pad) -> Tuple[int, int, int]:
    out = tuple(int(shape[a]) - crop_pad[a][0] - crop_pad[a][1] for a in range(3))
    if any(v <= 0 for v in out):
        raise ValueError(f"Chunked inference crop {crop_pad} is too large for input shape {tuple(shape)}.")
    return out


__all__ = ["normalize_crop_pad", "compute_affinity_crop_pad", "crop_spatial_by_pad", "crop_spatial_by_offsets",
           "resolve_selected_affinity_offsets", "resolve_global_prediction_crop", "cropped_shape", "ZERO_CROP"]
