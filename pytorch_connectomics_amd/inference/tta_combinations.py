"""Which test-time-augmentation views to run and how to ensemble them -- the contract of the reference's
connectomics/inference/tta_combinations.py: flips first, then rot90 in a plane; `flip_axes: all` expands to
[], [0], [1], [2], [0,1], [0,2], [1,2], [0,1,2]; with rotation planes the (flip, plane, k) triples are
de-duplicated by their effect on a probe tensor, keeping first occurrences."""
from __future__ import annotations

from itertools import combinations
from typing import Any, Optional

import torch

from ..utils.channel_slices import resolve_channel_range


def _to_plain_list(value) -> list:
    if isinstance(value, (list, tuple)):
        return [(_to_plain_list(v) if isinstance(v, (list, tuple)) else v) for v in value]
    if hasattr(value, "__iter__") and not isinstance(value, (str, bytes, dict)):
        return [(_to_plain_list(v) if hasattr(v, "__iter__") and not isinstance(v, (str, bytes)) else v)
                for v in value]
    return [value]


def _resolve_spatial_dims(ndim: int) -> int:
    if ndim == 5:
        return 3
    if ndim == 4:
        return 2
    raise ValueError(f"Unsupported data dimensions: {ndim}")


def _axes(axes: Any, spatial_dims: int, context: str) -> list[int]:
    if isinstance(axes, int):
        axes = [axes]
    if not isinstance(axes, (list, tuple)):
        raise ValueError(f"{context} must be an int or list of ints, got {axes!r}.")
    out: list[int] = []
    for raw in axes:
        a = int(raw)
        if a < 0 or a >= spatial_dims:
            raise ValueError(f"{context} axis must be in [0, {spatial_dims - 1}], got {a}.")
        if a not in out:
            out.append(a)
    return out


def _flip_variants(tta_cfg, spatial_dims: int) -> list[list[int]]:
    cfg = getattr(tta_cfg, "flip_axes", None)
    if isinstance(cfg, str) and cfg.lower() == "none":
        return [[]]
    if cfg == "all" or cfg == []:
        out: list[list[int]] = [[]]
        for r in range(1, spatial_dims + 1):
            out += [list(c) for c in combinations(range(spatial_dims), r)]
        return out
    if cfg is None:
        return [[]]
    return [[]] + [_axes(a, spatial_dims, "flip_axes") for a in _to_plain_list(cfg)]


def _rotation_planes(tta_cfg, spatial_dims: int) -> list[tuple[int, int]]:
    cfg = getattr(tta_cfg, "rotation90_axes", None)
    if cfg is None or (isinstance(cfg, str) and cfg.lower() == "none"):
        return []
    if cfg == "all":
        if spatial_dims == 3:
            return [(0, 1), (0, 2), (1, 2)]
        if spatial_dims == 2:
            return [(0, 1)]
        raise ValueError(f"Unsupported spatial dimensions: {spatial_dims}")
    planes: list[tuple[int, int]] = []
    for a in _to_plain_list(cfg):
        n = _axes(a, spatial_dims, "rotation90_axes")
        if len(n) != 2:
            raise ValueError(f"Invalid rotation plane: {a}. Each plane must contain exactly 2 axes.")
        if (n[0], n[1]) not in planes:
            planes.append((n[0], n[1]))
    return planes


def _rotation_ks(tta_cfg) -> list[int]:
    cfg = getattr(tta_cfg, "rotate90_k", None)
    if cfg is None:
        return [0, 1, 2, 3]
    out: list[int] = []
    for raw in _to_plain_list(cfg):
        k = int(raw) % 4
        if k not in out:
            out.append(k)
    return out or [0]


def apply_view(x: torch.Tensor, flip_axes, rotation_plane, k: int, *, first_spatial_dim: int) -> torch.Tensor:
    """The reference's view transform (inference/tta.py:712-719): flips, then rot90."""
    if flip_axes:
        x = torch.flip(x, dims=[a + first_spatial_dim for a in flip_axes])
    if rotation_plane is not None and k % 4:
        x = torch.rot90(x, k=k, dims=[rotation_plane[0] + first_spatial_dim, rotation_plane[1] + first_spatial_dim])
    return x


def _signature(spatial_dims: int, flip_axes, plane, k) -> tuple[int, ...]:
    base = torch.arange(30).reshape(2, 3, 5) if spatial_dims == 3 else torch.arange(10).reshape(2, 5)
    return tuple(int(v) for v in apply_view(base, flip_axes, plane, k, first_spatial_dim=0).reshape(-1).tolist())


def resolve_tta_augmentation_combinations(tta_cfg, *, spatial_dims: int):
    flips = _flip_variants(tta_cfg, spatial_dims)
    planes = _rotation_planes(tta_cfg, spatial_dims)
    if not planes:
        return [(f, None, 0) for f in flips]
    ks = _rotation_ks(tta_cfg)
    out, seen = [], set()
    for f in flips:
        for pl in planes:
            for k in ks:
                sig = _signature(spatial_dims, f, pl, k)
                if sig in seen:
                    continue
                seen.add(sig)
                out.append((f, pl, k))
    return out


def _resolve_ensemble_mode_map(ensemble_mode: Any, num_channels: int) -> list[str]:
    if isinstance(ensemble_mode, str):
        return [ensemble_mode] * num_channels
    raw = _to_plain_list(ensemble_mode)
    if not raw:
        raise ValueError("ensemble_mode must be a string or a list of [channel_selector, mode] pairs, "
                         f"got {ensemble_mode!r}.")
    if isinstance(raw[0], str) and len(raw) == 1:
        return [raw[0]] * num_channels
    modes: list[Optional[str]] = [None] * num_channels
    for entry in raw:
        if not isinstance(entry, (list, tuple)) or len(entry) != 2:
            raise ValueError(f"Each ensemble_mode entry must be [channel_selector, mode], got {entry!r}.")
        sel, mode = entry
        if mode not in ("mean", "min", "max"):
            raise ValueError(f"Unknown ensemble mode {mode!r} in per-channel spec. Use 'mean', 'min', or 'max'.")
        a, b = resolve_channel_range(str(sel), num_channels=num_channels, context="ensemble_mode channel selector")
        for c in range(a, b):
            modes[c] = mode
    unset = [i for i, m in enumerate(modes) if m is None]
    if unset:
        raise ValueError(f"ensemble_mode does not cover channels {unset}. Every channel must be assigned a mode.")
    return modes  # type: ignore[return-value]


__all__ = ["resolve_tta_augmentation_combinations", "apply_view", "_resolve_ensemble_mode_map",
           "_resolve_spatial_dims", "_to_plain_list"]
