"""Enumeration of test-time-augmentation views and the per-channel ensemble rule.

Contract (reference connectomics/inference/tta_combinations.py, pinned by tests/golden/tta_affinity_plans.json and the
reference `predict_with_tta` fixtures): a view is "flip some spatial axes, then rotate by k quarter turns in one plane";
`flip_axes: all` lists the flip subsets by size ([], [0], [1], [2], [0,1], ...); with rotation planes the
(flip, plane, k) triples are walked flips-outermost and a triple is dropped when an earlier one moves the voxels the
same way.

Design here: a view is an element of the hyperoctahedral group -- every output axis reads one input axis, forwards or
backwards -- held as a tuple `(source_axis, reversed)` per output axis (`ViewMap`).  Flips and quarter turns compose as
such maps, two triples are the same view exactly when their maps are equal, and the engine's 4-bit window code
(`tta.view_code`) is read off the same map.  No probe tensors are involved.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Iterable, Optional, Sequence

import torch

from ..utils.channel_slices import resolve_channel_range

_ENSEMBLE_MODES = ("mean", "min", "max")


# ------------------------------------------------------------------------------------------- view algebra
@dataclass(frozen=True)
class ViewMap:
    """Output axis i of the view reads input axis `src[i]`, back to front when `rev[i]`."""
    src: tuple
    rev: tuple

    @staticmethod
    def identity(nd: int) -> "ViewMap":
        return ViewMap(tuple(range(nd)), (False,) * nd)

    def flipped(self, axes: Iterable[int]) -> "ViewMap":
        rev = list(self.rev)
        for a in axes:
            rev[a] = not rev[a]
        return ViewMap(self.src, tuple(rev))

    def quarter_turn(self, plane: Sequence[int]) -> "ViewMap":
        """One `torch.rot90(k=1, dims=plane)`: exchange the plane's axes, then reverse the first of them."""
        a, b = plane
        src, rev = list(self.src), list(self.rev)
        src[a], src[b] = src[b], src[a]
        rev[a], rev[b] = rev[b], rev[a]
        rev[a] = not rev[a]
        return ViewMap(tuple(src), tuple(rev))

    @staticmethod
    def of(nd: int, flip_axes: Sequence[int], plane: Optional[Sequence[int]], k: int) -> "ViewMap":
        v = ViewMap.identity(nd).flipped(flip_axes)
        if plane is not None:
            for _ in range(int(k) % 4):
                v = v.quarter_turn(plane)
        return v


def apply_view(x: torch.Tensor, flip_axes, rotation_plane, k: int, *, first_spatial_dim: int) -> torch.Tensor:
    """The view as a tensor op (flips, then rot90 -- reference inference/tta.py:712-719); spatial axes start at
    `first_spatial_dim`."""
    dims = [int(a) + first_spatial_dim for a in (flip_axes or [])]
    if dims:
        x = torch.flip(x, dims=dims)
    if rotation_plane is not None and int(k) % 4:
        x = torch.rot90(x, k=int(k), dims=[int(rotation_plane[0]) + first_spatial_dim, int(rotation_plane[1]) + first_spatial_dim])
    return x


# ------------------------------------------------------------------------------------------- config parsing
def _to_plain_list(value) -> list:
    """Config containers (OmegaConf lists, tuples, generators) -> nested python lists; a scalar becomes [scalar]."""
    if isinstance(value, (str, bytes, dict)) or not hasattr(value, "__iter__"):
        return [value]
    return [_to_plain_list(v) if (hasattr(v, "__iter__") and not isinstance(v, (str, bytes, dict))) else v for v in value]


def _resolve_spatial_dims(ndim: int) -> int:
    try:
        return {5: 3, 4: 2}[ndim]
    except KeyError:
        raise ValueError(f"Unsupported data dimensions: {ndim}") from None


def _axis_set(spec: Any, nd: int, what: str) -> list:
    """An int or a list of ints -> distinct spatial axes in the order given."""
    items = [spec] if isinstance(spec, int) else spec
    if not isinstance(items, (list, tuple)):
        raise ValueError(f"{what} must be an int or list of ints, got {spec!r}.")
    axes: list = []
    for item in items:
        a = int(item)
        if not 0 <= a < nd:
            raise ValueError(f"{what} axis must be in [0, {nd - 1}], got {a}.")
        if a not in axes:
            axes.append(a)
    return axes


def _is_word(value, word: str) -> bool:
    return isinstance(value, str) and value.lower() == word


def _flip_sets(tta_cfg, nd: int) -> list:
    spec = getattr(tta_cfg, "flip_axes", None)
    if spec is None or _is_word(spec, "none"):
        return [[]]
    if spec == "all" or spec == []:
        # subsets by size, each size in lexicographic order = ascending popcount, then ascending axis tuples
        subsets = [[a for a in range(nd) if mask >> a & 1] for mask in range(1 << nd)]
        return sorted(subsets, key=lambda s: (len(s), s))
    return [[]] + [_axis_set(entry, nd, "flip_axes") for entry in _to_plain_list(spec)]


def _planes(tta_cfg, nd: int) -> list:
    spec = getattr(tta_cfg, "rotation90_axes", None)
    if spec is None or _is_word(spec, "none"):
        return []
    if spec == "all":
        if nd not in (2, 3):
            raise ValueError(f"Unsupported spatial dimensions: {nd}")
        return [(a, b) for a in range(nd) for b in range(a + 1, nd)]
    planes: list = []
    for entry in _to_plain_list(spec):
        axes = _axis_set(entry, nd, "rotation90_axes")
        if len(axes) != 2:
            raise ValueError(f"Invalid rotation plane: {entry}. Each plane must contain exactly 2 axes.")
        if tuple(axes) not in planes:
            planes.append(tuple(axes))
    return planes


def _quarter_turns(tta_cfg) -> list:
    spec = getattr(tta_cfg, "rotate90_k", None)
    if spec is None:
        return [0, 1, 2, 3]
    ks = list(dict.fromkeys(int(v) % 4 for v in _to_plain_list(spec)))
    return ks or [0]


def resolve_tta_augmentation_combinations(tta_cfg, *, spatial_dims: int):
    """-> [(flip_axes, rotation_plane | None, k)] in the order the ensemble visits them."""
    flips, planes = _flip_sets(tta_cfg, spatial_dims), _planes(tta_cfg, spatial_dims)
    if not planes:
        return [(f, None, 0) for f in flips]
    ks = _quarter_turns(tta_cfg)
    views: dict = {}           # ViewMap -> first triple that produces it (dicts keep insertion order)
    for f in flips:
        for plane in planes:
            for k in ks:
                views.setdefault(ViewMap.of(spatial_dims, f, plane, k), (f, plane, k))
    return list(views.values())


# ------------------------------------------------------------------------------------------- ensemble rule
def _resolve_ensemble_mode_map(ensemble_mode: Any, num_channels: int) -> list:
    """'mean' | 'min' | 'max', or [[channel_selector, mode], ...] covering every channel -> one mode per channel."""
    if isinstance(ensemble_mode, str):
        return [ensemble_mode] * num_channels
    entries = _to_plain_list(ensemble_mode)
    if not entries:
        raise ValueError("ensemble_mode must be a string or a list of [channel_selector, mode] pairs, "
                         f"got {ensemble_mode!r}.")
    if len(entries) == 1 and isinstance(entries[0], str):
        return [entries[0]] * num_channels
    per_channel: dict = {}
    for entry in entries:
        if not isinstance(entry, (list, tuple)) or len(entry) != 2:
            raise ValueError(f"Each ensemble_mode entry must be [channel_selector, mode], got {entry!r}.")
        selector, mode = entry
        if mode not in _ENSEMBLE_MODES:
            raise ValueError(f"Unknown ensemble mode {mode!r} in per-channel spec. Use 'mean', 'min', or 'max'.")
        lo, hi = resolve_channel_range(str(selector), num_channels=num_channels, context="ensemble_mode channel selector")
        per_channel.update({c: mode for c in range(lo, hi)})
    missing = [c for c in range(num_channels) if c not in per_channel]
    if missing:
        raise ValueError(f"ensemble_mode does not cover channels {missing}. Every channel must be assigned a mode.")
    return [per_channel[c] for c in range(num_channels)]


__all__ = ["ViewMap", "resolve_tta_augmentation_combinations", "apply_view", "_resolve_ensemble_mode_map",
           "_resolve_spatial_dims", "_to_plain_list"]
