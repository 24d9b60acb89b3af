"""Affinity-aware TTA: which output channel an augmented view's channel lands in, and by how much it is displaced.

Reference: connectomics/inference/tta_affinity.py:72-393 (plan + `invert_view`) and the label-layout resolvers of
connectomics/data/processing/affinity.py:142-256.  An affinity channel with offset `o` predicts "voxel p and voxel
p+o belong together".  Flipping / rotating the input maps `o` to `o' = T(o)`; when `o'` is another configured offset
the channel simply moves there, when only `-o'` is configured the edge is the same one seen from its other end, so the
channel moves AND its values shift by the full offset (sign by storage convention: deepem stores the edge at the
destination voxel, banis at the source).  Shifted channels lose a band of `|o|` voxels on one face of every window:
those faces are tracked as missing validity and never enter the mean / min / max.

This module is host-side integer logic only; the data movement happens inside the blending kernel
(`pytc_blend_accumulate_mapped`: channel map + per-channel shift applied while a window is scattered into the
accumulators), so no re-ordered copy of a prediction is ever materialised.
"""
from __future__ import annotations

from collections.abc import Mapping, Sequence
from dataclasses import dataclass
from typing import Any, List, Optional, Tuple

from ..utils.channel_slices import resolve_channel_range

Offset = Tuple[int, ...]
Group = Tuple[Tuple[int, int], List[Offset]]


# ---- label layout (data/processing/affinity.py:86-256) ---------------------------------------------------------------
def _get(node: Any, key: str, default: Any = None) -> Any:
    if isinstance(node, Mapping):
        return node.get(key, default)
    return getattr(node, key, default)


def _target_name(task: Any) -> Optional[str]:
    if isinstance(task, str):
        return task
    for key in ("name", "task", "type"):
        v = _get(task, key, None)
        if v is not None:
            return v
    return None


def _target_kwargs(task: Any) -> dict:
    raw = None if isinstance(task, str) else _get(task, "kwargs", None)
    return {} if raw is None else {k: raw[k] for k in raw}


def _targets(cfg: Any) -> list:
    lt = _get(_get(cfg, "data", None), "label_transform", None)
    t = None if lt is None else _get(lt, "targets", None)
    if t is None:
        return []
    return [t] if isinstance(t, str) else list(t)


def parse_affinity_offsets(offsets: Sequence[Any]) -> List[Offset]:
    out: List[Offset] = []
    for o in offsets:
        if isinstance(o, str):
            parts = o.split("-")
            if len(parts) != 3:
                raise ValueError(f"Invalid affinity offset {o!r}. Expected 'z-y-x' format.")
            out.append(tuple(int(p) for p in parts))
        elif isinstance(o, (list, tuple)) and len(o) == 3:
            out.append(tuple(int(v) for v in o))
        else:
            raise ValueError(f"Unsupported affinity offset {o!r}. Expected 'z-y-x' string or length-3 sequence.")
    return out


def resolve_affinity_offsets_from_kwargs(kwargs: dict) -> List[Offset]:
    lr = kwargs.get("long_range", None)
    if lr is not None:
        lr = int(lr)
        return [(1, 0, 0), (0, 1, 0), (0, 0, 1), (lr, 0, 0), (0, lr, 0), (0, 0, lr)]
    offsets = kwargs.get("offsets", None)
    if offsets is None or len(offsets) == 0:
        offsets = ["1-0-0", "0-1-0", "0-0-1"]
    return parse_affinity_offsets(offsets)


def resolve_affinity_mode_from_cfg(cfg: Any) -> Optional[str]:
    modes = []
    for task in _targets(cfg):
        if _target_name(task) != "affinity":
            continue
        mode = _target_kwargs(task).get("affinity_mode")
        if mode is None:
            raise ValueError("Affinity targets require kwargs.affinity_mode: 'deepem' or 'banis'.")
        m = str(mode).strip().lower()
        if m not in ("deepem", "banis"):
            raise ValueError(f"Unsupported affinity_mode {mode!r}. Expected one of: deepem, banis.")
        modes.append(m)
    if not modes:
        return None
    uniq = sorted(set(modes))
    if len(uniq) != 1:
        raise ValueError(f"Mixed affinity_mode values are not supported in one label stack: {uniq}")
    return uniq[0]


def _stacked_label_layout(cfg: Any) -> Tuple[int, List[Group]]:
    lt = _get(_get(cfg, "data", None), "label_transform", None)
    tasks = _targets(cfg)
    if lt is None or not tasks or not bool(_get(lt, "stack_outputs", True)):
        return 0, []
    groups: List[Group] = []
    start = 0
    for task in tasks:
        name, kw = _target_name(task), _target_kwargs(task)
        if name == "affinity":
            offs = resolve_affinity_offsets_from_kwargs(kw)
            groups.append(((start, start + len(offs)), offs))
            width = len(offs)
        elif name == "polarity":
            width = 1 if bool(kw.get("exclusive", False)) else 3
        else:
            width = 1
        start += width
    return start, groups


def resolve_affinity_channel_groups_from_cfg(cfg: Any) -> List[Group]:
    return _stacked_label_layout(cfg)[1]


def resolve_stacked_label_channel_count(cfg: Any) -> int:
    return _stacked_label_layout(cfg)[0]


# ---- plan (inference/tta_affinity.py:22-347) -------------------------------------------------------------------------
@dataclass(frozen=True)
class ChannelMove:
    src: int
    dst: int
    shift: Optional[Offset] = None


@dataclass(frozen=True)
class AffinityViewPlan:
    moves: Tuple[ChannelMove, ...]
    partial_channels: frozenset

    def shift_for_channel(self, channel: int) -> Optional[Offset]:
        for m in self.moves:
            if m.dst == channel:
                return m.shift
        return None

    def channel_map(self, num_channels: int) -> Tuple[List[int], List[Offset]]:
        """(src[dst], shift[dst]) for every output channel; identity / zero shift where the plan has no move."""
        src = list(range(num_channels))
        shift: List[Offset] = [(0, 0, 0)] * num_channels
        for m in self.moves:
            src[m.dst] = m.src
            shift[m.dst] = tuple(m.shift) if m.shift is not None else (0, 0, 0)
        return src, shift


@dataclass(frozen=True)
class ViewValidity:
    """Per output channel of one inverted view: None = valid everywhere, a tuple of slices = the box that received real
    (non-wrapped) values, or a boolean mask (reference tta_affinity.py ViewValidity / ValidityEntry)."""
    channels: tuple

    @classmethod
    def all_valid(cls, num_channels: int) -> "ViewValidity":
        """Every channel valid everywhere (a view that moved no affinity channel)."""
        return cls((None,) * int(num_channels))

    def select(self, indices: Optional[Sequence[int]]) -> "ViewValidity":
        """The validity of a channel selection, in the selection's order (None = all channels)."""
        return self if indices is None else ViewValidity(tuple(self.channels[int(i)] for i in indices))


@dataclass(frozen=True)
class AffinityTTAPlan:
    views: Tuple[AffinityViewPlan, ...]
    partial_channels: frozenset
    shifts: frozenset
    num_channels: int
    spatial_rank: int


def validate_affinity_output(plan: Optional["AffinityTTAPlan"], prediction) -> None:
    """A prediction (N, C, *spatial) must have the channel count and spatial rank the plan was derived for
    (reference tta_affinity.py:332-347); no plan, nothing to check."""
    if plan is None:
        return
    channels, rank = int(prediction.shape[1]), int(prediction.dim()) - 2
    if channels != int(plan.num_channels):
        raise ValueError(f"Affinity TTA plan expects {plan.num_channels} raw output channels, but the model produced {channels}.")
    if plan.spatial_rank and int(plan.spatial_rank) != rank:
        raise ValueError(f"Affinity offset rank {plan.spatial_rank} does not match raw output spatial rank {rank}.")


def transform_offset(offset: Sequence[int], *, flip_axes: Sequence[int], rotation_plane_spatial, k: int) -> Offset:
    """Linear part of (inverse rotation, then inverse flips) applied to an offset vector."""
    v = [int(c) for c in offset]
    if rotation_plane_spatial is not None:
        p, q = (int(a) for a in rotation_plane_spatial)
        if p == q or min(p, q) < 0 or max(p, q) >= len(v):
            raise ValueError(f"Rotation plane {rotation_plane_spatial} is invalid for an offset with rank {len(v)}.")
        for _ in range((-int(k)) % 4):
            v[p], v[q] = -v[q], v[p]
    for a in flip_axes:
        a = int(a)
        if a < 0 or a >= len(v):
            raise ValueError(f"Flip axis {a} is invalid for an offset with rank {len(v)}.")
        v[a] = -v[a]
    return tuple(v)


def valid_slices_for_shift(spatial_shape: Sequence[int], shift: Sequence[int]) -> Tuple[slice, ...]:
    """Box of positions that receive real (non-wrapped) values after displacing a window by `shift`."""
    if len(spatial_shape) != len(shift):
        raise ValueError(f"Roll shift rank {len(shift)} does not match spatial rank {len(spatial_shape)}.")
    out = []
    for n, s in zip(spatial_shape, shift):
        n, s = int(n), int(s)
        out.append(slice(min(s, n), n) if s > 0 else (slice(0, max(0, n + s)) if s < 0 else slice(0, n)))
    return tuple(out)


def invert_view(prediction, *, flip_axes: Sequence[int], rotation_plane_spatial, k: int, view_plan: Optional["AffinityViewPlan"],
                tta_plan: Optional["AffinityTTAPlan"]):
    """A whole prediction (N, C, *spatial) of one TTA view mapped back to the canonical frame: inverse quarter turns, inverse flips,
    then the affinity channel moves of `view_plan` -- a shifted channel is displaced by its full offset, the positions that would
    wrap are zeroed and reported as missing validity (reference tta_affinity.py:350-393).  -> (tensor, ViewValidity).

    Public adapter for callers of the reference API.  The engine itself never materialises an un-inverted prediction: the same
    index map runs per window inside `pytc_blend_accumulate_mapped` (AffinityViewPlan.channel_map).  Pure data movement (torch
    indexing on whatever device the prediction lives on)."""
    import torch
    out = prediction
    if rotation_plane_spatial is not None and int(k) % 4:
        out = torch.rot90(out, k=-int(k), dims=tuple(int(a) + 2 for a in rotation_plane_spatial))
    if flip_axes:
        out = torch.flip(out, dims=[int(a) + 2 for a in flip_axes])
    validate_affinity_output(tta_plan, out)
    validity: list = [None] * int(out.shape[1])
    if view_plan is None or not view_plan.moves:
        return out, ViewValidity(tuple(validity))
    fixed = out.clone()
    spatial = tuple(int(v) for v in out.shape[2:])
    for m in view_plan.moves:
        if m.shift is None:
            fixed[:, m.dst] = out[:, m.src]
            continue
        if len(m.shift) != len(spatial):
            raise ValueError(f"Affinity roll shift rank {len(m.shift)} does not match raw output spatial rank {len(spatial)}.")
        dst_box = valid_slices_for_shift(spatial, m.shift)
        src_box = tuple(slice(b.start - int(s), b.stop - int(s)) for b, s in zip(dst_box, m.shift))
        fixed[:, m.dst].zero_()
        fixed[(slice(None), m.dst) + dst_box] = out[(slice(None), m.src) + src_box]
        validity[m.dst] = dst_box
    return fixed, ViewValidity(tuple(validity))


def _raw_groups(cfg: Any, *, num_raw: int, requested_head: Optional[str]) -> List[Group]:
    """Affinity groups expressed in RAW output channels of the selected head (tta_affinity.py:140-229)."""
    label_groups = resolve_affinity_channel_groups_from_cfg(cfg)
    if not label_groups:
        return []
    total = resolve_stacked_label_channel_count(cfg)
    model_cfg = _get(cfg, "model", None)
    heads = _get(model_cfg, "heads", {}) or {}
    if not isinstance(heads, Mapping):
        heads = {}
    if not heads:
        declared = _get(model_cfg, "out_channels", None)
        if declared is None or int(declared) != num_raw or total != num_raw:
            raise ValueError(
                "Affinity TTA requires an unambiguous raw-output to stacked-label mapping. "
                f"Got model.out_channels={declared}, raw output channels={num_raw}, and "
                f"stacked label channels={total}; all three must match.")
        window = (0, num_raw)
    else:
        name = requested_head
        if name is None and len(heads) == 1:
            name = next(iter(heads))
        if name is None or name not in heads:
            raise ValueError("Affinity TTA cannot map a named raw output to label channels. Select one "
                             "model head and declare model.heads.<name>.target_slice.")
        hc = heads[name]
        ts = _get(hc, "target_slice", None)
        if ts is not None:
            a, b = resolve_channel_range(ts, num_channels=total, context=f"model.heads.{name}.target_slice")
            if b - a != num_raw:
                raise ValueError(f"model.heads.{name}.target_slice resolves to width {b - a}, "
                                 f"but the raw output has {num_raw} channels.")
            window = (a, b)
        else:
            width = int(_get(hc, "out_channels", 0))
            if len(heads) != 1 or width != num_raw or total != num_raw:
                raise ValueError(
                    f"Affinity TTA cannot prove the label mapping for model head {name!r}. "
                    f"Got {len(heads)} configured head(s), head out_channels={width}, "
                    f"raw output channels={num_raw}, and stacked label channels={total}. "
                    f"Declare model.heads.{name}.target_slice.")
            window = (0, num_raw)
    out: List[Group] = []
    for (g0, g1), offs in label_groups:
        lo, hi = max(g0, window[0]), min(g1, window[1])
        if lo >= hi:
            continue
        if len(offs) != g1 - g0:
            raise ValueError(f"Affinity group [{g0}, {g1}) declares {len(offs)} offsets; "
                             "its width and offset count must match.")
        # a head may cover a contiguous part of a multi-radius group; offsets are positional inside the group
        out.append(((lo - window[0], hi - window[0]), [tuple(o) for o in offs[lo - g0:hi - g0]]))
    return out


def build_affinity_tta_plan(cfg: Any, *, augmentation_combinations, num_raw: int,
                            requested_head: Optional[str]) -> Optional[AffinityTTAPlan]:
    """tta_affinity.py:232-326: one AffinityViewPlan per configured view, or None without affinity targets."""
    if not resolve_affinity_channel_groups_from_cfg(cfg):
        return None
    groups = _raw_groups(cfg, num_raw=int(num_raw), requested_head=requested_head)
    mode = resolve_affinity_mode_from_cfg(cfg)
    if mode is None:
        raise ValueError("Affinity channel groups exist but no affinity_mode could be resolved.")
    ranks = {len(o) for _r, offs in groups for o in offs}
    if len(ranks) > 1:
        raise ValueError(f"Mixed affinity offset ranks are not supported: {sorted(ranks)}.")
    rank = next(iter(ranks), 0)
    for rng, offs in groups:
        if len(set(offs)) != len(offs):
            raise ValueError(f"Affinity group {rng} contains duplicate offsets: {offs!r}.")
    views, all_partial, all_shifts = [], set(), set()
    for flips, plane, k in augmentation_combinations:
        moves: List[ChannelMove] = []
        taken = set()
        for (start, stop), offs in groups:
            if stop - start != len(offs):
                raise ValueError(f"Affinity group [{start}, {stop}) width does not match its "
                                 f"{len(offs)} configured offsets.")
            for si, off in enumerate(offs):
                d = transform_offset(off, flip_axes=flips, rotation_plane_spatial=plane, k=k)
                exact = [i for i, t in enumerate(offs) if t == d]
                mirrored = [i for i, t in enumerate(offs) if tuple(-c for c in t) == d]
                cand = exact if exact else mirrored
                if len(cand) != 1:
                    kind = "exact" if exact else "sign-reversed"
                    raise ValueError(f"Affinity offset {off} transforms to {d}, but group "
                                     f"{offs!r} has {len(cand)} {kind} counterpart(s).")
                ti = cand[0]
                dst = start + ti
                if dst in taken:
                    raise ValueError("Affinity TTA channel mapping is not bijective: multiple source "
                                     f"channels target raw channel {dst}.")
                taken.add(dst)
                shift = None
                if not exact:
                    sign = -1 if mode == "banis" else 1
                    shift = tuple(sign * int(c) for c in offs[ti])
                    if any(shift):
                        all_partial.add(dst)
                        all_shifts.add(shift)
                    else:
                        shift = None
                moves.append(ChannelMove(src=start + si, dst=dst, shift=shift))
            if {m.dst for m in moves if start <= m.dst < stop} != set(range(start, stop)):
                raise ValueError(f"Affinity TTA mapping for group [{start}, {stop}) is not bijective.")
        views.append(AffinityViewPlan(tuple(moves), frozenset(m.dst for m in moves if m.shift is not None)))
    return AffinityTTAPlan(views=tuple(views), partial_channels=frozenset(all_partial), shifts=frozenset(all_shifts),
                           num_channels=int(num_raw), spatial_rank=rank)


__all__ = ["AffinityTTAPlan", "AffinityViewPlan", "ChannelMove", "ViewValidity", "invert_view", "validate_affinity_output", "build_affinity_tta_plan", "transform_offset",
           "valid_slices_for_shift", "parse_affinity_offsets", "resolve_affinity_offsets_from_kwargs",
           "resolve_affinity_mode_from_cfg", "resolve_affinity_channel_groups_from_cfg",
           "resolve_stacked_label_channel_count"]
