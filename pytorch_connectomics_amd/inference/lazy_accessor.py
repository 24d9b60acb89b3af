"""Disk-backed test volumes for the lazy / chunked sliding-window path: the reader the reference calls `LazyVolumeAccessor`
(connectomics/inference/lazy.py:456-917) re-designed around the device.

What a window of the reference's reader goes through -- `val_transpose`, `resize` (nearest for labels / masks, trilinear
align_corners=True for images), `pad_size` context padding (constant / reflect / edge), the window's own outer padding, mask
binarisation and `smart_normalize` (data/augmentation/augment_ops.py:552-611) -- is split here into

  host   one `VolumeSource.read_box`: the RAW stored values of the storage box a region needs (storage dtype, storage axis order),
         optionally into pinned memory on an IO thread (`RegionPrefetcher`), plus three small per-axis index tables (`AxisMap.table`)
         that say which raw indices every output index reads;
  device `pytc_resample_region` (transpose + resize + context pad in one gather over the raw box -> fp32 (C, z, y, x) region in HBM),
         the sliding-window gather kernel (outer padding, as for in-memory volumes), and `pytc_window_normalize` (binarise, percentile
         clip, z-score / min-max / divide-K with the statistics of each WINDOW) -- `LazyVolumeAccessor.finish_windows`.

Nothing is resampled or normalised in numpy; `read_patch` / `read_region` / `load_full` (the reference's host-array API, kept for its
callers and for the reference fixtures of tests/golden/lazy_accessor.npz) run the same kernels and copy the result back.
"""
from __future__ import annotations

import math
import threading
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from .. import _native as nat
from .. import hip_ops as ops
from .volume_source import VolumeSource, ZarrV2Array, box_strides, detect_format

_NORM_CODES = {"none": nat.NORM_NONE, "normal": nat.NORM_ZSCORE, "0-1": nat.NORM_MINMAX, "divide": nat.NORM_DIVIDE}


# ------------------------------------------------------------------------------------------------ geometry
@dataclass(frozen=True)
class AxisMap:
    """One axis of the presented volume: `stored` raw samples -> `resized` samples (nearest or linear, align_corners) -> context
    pad of (before, after) samples in `pad_mode`.  `table(lo, hi)` composes the three steps BACKWARDS for output indices
    [lo, hi): -> (i0, i1, f) with out = (1 - f) * raw[i0] + f * raw[i1]; i0 = -1 marks a sample outside a constant pad."""
    stored: int
    resized: int
    pad: Tuple[int, int]
    pad_mode: str
    linear: bool
    resizes: bool

    @property
    def length(self) -> int:
        return self.resized + self.pad[0] + self.pad[1]

    def _unpad(self, idx: np.ndarray):
        t = self.resized
        if self.pad_mode == "constant":
            return (idx >= 0) & (idx < t), np.clip(idx, 0, max(t - 1, 0))
        ok = np.ones(idx.shape, dtype=bool)
        if self.pad_mode == "edge" or t <= 1:
            return ok, np.clip(idx, 0, max(t - 1, 0))
        if self.pad_mode == "reflect":
            period = 2 * t - 2
            m = np.abs(idx) % period
            return ok, np.where(m < t, m, period - m)
        raise ValueError(f"Unsupported context pad mode '{self.pad_mode}'.")

    def table(self, lo: int, hi: int):
        valid, m = self._unpad(np.arange(int(lo), int(hi), dtype=np.int64) - int(self.pad[0]))
        f = np.zeros(m.shape, dtype=np.float32)
        if not self.resizes:
            i0 = i1 = m
        elif self.stored <= 1 or self.resized <= 1:
            i0 = i1 = np.zeros_like(m)
        elif not self.linear:
            # floor(i * in / out) in float32, the arithmetic of the reference's nearest map
            c = np.floor(m.astype(np.float32) * np.float32(self.stored) / np.float32(self.resized))
            i0 = i1 = np.clip(c, 0, self.stored - 1).astype(np.int64)
        else:
            c = m.astype(np.float32) * np.float32(self.stored - 1) / np.float32(self.resized - 1)      # align_corners=True
            i0 = np.clip(np.floor(c), 0, self.stored - 1).astype(np.int64)
            i1 = np.minimum(i0 + 1, self.stored - 1)
            f = (c - i0.astype(np.float32)).astype(np.float32)
            f[i1 == i0] = 0.0
        i0 = np.where(valid, i0, -1)
        return i0.astype(np.int64), np.where(valid, i1, -1).astype(np.int64), np.where(valid, f, 0).astype(np.float32)


def _resized_length(length: int, factor: float) -> int:
    if factor <= 0:
        raise ValueError(f"scale factor must be positive, got {factor}.")
    return max(1, int(math.floor(float(length) * float(factor) + 1e-6)))


def _pad_mode_name(mode: str) -> str:
    m = str(mode).lower()
    return "edge" if m == "replicate" else m


class StagedRegion:
    """Raw storage bytes of one box + the index tables that turn them into the presented region: everything the device needs.
    `raw` is a host uint8 tensor (pinned when staged by the prefetcher); `to_device` does the H2D copy and ONE kernel."""

    def __init__(self, box: Tuple[Tuple[int, int, int], Tuple[int, int, int]], raw: Optional[torch.Tensor], raw_dtype: str,
                 strides, channels: int, tables, dims):
        self.box = box
        self.raw, self.raw_dtype, self.strides, self.channels = raw, raw_dtype, strides, int(channels)
        self.tables, self.dims = tables, tuple(int(v) for v in dims)

    @property
    def shape(self):
        return (self.channels,) + self.dims

    def pin(self) -> "StagedRegion":
        if self.raw is not None and torch.cuda.is_available() and not self.raw.is_pinned():
            self.raw = self.raw.pin_memory()
        return self

    def to_device(self, device, non_blocking: bool = True) -> torch.Tensor:
        dev = torch.device(device)
        ops.require_device(dev, "LazyVolumeAccessor")
        if self.raw is None:                                   # the whole box lies in a constant context pad
            return torch.zeros(self.shape, dtype=torch.float32, device=dev)
        i0, i1, f = (t.to(dev, non_blocking=non_blocking) for t in self.tables)
        return ops.resample_region(self.raw.to(dev, non_blocking=non_blocking), self.raw_dtype, self.strides, self.channels,
                                   i0, i1, f, self.dims)


# ------------------------------------------------------------------------------------------------ the accessor
class LazyVolumeAccessor:
    """Random access to the test volume as the model sees it (channel-first, transposed, resized, context-padded), computed on
    the device from raw storage boxes.  Attribute names follow the reference class so configuration code reads alike."""

    ndim = 4

    def __init__(self, path: str, *, kind: str, transpose_axes: Sequence[int] = (), scale_factors=None, context_pad=None,
                 context_pad_mode: str = "constant", normalize_mode: str = "none", clip_percentile_low: float = 0.0,
                 clip_percentile_high: float = 1.0, binarize: bool = False, threshold: float = 0.0, tile_read_workers: int = 1,
                 device=None):
        self.path, self.kind = str(path), kind
        self.tile_read_workers = max(1, int(tile_read_workers))
        axes = tuple(int(a) for a in (transpose_axes or ()))
        if axes and (len(axes) != 3 or sorted(axes) != [0, 1, 2]):
            raise ValueError(f"transpose_axes must be a permutation of [0,1,2], got {transpose_axes}")
        self.transpose_axes = axes
        self.scale_factors = tuple(float(v) for v in scale_factors) if scale_factors is not None else None
        self.context_pad = tuple(tuple(int(v) for v in p) for p in (context_pad or ((0, 0),) * 3))
        self.context_pad_mode = context_pad_mode
        self.normalize_mode = normalize_mode
        self.clip_percentile_low, self.clip_percentile_high = float(clip_percentile_low), float(clip_percentile_high)
        self.binarize, self.threshold = bool(binarize), float(threshold)
        self.device = device
        self._norm_code, self._divisor = self._parse_normalize(normalize_mode) if kind == "image" else (nat.NORM_NONE, 1.0)
        self.source = VolumeSource(self.path, kind=kind, read_workers=self.tile_read_workers)
        self.fmt = self.source.fmt
        self.channel_count = self.source.channels
        self.raw_spatial_shape = self.source.spatial_shape
        self._stored_axis = axes or (0, 1, 2)                  # logical axis a is stored along spatial axis _stored_axis[a]
        self.logical_spatial_shape = tuple(self.raw_spatial_shape[s] for s in self._stored_axis)
        linear = kind not in {"label", "mask"}
        mode = _pad_mode_name(context_pad_mode)
        self.axes = tuple(AxisMap(stored=self.logical_spatial_shape[a],
                                  resized=_resized_length(self.logical_spatial_shape[a], (self.scale_factors or (1.0,) * 3)[a]),
                                  pad=self.context_pad[a], pad_mode=mode, linear=linear, resizes=self.scale_factors is not None)
                          for a in range(3))
        self.transformed_spatial_shape = tuple(ax.resized for ax in self.axes)
        self.padded_spatial_shape = tuple(ax.length for ax in self.axes)

    @staticmethod
    def _parse_normalize(mode: str):
        if mode.startswith("divide-"):
            try:
                return nat.NORM_DIVIDE, float(mode.split("-", 1)[1])
            except ValueError as exc:
                raise ValueError(f"Invalid divide mode '{mode}'. Format should be 'divide-K' where K is a number "
                                 "(e.g., 'divide-255').") from exc
        if mode == "divide":
            raise ValueError("smart_normalize mode='divide' requires a non-zero divide_value (or use 'divide-K' form to embed "
                             "the divisor in the mode string).")
        if mode not in _NORM_CODES:
            raise ValueError(f"Unknown smart_normalize mode '{mode}'. Expected 'none', 'normal', '0-1', 'divide', or 'divide-K'.")
        return _NORM_CODES[mode], 1.0

    # ---- life cycle
    def close(self) -> None:
        if self.source is not None:
            self.source.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def shape(self):
        return (self.channel_count, *self.padded_spatial_shape)

    # ---- host half: raw bytes + tables
    def stage_region(self, start: Sequence[int], stop: Sequence[int], *, context: bool = True) -> StagedRegion:
        """Everything the device needs for box [start, stop) of the padded volume (`context=False`: of the resized volume without
        its context pad -- `load_full`).  Reads the raw storage box; no arithmetic on the values."""
        lo, hi = tuple(int(v) for v in start), tuple(int(v) for v in stop)
        axes = self.axes if context else tuple(AxisMap(ax.stored, ax.resized, (0, 0), "constant", ax.linear, ax.resizes)
                                               for ax in self.axes)
        tabs = [axes[a].table(lo[a], hi[a]) for a in range(3)]
        dims = tuple(hi[a] - lo[a] for a in range(3))
        used = [t[0] >= 0 for t in tabs]
        if not all(u.any() for u in used):
            return StagedRegion((lo, hi), None, str(self.source.dtype), (0, 0, 0, 0), self.channel_count, None, dims)
        raw_lo = [int(tabs[a][0][used[a]].min()) for a in range(3)]
        raw_hi = [int(tabs[a][1][used[a]].max()) + 1 for a in range(3)]
        # storage box: logical axis a lives on stored spatial axis _stored_axis[a]
        s_lo, s_hi = [0, 0, 0], [0, 0, 0]
        for a in range(3):
            s_lo[self._stored_axis[a]], s_hi[self._stored_axis[a]] = raw_lo[a], raw_hi[a]
        box = self.source.read_box(s_lo, s_hi)
        if str(box.dtype) not in nat.RAW_DTYPES:
            # a storage dtype the gather kernel has no reader for (64-bit ids, float16, bool): the one case with a host conversion
            # -- to float32, which is what the presented region holds anyway (and what the reference's reader converts to)
            box = np.ascontiguousarray(box, dtype=np.float32)
        strides = box_strides(self.source, box, self._stored_axis)
        shift = lambda t, a: np.where(t >= 0, t - raw_lo[a], -1).astype(np.int32)      # noqa: E731  box-local indices
        i0 = torch.from_numpy(np.concatenate([shift(tabs[a][0], a) for a in range(3)]))
        i1 = torch.from_numpy(np.concatenate([shift(tabs[a][1], a) for a in range(3)]))
        f = torch.from_numpy(np.concatenate([tabs[a][2] for a in range(3)]))
        raw = torch.from_numpy(box.reshape(-1).view(np.uint8))
        return StagedRegion((lo, hi), raw, str(box.dtype), strides, self.channel_count, (i0, i1, f), dims)

    # ---- device half
    def _device(self, device=None):
        dev = torch.device(device if device is not None else (self.device if self.device is not None else "cuda"))
        ops.require_device(dev, "LazyVolumeAccessor")
        if not torch.cuda.is_available():
            raise RuntimeError("LazyVolumeAccessor (pytorch_connectomics_amd) needs a CUDA(HIP) device: there is no CPU path")
        return dev

    @property
    def needs_window_statistics(self) -> bool:
        """A window's values depend on statistics of THAT window (z-score, min-max, percentile clipping)."""
        if self.kind != "image":
            return False
        return self._clips or self._norm_code in (nat.NORM_ZSCORE, nat.NORM_MINMAX)

    needs_per_patch_host_path = needs_window_statistics          # the reference-era name; nothing runs on the host any more

    @property
    def _clips(self) -> bool:
        return self.kind == "image" and self.normalize_mode != "none" and (self.clip_percentile_low > 0.0 or self.clip_percentile_high < 1.0)

    def finish_windows(self, x: torch.Tensor) -> torch.Tensor:
        """The per-window tail of the pipeline on a batch of gathered windows x fp32 (B, ...) on the device, in place:
        binarise (masks), then for images percentile clip + normalisation with each window's own statistics."""
        norm = self._norm_code if self.kind == "image" else nat.NORM_NONE
        if not self.binarize and norm == nat.NORM_NONE:
            return x
        clip = None
        if self._clips:
            if self.binarize:
                ops.window_normalize(x, binarize=True, threshold=self.threshold)
            clip = _percentile_bounds(x.reshape(x.shape[0], -1), self.clip_percentile_low, self.clip_percentile_high)
            return ops.window_normalize(x, mode=norm, divide=self._divisor, clip=clip)
        return ops.window_normalize(x, mode=norm, binarize=self.binarize, threshold=self.threshold, divide=self._divisor)

    def region_to_device(self, start, stop, device=None) -> torch.Tensor:
        """fp32 (C, *box) of the padded volume, clipped to it, un-normalised, resident on the device."""
        lo = tuple(max(0, int(v)) for v in start)
        hi = tuple(min(int(self.padded_spatial_shape[a]), int(stop[a])) for a in range(3))
        return self.stage_region(lo, hi).to_device(self._device(device), non_blocking=False)

    # ---- the reference's host-array API (device compute, one copy back)
    def read_patch(self, location, patch_size, *, outer_pad_mode: str, outer_pad_value: float) -> np.ndarray:
        """(C, *patch_size) fp32 window at `location` of the padded volume, outer-padded where it overhangs (lazy.py:852-904)."""
        start = tuple(int(v) for v in location)
        size = tuple(int(v) for v in patch_size)
        lo = tuple(max(0, start[a]) for a in range(3))
        hi = tuple(min(int(self.padded_spatial_shape[a]), start[a] + size[a]) for a in range(3))
        dev = self._device()
        if any(hi[a] <= lo[a] for a in range(3)):
            x = torch.full((1,) + size + (self.channel_count,), float(outer_pad_value), dtype=torch.float32, device=dev)
        else:
            inner = self.stage_region(lo, hi).to_device(dev, non_blocking=False)
            rel = tuple(start[a] - lo[a] for a in range(3))
            x = ops.gather_windows(inner, [rel], size, pad_mode=_gather_mode(outer_pad_mode), cval=float(outer_pad_value))
        x = self.finish_windows(x)
        return x[0].permute(3, 0, 1, 2).contiguous().cpu().numpy()

    def read_region(self, start, stop) -> np.ndarray:
        """(C, *box) fp32 of the padded volume clipped to it, with the POINTWISE finishing (binarise, divide-K) applied; modes
        that need window statistics have no region form."""
        if self.needs_window_statistics:
            raise RuntimeError("read_region: per-patch normalisation statistics need read_patch")
        reg = self.region_to_device(start, stop)
        return self.finish_windows(reg.unsqueeze(0))[0].cpu().numpy()

    def load_full(self) -> np.ndarray:
        """The whole resized volume without context pad or normalisation (masks binarised), as the reference's `load_full`."""
        full = self.stage_region((0, 0, 0), self.transformed_spatial_shape, context=False).to_device(self._device(), non_blocking=False)
        if self.binarize:
            ops.window_normalize(full.unsqueeze(0), binarize=True, threshold=self.threshold)
        return full.cpu().numpy()


def _gather_mode(mode: str) -> str:
    m = str(mode).lower()
    return {"edge": "replicate", "wrap": "circular"}.get(m, m)


def _percentile_bounds(flat: torch.Tensor, q_lo: float, q_hi: float) -> torch.Tensor:
    """np.percentile (linear interpolation) of every window, on the device: (B, n) -> float32 (B, 2).  Order statistics come from a
    device sort (torch.sort); this is the one step of the pipeline that is not a kernel of this package -- percentile clipping is a
    rarely used option and needs a selection algorithm, not arithmetic."""
    n = flat.shape[1]
    srt = torch.sort(flat, dim=1).values
    out = []
    for q in (q_lo, q_hi):
        pos = float(q) * (n - 1)
        k = int(math.floor(pos))
        frac = pos - k
        a, b = srt[:, k], srt[:, min(k + 1, n - 1)]
        out.append(a + (b - a) * frac if frac < 0.5 else b - (b - a) * (1.0 - frac))      # numpy's lerp form
    return torch.stack(out, 1).to(torch.float32).contiguous()


# ------------------------------------------------------------------------------------------------ configuration
def get_padsize(pad_size, ndim: int = 3):
    """`pad_size` in any of the config's spellings -> ((before, after),) * ndim: an int, one value, one per axis, or
    before / after per axis (reference data/processing/misc.py:20-41)."""
    if isinstance(pad_size, int):
        return ((pad_size, pad_size),) * ndim
    values = list(pad_size)
    if len(values) == 1:
        return ((values[0], values[0]),) * ndim
    if len(values) == ndim:
        return tuple((v, v) for v in values)
    if len(values) == 2 * ndim:
        return tuple((values[2 * i], values[2 * i + 1]) for i in range(ndim))
    raise ValueError(f"pad_size length must be 1, {ndim}, or {2 * ndim}, got {len(values)}")


def _resolve_scale_factors(cfg, *, kind: str, mode: str):
    """Per-axis resize factors of the test volume (reference lazy.py:420-453): `data_transform.resize` relative to the patch size
    in test / tune mode, else the per-kind `resize` factors."""
    data = cfg.data
    shared = getattr(data, "data_transform", None)
    target = getattr(shared, "resize", None)
    if mode in {"test", "tune"} and target:
        patch = getattr(getattr(data, "dataloader", None), "patch_size", None)
        if patch and len(patch) == len(target) and all(float(v) > 0 for v in patch):
            return tuple(float(o) / float(i) for o, i in zip(target, patch))
        raise ValueError("Lazy sliding-window inference requires data.dataloader.patch_size when "
                         "data_transform.resize is configured.")
    holder = {"image": getattr(data, "image_transform", None), "label": getattr(data, "image_transform", None),
              "mask": getattr(data, "mask_transform", None) or shared}.get(kind)
    factors = getattr(holder, "resize", None)
    return tuple(float(v) for v in factors) if factors else None


def build_accessor(cfg, path: str, *, kind: str, mode: str = "test", device=None) -> LazyVolumeAccessor:
    """The accessor for `path` under the config's test-time transforms (reference lazy.py:920-959)."""
    data = cfg.data
    shared = getattr(data, "data_transform", None)
    image = getattr(data, "image_transform", None)
    pad = get_padsize(getattr(shared, "pad_size", [0, 0, 0]) or [0, 0, 0], ndim=3)
    options = dict(normalize_mode="none", clip_percentile_low=0.0, clip_percentile_high=1.0, binarize=False, threshold=0.0)
    if kind == "image":
        options.update(normalize_mode=getattr(image, "normalize", "none") or "none",
                       clip_percentile_low=float(getattr(image, "clip_percentile_low", 0.0)),
                       clip_percentile_high=float(getattr(image, "clip_percentile_high", 1.0)))
    elif kind == "mask":
        mask = getattr(data, "mask_transform", None) or shared
        options.update(binarize=bool(getattr(mask, "binarize", False)), threshold=float(getattr(mask, "threshold", 0.0)))
    return LazyVolumeAccessor(
        path, kind=kind, transpose_axes=getattr(shared, "val_transpose", None) or (),
        scale_factors=_resolve_scale_factors(cfg, kind=kind, mode=mode),
        context_pad=pad if kind in {"image", "mask"} else ((0, 0),) * 3,
        context_pad_mode=getattr(shared, "pad_mode", "reflect") if kind == "image" else "constant",
        tile_read_workers=max(1, int(getattr(getattr(cfg, "system", None), "num_workers", 1) or 1)), device=device, **options)


def load_lazy_volume(cfg, path: str, *, kind: str, mode: str = "test") -> np.ndarray:
    with build_accessor(cfg, path, kind=kind, mode=mode) as acc:
        return acc.load_full()


# ------------------------------------------------------------------------------------------------ read-ahead
class RegionPrefetcher:
    """Stages the regions of an accessor one ahead of the consumer on an IO thread: while the GPU predicts chunk i, the disk read
    and decompression of chunk i+1's RAW box land in pinned memory.  `get()` hands over a `StagedRegion`;
    its `to_device` is one H2D copy of the stored bytes and one kernel."""

    def __init__(self, accessor: LazyVolumeAccessor, regions, *, pin: bool = True):
        self.acc, self.regions, self.pin = accessor, list(regions), bool(pin)
        self._next = 0
        self._slot = None
        self._err: Optional[BaseException] = None
        self._thread: Optional[threading.Thread] = None
        self._kick()

    def _stage(self, region) -> None:
        try:
            lo = tuple(max(0, int(v)) for v in region[0])
            hi = tuple(min(int(self.acc.padded_spatial_shape[a]), int(region[1][a])) for a in range(3))
            staged = self.acc.stage_region(lo, hi)
            self._slot = (region, staged.pin() if self.pin else staged)
        except BaseException as e:      # surfaced by get()
            self._err = e

    def _kick(self) -> None:
        self._thread = None
        if self._next < len(self.regions):
            self._thread = threading.Thread(target=self._stage, args=(self.regions[self._next],), name="pytc-region-prefetch", daemon=True)
            self._thread.start()

    def get(self):
        """-> (region, StagedRegion) of the next region; starts staging the one after it."""
        if self._thread is None:
            raise StopIteration
        self._thread.join()
        if self._err is not None:
            raise self._err
        out, self._slot = self._slot, None
        self._next += 1
        self._kick()
        return out


__all__ = ["LazyVolumeAccessor", "AxisMap", "StagedRegion", "ZarrV2Array", "build_accessor", "load_lazy_volume", "get_padsize",
           "RegionPrefetcher", "detect_format"]
