"""Oracle: the device half of the disk-backed volume reader, restated in numpy.

TEST INFRASTRUCTURE (see oracle/__init__.py): only tests import this.  It executes a `StagedRegion` (raw storage bytes + per-axis
index tables, pytorch_connectomics_amd/inference/lazy_accessor.py) the way `pytc_resample_region` does, and applies the per-window
finishing of the reference's `LazyVolumeAccessor.read_patch` (connectomics/inference/lazy.py:852-904) + `smart_normalize`
(connectomics/data/augmentation/augment_ops.py:552-611: percentile clip, then 'none' | 'normal' | '0-1' | 'divide-K').

PINNED by tests/golden/lazy_accessor.npz -- 26 arrays produced by the REFERENCE's accessor (make_golden.py --accessor) -- in
tests/test_host_lazy_accessor.py; the HIP kernels are then compared with the same fixtures in tests/test_gpu_lazy_accessor.py.
"""
from __future__ import annotations

import numpy as np


def execute_staged(staged) -> np.ndarray:
    """(C, nz, ny, nx) fp32: out = trilinear / nearest blend of the raw box through the tables (i0 < 0 -> 0)."""
    nz, ny, nx = staged.dims
    C = staged.channels
    if staged.raw is None:
        return np.zeros((C, nz, ny, nx), np.float32)
    raw = np.frombuffer(staged.raw.numpy().tobytes(), dtype=np.dtype(staged.raw_dtype)).astype(np.float32)
    sc, sz, sy, sx = (int(v) for v in staged.strides)
    i0, i1, f = (t.numpy() for t in staged.tables)
    tabs = []
    for lo, n in ((0, nz), (nz, ny), (nz + ny, nx)):
        tabs.append((i0[lo:lo + n].astype(np.int64), i1[lo:lo + n].astype(np.int64), f[lo:lo + n].astype(np.float32)))
    (z0, z1, fz), (y0, y1, fy), (x0, x1, fx) = tabs
    valid = (z0 >= 0)[:, None, None] & (y0 >= 0)[None, :, None] & (x0 >= 0)[None, None, :]
    cl = lambda a: np.maximum(a, 0)          # noqa: E731

    def at(zi, yi, xi):
        off = (cl(zi) * sz)[:, None, None] + (cl(yi) * sy)[None, :, None] + (cl(xi) * sx)[None, None, :]
        return np.stack([raw[c * sc + off] for c in range(C)])

    gx0, gy0, gz0 = (np.float32(1) - fx)[None, None, None, :], (np.float32(1) - fy)[None, None, :, None], (np.float32(1) - fz)[None, :, None, None]
    wx, wy, wz = fx[None, None, None, :], fy[None, None, :, None], fz[None, :, None, None]
    r00 = at(z0, y0, x0) * gx0 + at(z0, y0, x1) * wx
    r01 = at(z0, y1, x0) * gx0 + at(z0, y1, x1) * wx
    r10 = at(z1, y0, x0) * gx0 + at(z1, y0, x1) * wx
    r11 = at(z1, y1, x0) * gx0 + at(z1, y1, x1) * wx
    out = (r00 * gy0 + r01 * wy) * gz0 + (r10 * gy0 + r11 * wy) * wz
    return (out * valid[None]).astype(np.float32)


def smart_normalize(volume: np.ndarray, mode: str, clip_low: float = 0.0, clip_high: float = 1.0) -> np.ndarray:
    """augment_ops.py:552-611."""
    divide = None
    if mode.startswith("divide-"):
        divide, mode = float(mode.split("-", 1)[1]), "divide"
    v = volume.copy()
    if clip_low > 0.0 or clip_high < 1.0:
        v = np.clip(v, np.percentile(v, clip_low * 100), np.percentile(v, clip_high * 100))
    if mode == "normal":
        mean, std = v.mean(), v.std()
        if std > 1e-8:
            v = (v - mean) / std
    elif mode == "0-1":
        lo, hi = v.min(), v.max()
        if hi > lo:
            v = (v - lo) / (hi - lo)
    elif mode == "divide":
        v = v / divide
    elif mode != "none":
        raise ValueError(mode)
    return v


def finish(acc, patch: np.ndarray) -> np.ndarray:
    """Per-window tail: binarise (masks), then smart_normalize for images (lazy.py:896-904)."""
    if acc.binarize:
        patch = (patch > acc.threshold).astype(np.float32)
    if acc.kind == "image" and acc.normalize_mode != "none":
        patch = smart_normalize(patch, acc.normalize_mode, acc.clip_percentile_low, acc.clip_percentile_high)
    return patch.astype(np.float32)


def read_patch(acc, location, size, *, outer_pad_mode: str, outer_pad_value: float) -> np.ndarray:
    start = tuple(int(v) for v in location)
    end = tuple(start[a] + int(size[a]) for a in range(3))
    lo = tuple(max(0, start[a]) for a in range(3))
    hi = tuple(min(int(acc.padded_spatial_shape[a]), end[a]) for a in range(3))
    inner = execute_staged(acc.stage_region(lo, hi))
    pads = [(0, 0)] + [(lo[a] - start[a], end[a] - hi[a]) for a in range(3)]
    mode = {"replicate": "edge", "circular": "wrap"}.get(str(outer_pad_mode).lower(), str(outer_pad_mode).lower())
    if any(b or a for b, a in pads):
        if mode == "constant":
            inner = np.pad(inner, pads, mode="constant", constant_values=outer_pad_value)
        else:
            if mode == "reflect" and any(s <= 1 for s in inner.shape[1:]):
                mode = "edge"
            inner = np.pad(inner, pads, mode=mode)
    return finish(acc, inner)


def load_full(acc) -> np.ndarray:
    full = execute_staged(acc.stage_region((0, 0, 0), acc.transformed_spatial_shape, context=False))
    return (full > acc.threshold).astype(np.float32) if acc.binarize else full
