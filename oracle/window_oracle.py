"""Oracle: sliding-window planner, blending maps and eager overlap-add engine.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Plain Python integers for the grid, numpy for
the maps, and a straightforward per-window loop for the engine.  Each function cites the
reference lines it restates (paths relative to /root/reference/connectomics/).
"""
from __future__ import annotations

import itertools
import math
from typing import Callable, Sequence

import numpy as np
import torch

DISTANCE_MODES = {"distance", "distance_transform", "distance-transform",
                  "distance_transform_cdt", "banis", "banis_distance"}  # inference/window.py:30-37


def scan_interval(image_size: Sequence[int], roi: Sequence[int], overlap) -> tuple[int, ...]:
    """inference/window.py:57-89 -- stride = max(1, round(roi*(1-ov))), or img when img <= roi."""
    nd = len(roi)
    ovs = [float(o) for o in overlap] if isinstance(overlap, (list, tuple)) else [float(overlap)] * nd
    out = []
    for a in range(nd):
        ov = min(max(ovs[a], 0.0), 0.99)
        if int(image_size[a]) <= int(roi[a]):
            out.append(int(image_size[a]))
        else:
            out.append(max(1, int(round(int(roi[a]) * (1.0 - ov)))))
    return tuple(out)


def window_starts(image_size: Sequence[int], roi: Sequence[int], interval: Sequence[int]) -> list[tuple[int, ...]]:
    """inference/window.py:92-134 -- per-axis starts 0, s, 2s.. plus a snapped last start
    img-roi; one start (0) when img <= roi; row-major product (first axis outermost)."""
    per_axis = []
    for a in range(len(roi)):
        img, r, s = int(image_size[a]), int(roi[a]), max(1, int(interval[a]))
        if img <= r:
            per_axis.append([0])
            continue
        st = list(range(0, img - r + 1, s))
        if st[-1] != img - r:
            st.append(img - r)
        per_axis.append(st)
    return [tuple(p) for p in itertools.product(*per_axis)]


def _bump_axis(n: int, dtype=np.float32) -> np.ndarray:
    """inference/window.py:174-187 -- per-axis Wu bump, peak-normalised, evaluated in `dtype`."""
    tiny = np.finfo(dtype).tiny
    idx = np.arange(n, dtype=dtype)
    u = (idx + dtype(1.0)) / dtype(n + 1.0) * dtype(2.0) - dtype(1.0)
    den = np.maximum(dtype(1.0) - u * u, tiny)
    k = np.exp(dtype(-1.0) / den).astype(dtype)
    return (k / np.maximum(k.max(), tiny)).astype(dtype)


def importance_axes(roi: Sequence[int], mode: str) -> list[np.ndarray] | None:
    """Separable per-axis factors of the map (constant / bump); None for distance modes."""
    mode = str(mode).strip().lower()
    if mode == "constant":
        return [np.ones(int(n), np.float32) for n in roi]
    if mode == "bump":
        return [_bump_axis(int(n)) for n in roi]
    return None


def importance_map(roi: Sequence[int], mode: str, min_value: float = 1e-5) -> np.ndarray:
    """inference/window.py:137-243 (fp32).  constant -> ones; bump -> product of per-axis bumps,
    clamp tiny, then clamp_min(min_value); distance* -> min over axes of min(i+1, n-i)."""
    mode = str(mode).strip().lower()
    roi = tuple(int(v) for v in roi)
    if any(v <= 0 for v in roi):
        raise ValueError(f"roi_size must contain positive values, got {roi}.")
    if mode in DISTANCE_MODES:
        out = None
        for a, n in enumerate(roi):
            c = np.arange(n, dtype=np.float32)
            d = np.minimum(c + 1, np.float32(n) - c)
            shape = [1] * len(roi)
            shape[a] = n
            d = d.reshape(shape)
            out = d if out is None else np.minimum(out, d)
        return np.broadcast_to(out, roi).astype(np.float32).copy()
    axes = importance_axes(roi, mode)
    if axes is None:
        raise ValueError(f"unsupported blending mode {mode!r}")
    out = None
    for a, k in enumerate(axes):
        shape = [1] * len(roi)
        shape[a] = roi[a]
        out = k.reshape(shape) if out is None else out * k.reshape(shape)
    out = np.broadcast_to(out, roi).astype(np.float32)
    if mode == "bump":
        out = np.maximum(out, np.finfo(np.float32).tiny)
    if min_value > 0:
        out = np.maximum(out, np.float32(min_value))
    return np.ascontiguousarray(out)


def normalize_accumulator(value: np.ndarray, weight: np.ndarray) -> np.ndarray:
    """inference/window.py:275-294 -- value / max(weight, 1e-4), dtype of value preserved."""
    clamp = 1e-4
    if value.dtype == np.float16:
        clamp = max(clamp, float(np.finfo(np.float16).tiny))
    div = np.maximum(weight, np.asarray(clamp, dtype=weight.dtype)).astype(value.dtype)
    return (value / div).astype(value.dtype)


def extract_window(vol: torch.Tensor, start: Sequence[int], roi: Sequence[int],
                   padding_mode: str = "constant", cval: float = 0.0) -> torch.Tensor:
    """inference/window.py:464-527 for ONE window: slice the in-volume part, pad the rest with
    `padding_mode`; reflect/circular fall back to constant when a pad >= the inner extent."""
    nd = len(roi)
    img = vol.shape[-nd:]
    lo = [max(0, int(start[a])) for a in range(nd)]
    hi = [min(int(img[a]), int(start[a]) + int(roi[a])) for a in range(nd)]
    inner = vol[(slice(None), slice(None)) + tuple(slice(lo[a], hi[a]) for a in range(nd))]
    before = [max(0, -int(start[a])) for a in range(nd)]
    after = [max(0, int(start[a]) + int(roi[a]) - int(img[a])) for a in range(nd)]
    if any(before) or any(after):
        mode = padding_mode
        if mode in ("reflect", "circular"):
            if any(before[a] >= inner.shape[2 + a] or after[a] >= inner.shape[2 + a] for a in range(nd)):
                mode = "constant"
        pad = []
        for a in reversed(range(nd)):
            pad += [before[a], after[a]]
        if mode == "constant":
            inner = torch.nn.functional.pad(inner, pad, mode="constant", value=cval)
        else:
            inner = torch.nn.functional.pad(inner, pad, mode=mode)
    return inner


def eager_sliding_window(vol: torch.Tensor, network: Callable[[torch.Tensor], torch.Tensor], *,
                         roi: Sequence[int], overlap=0.5, mode: str = "bump",
                         sw_batch_size: int = 1, padding_mode: str = "constant",
                         cval: float = 0.0) -> torch.Tensor:
    """inference/window.py:563-683 -- grow-to-roi constant pad, probe window 0, then batches of
    `sw_batch_size`; value += pred*w, weight += w; normalise; crop back.  CPU fp32/whatever the
    network returns.  Batch size must be 1 (window.py:573-577)."""
    nd = len(roi)
    if vol.dim() < nd + 2:
        raise ValueError("inputs must have shape (B, C, *spatial)")
    if vol.shape[0] != 1:
        raise ValueError("eager sliding window expects batch size 1")
    orig = tuple(int(v) for v in vol.shape[-nd:])
    grow = [max(0, int(roi[a]) - orig[a]) for a in range(nd)]
    if any(grow):
        pad = []
        for a in reversed(range(nd)):
            pad += [0, grow[a]]
        vol = torch.nn.functional.pad(vol, pad, mode="constant", value=cval)
    img = tuple(int(v) for v in vol.shape[-nd:])
    starts = window_starts(img, roi, scan_interval(img, roi, overlap))

    def run(batch_starts):
        xs = torch.cat([extract_window(vol, s, roi, padding_mode, cval) for s in batch_starts], 0)
        with torch.no_grad():
            y = network(xs)
        if not isinstance(y, torch.Tensor):
            raise ValueError("`network` must return a torch.Tensor")
        return y

    probe = run(starts[:1])
    c_out, dt = int(probe.shape[1]), probe.dtype
    wmap = torch.from_numpy(importance_map(roi, mode)).to(dt)
    val = torch.zeros((1, c_out) + img, dtype=dt)
    wgt = torch.zeros((1, 1) + img, dtype=dt)

    def acc(pred, s):
        sl = (slice(None), slice(None)) + tuple(slice(s[a], s[a] + int(roi[a])) for a in range(nd))
        val[sl] += pred.to(dt) * wmap
        wgt[sl] += wmap

    acc(probe[0:1], starts[0])
    rest = starts[1:]
    for b in range(0, len(rest), max(1, int(sw_batch_size))):
        chunk = rest[b:b + max(1, int(sw_batch_size))]
        out = run(chunk)
        for i, s in enumerate(chunk):
            acc(out[i:i + 1], s)
    clamp = 1e-4
    if dt == torch.float16:
        clamp = max(clamp, float(torch.finfo(torch.float16).tiny))
    val /= torch.clamp_min(wgt, clamp).to(dt)
    if any(grow):
        val = val[(slice(None), slice(None)) + tuple(slice(0, orig[a]) for a in range(nd))].contiguous()
    return val


# --------------------------------------------------------------------------- chunk grid
def chunk_grid(volume_shape: Sequence[int], chunk_shape: Sequence[int]):
    """chunked/chunk_grid.py:32-43 -- ceil-div counts, row-major product, last chunk clipped.
    Returns list of (index, start, stop)."""
    counts = [int(math.ceil(int(v) / int(c))) for v, c in zip(volume_shape, chunk_shape)]
    out = []
    for idx in itertools.product(*[range(n) for n in counts]):
        start = tuple(int(i) * int(c) for i, c in zip(idx, chunk_shape))
        stop = tuple(min(int(s) + int(c), int(v)) for s, c, v in zip(start, chunk_shape, volume_shape))
        out.append((tuple(int(i) for i in idx), start, stop))
    return out


def halo_region(start, stop, halo, volume_shape, crop_before=(0, 0, 0)):
    """chunked/halo.py:12-40 -- (core shifted by crop_before) +- halo, clipped to the input
    volume; returns (read_start, read_stop, core_start_in_read, core_stop_in_read)."""
    cs = tuple(int(s) + int(c) for s, c in zip(start, crop_before))
    ce = tuple(int(e) + int(c) for e, c in zip(stop, crop_before))
    rs = tuple(max(0, s - int(h)) for s, h in zip(cs, halo))
    re_ = tuple(min(int(v), e + int(h)) for e, h, v in zip(ce, halo, volume_shape))
    lo = tuple(s - r for s, r in zip(cs, rs))
    hi = tuple(e - r for e, r in zip(ce, rs))
    return rs, re_, lo, hi


# --------------------------------------------------------------------------- lazy / region engine
def lazy_axis_offsets(image_size, roi, overlap, snap_to_edge=False):
    """inference/lazy.py:269-334 -- per-axis window offsets incl. the face-centred boundary windows
    (from -border_pad to img-roi+border_pad, border_pad = roi - stride); snap_to_edge uses int() not round()."""
    nd = 3
    ovs = [float(o) for o in overlap] if isinstance(overlap, (list, tuple)) else [float(overlap)] * nd
    if snap_to_edge:
        strides = [max(1, int(int(roi[a]) * (1.0 - ovs[a]))) for a in range(nd)]
    else:
        strides = list(scan_interval(image_size, roi, ovs))
    out = []
    for a in range(nd):
        img, r, s = int(image_size[a]), int(roi[a]), max(1, int(strides[a]))
        if img <= r:
            out.append([0])
            continue
        bp = max(0, r - s)
        lo, hi = -bp, img - r + bp
        offs = list(range(lo, hi + 1, s))
        if offs[-1] != hi:
            offs.append(hi)
        out.append(offs)
    return out


def lazy_sliding_window(vol: np.ndarray, network, *, roi, overlap=0.5, mode="bump", sw_batch_size=1,
                        padding_mode="reflect", cval=0.0, snap_to_edge=False, target_context=(0, 0, 0),
                        border_mask=(0, 0, 0), region=None, window_post=None) -> torch.Tensor:
    """inference/lazy.py:986-1258 for a numpy (C,Z,Y,X) volume: global-grid windows intersecting `region`
    ((start, stop) or None), np.pad outer padding, `window_post` (activation / channel select) applied to
    every window prediction BEFORE blending, only the intersection is accumulated, then normalisation."""
    C = vol.shape[0]
    bounds = tuple(int(v) for v in vol.shape[1:])
    start, stop = ((0, 0, 0), bounds) if region is None else (tuple(region[0]), tuple(min(bounds[a], region[1][a]) for a in range(3)))
    ctx = tuple(int(v) for v in target_context)
    offs = lazy_axis_offsets(bounds, roi, overlap, snap_to_edge)
    keep = [[o for o in offs[a] if o < stop[a] and o + int(roi[a]) > start[a]] for a in range(3)]
    wins = list(itertools.product(*keep))
    wmap = importance_map(roi, mode)
    bm = [int(b) for b in border_mask]
    if any(bm):
        wmap = wmap.copy()
        for a, k in enumerate(bm):
            if k > 0:
                idx = [slice(None)] * 3
                idx[a] = slice(0, k); wmap[tuple(idx)] = 0
                idx[a] = slice(wmap.shape[a] - k, None); wmap[tuple(idx)] = 0
    wmap_t = torch.from_numpy(wmap)
    np_mode = {"replicate": "edge", "circular": "wrap"}.get(padding_mode, padding_mode)
    out_size = tuple(stop[a] - start[a] for a in range(3))
    val = None
    wgt = torch.zeros((1, 1) + out_size)

    def read(w):
        s = [w[a] - ctx[a] for a in range(3)]
        e = [w[a] + int(roi[a]) + ctx[a] for a in range(3)]
        lo = [max(0, s[a]) for a in range(3)]
        hi = [min(bounds[a], e[a]) for a in range(3)]
        inner = vol[:, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
        pads = [(0, 0)] + [(max(0, -s[a]), max(0, e[a] - bounds[a])) for a in range(3)]
        if not any(p != (0, 0) for p in pads):
            return inner
        if np_mode == "constant":
            return np.pad(inner, pads, mode="constant", constant_values=cval)
        return np.pad(inner, pads, mode=np_mode)

    for b0 in range(0, len(wins), max(1, sw_batch_size)):
        chunk = wins[b0:b0 + max(1, sw_batch_size)]
        x = torch.from_numpy(np.stack([read(w) for w in chunk]).astype(np.float32))
        with torch.no_grad():
            p = network(x)
        if any(ctx):
            p = p[:, :, ctx[0]:ctx[0] + roi[0], ctx[1]:ctx[1] + roi[1], ctx[2]:ctx[2] + roi[2]]
        if window_post is not None:
            p = window_post(p)
        if val is None:
            val = torch.zeros((1, p.shape[1]) + out_size)
        for i, w in enumerate(chunk):
            ilo = [max(w[a], start[a]) for a in range(3)]
            ihi = [min(w[a] + int(roi[a]), stop[a]) for a in range(3)]
            ps = tuple(slice(ilo[a] - w[a], ihi[a] - w[a]) for a in range(3))
            os_ = tuple(slice(ilo[a] - start[a], ihi[a] - start[a]) for a in range(3))
            val[(slice(None), slice(None)) + os_] += p[(slice(i, i + 1), slice(None)) + ps] * wmap_t[ps]
            wgt[(slice(None), slice(None)) + os_] += wmap_t[ps]
    val /= torch.clamp_min(wgt, 1e-4)
    return val
