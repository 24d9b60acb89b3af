"""Region / whole-volume sliding-window inference from the GLOBAL window grid -- counterpart of the engine
half of the reference's connectomics/inference/lazy.py (_snap_offsets :269-284, _build_window_axis_offsets
:307-334, _build_intersecting_window_slices :337-365, _resolve_target_context :368-386,
_lazy_sliding_window :986-1258, lazy_predict_region :1261, lazy_predict_volume :1295).

Semantics kept: boundary windows centred on the volume faces (offsets from -border_pad, border_pad =
roi - stride), snap_to_edge stride = int(roi*(1-overlap)), per-window outer padding by `padding_mode`,
activation / channel selection applied to each window BEFORE blending, optional target_context (predict
roi + 2*ctx, keep the centre) and border_mask, only the intersection with the requested region is
accumulated -- so a chunk's prediction equals the matching slice of the whole-volume prediction.

MI355X design: the source volume (numpy / memmap / tensor; host or device) is brought to HBM once per region
(the bounding box of the intersecting windows), windows are gathered / blended by the HIP kernels and the
fp32 accumulators never leave the device.  Disk-backed sources go through inference/lazy_accessor.py (`LazyVolumeAccessor`
over HDF5 / .npy / zarr v2 with the reference's lazy transforms): a path or an accessor may be passed as `volume`; the region
box's RAW storage bytes are staged once (`stage_region`, optionally read ahead into pinned memory by the chunked runner), turned
into the transformed, context-padded fp32 region by one kernel, the windows are gathered from it on the device, and the per-window
tail of the reader's pipeline (binarise, percentile clip, z-score / min-max / divide-K with each window's own statistics) runs on
the gathered batch (`LazyVolumeAccessor.finish_windows`).  Nothing is resampled or normalised on the host.
"""
from __future__ import annotations

import itertools
import logging
from typing import Callable, Optional, Sequence

import numpy as np
import torch

from .. import _native as nat
from .. import hip_ops as ops
from ..utils.channel_slices import resolve_channel_indices
from ..utils.model_outputs import get_inference_channel_activations, get_inference_select_channel, select_output_tensor
from .lazy_accessor import LazyVolumeAccessor, build_accessor, load_lazy_volume
from .lazy_distributed import (distributed_context, is_distributed_window_sharding_enabled, make_accumulator_reduce_hook,
                               validate_distributed_patch_shard)
from .window import (_axis_kernels, compute_scan_interval, pipeline_streams_for, resolve_border_mask,
                     resolve_inferer_overlap, resolve_inferer_roi_size, resolve_model_output_dtype)

logger = logging.getLogger(__name__)


def _coerce_overlap(overlap, spatial_dims: int) -> tuple[float, ...]:
    if isinstance(overlap, (list, tuple)):
        if len(overlap) != spatial_dims:
            raise ValueError(f"Overlap rank mismatch: expected {spatial_dims} values, got {overlap}.")
        return tuple(float(v) for v in overlap)
    return tuple(float(overlap) for _ in range(spatial_dims))


def _snap_offsets(image_size: int, roi_size: int, stride: int, *, border_pad: int = 0) -> list[int]:
    if image_size <= roi_size:
        return [0]
    stride = max(1, stride)
    lo, hi = -int(border_pad), image_size - roi_size + int(border_pad)
    offs = list(range(lo, hi + 1, stride))
    if not offs or offs[-1] != hi:
        offs.append(hi)
    return offs


def _build_window_axis_offsets(image_size, roi_size, overlap, *, snap_to_edge: bool) -> list[list[int]]:
    if snap_to_edge:
        strides = tuple(max(1, int(int(roi_size[a]) * (1.0 - float(overlap[a])))) for a in range(3))
    else:
        strides = compute_scan_interval(tuple(int(v) for v in image_size), tuple(int(v) for v in roi_size),
                                        overlap=tuple(float(v) for v in overlap))
    return [_snap_offsets(int(image_size[a]), int(roi_size[a]), int(strides[a]),
                          border_pad=max(0, int(roi_size[a]) - int(strides[a]))) for a in range(3)]


def _build_window_slices(image_size, roi_size, overlap, *, snap_to_edge: bool):
    offs = _build_window_axis_offsets(image_size, roi_size, overlap, snap_to_edge=snap_to_edge)
    return [tuple(slice(int(s[a]), int(s[a]) + int(roi_size[a])) for a in range(3)) for s in itertools.product(*offs)]


def _build_intersecting_window_slices(image_size, roi_size, overlap, *, region_start, region_stop, snap_to_edge: bool):
    offs = _build_window_axis_offsets(image_size, roi_size, overlap, snap_to_edge=snap_to_edge)
    keep = [[o for o in offs[a] if o < int(region_stop[a]) and o + int(roi_size[a]) > int(region_start[a])]
            for a in range(3)]
    return [tuple(slice(int(s[a]), int(s[a]) + int(roi_size[a])) for a in range(3)) for s in itertools.product(*keep)]


def _resolve_target_context(sliding_cfg, roi_size) -> tuple[int, int, int]:
    ctx = list(getattr(sliding_cfg, "target_context", []) or [])
    if not ctx:
        return (0, 0, 0)
    if len(ctx) == 1:
        ctx = ctx * 3
    if len(ctx) != 3:
        raise ValueError(f"inference.sliding_window.target_context must have length 1 or 3, got {ctx}.")
    ctx = tuple(int(v) for v in ctx)
    if any(v < 0 for v in ctx):
        raise ValueError(f"inference.sliding_window.target_context values must be non-negative, got {ctx}.")
    return ctx


def get_lazy_image_reference_shape(cfg, image_path=None, *, mode: str = "test") -> tuple:
    """The reference's `get_lazy_image_reference_shape(cfg, image_path)` (lazy.py:962-978): (1, C, *padded spatial shape) of the
    test volume under the config's test-time transforms, refusing a transformed volume smaller than `data.dataloader.patch_size`.
    This package's earlier form -- `(volume, cfg=None)` with an array / accessor / path first -> the ZYX extent -- is recognised by
    its argument types and kept."""
    def is_source(v):
        return isinstance(v, (str, bytes, LazyVolumeAccessor)) or hasattr(v, "__fspath__") or hasattr(v, "shape")
    if is_source(cfg) and not is_source(image_path):
        volume, cfg = cfg, image_path
        if isinstance(volume, (str, bytes)) or hasattr(volume, "__fspath__"):
            with open_lazy_source(cfg, volume) as acc:
                return tuple(int(v) for v in acc.padded_spatial_shape)
        return tuple(int(v) for v in (volume.padded_spatial_shape if isinstance(volume, LazyVolumeAccessor) else volume.shape[-3:]))
    with build_accessor(_DefaultDataCfg(cfg), str(image_path), kind="image", mode=mode) as acc:
        patch = getattr(getattr(getattr(cfg, "data", None), "dataloader", None), "patch_size", None)
        if patch and any(acc.transformed_spatial_shape[a] < int(patch[a]) for a in range(3)):
            raise ValueError("Lazy sliding-window inference currently requires the transformed test volume to be at least as large "
                             f"as data.dataloader.patch_size in every axis. Got transformed_shape={acc.transformed_spatial_shape}, "
                             f"patch_size={tuple(int(v) for v in patch)}.")
        return (1, int(acc.channel_count), *(int(v) for v in acc.padded_spatial_shape))


class _DefaultDataCfg:
    """cfg.data view with the reference's defaults for configs that carry no data section (tests, array sources)."""

    def __init__(self, cfg):
        from types import SimpleNamespace as NS
        d = getattr(cfg, "data", None)
        self.system = getattr(cfg, "system", None) or NS(num_workers=1)
        self.data = NS(dataloader=getattr(d, "dataloader", None) or NS(patch_size=None),
                       data_transform=getattr(d, "data_transform", None) or NS(val_transpose=None, pad_size=[0, 0, 0],
                                                                               pad_mode="reflect", resize=None),
                       image_transform=getattr(d, "image_transform", None) or NS(normalize="none", clip_percentile_low=0.0,
                                                                                 clip_percentile_high=1.0, resize=None),
                       mask_transform=getattr(d, "mask_transform", None))


def open_lazy_source(cfg, volume):
    """path -> LazyVolumeAccessor under the config's test-time transforms; accessors pass through."""
    if isinstance(volume, LazyVolumeAccessor):
        return volume
    return build_accessor(_DefaultDataCfg(cfg), str(volume if not isinstance(volume, bytes) else volume.decode()), kind="image",
                          mode="test")


def _as_channel_first(volume):
    if volume.ndim == 3:
        return volume[None]
    if volume.ndim == 5 and volume.shape[0] == 1:
        return volume[0]
    if volume.ndim != 4:
        raise ValueError(f"volume must be (Z,Y,X), (C,Z,Y,X) or (1,C,Z,Y,X); got shape {tuple(volume.shape)}")
    return volume


def _window_preprocess(cfg, pred_cl: torch.Tensor) -> torch.Tensor:
    """Activations + channel selection on a channels-last window batch (lazy path: BEFORE blending)."""
    pred_cl = pred_cl.float() if pred_cl.dtype != torch.float32 else pred_cl
    C = int(pred_cl.shape[-1])
    for entry in get_inference_channel_activations(cfg):
        ch = entry.get("channels", ":") if isinstance(entry, dict) else getattr(entry, "channels", ":")
        act = entry.get("activation") if isinstance(entry, dict) else getattr(entry, "activation", None)
        idx = resolve_channel_indices(ch, num_channels=C, context="inference.model.channel_activations channels")
        if act is None or (isinstance(act, str) and act.lower() == "none"):
            continue
        scale = 1.0
        if act == "sigmoid":
            code = nat.ACT_SIGMOID
        elif isinstance(act, str) and act.startswith("scale_sigmoid"):
            code, scale = nat.ACT_SIGMOID, (float(act.split(":", 1)[1]) if ":" in act else 0.2)
        elif act == "tanh":
            code = nat.ACT_TANH
        elif act == "softmax":
            if len(idx) <= 1:
                continue
            code = nat.ACT_SOFTMAX
        else:
            raise ValueError(f"Unknown activation '{act}' for channels {idx}.")
        contiguous = idx == list(range(idx[0], idx[-1] + 1))
        groups = [(idx[0], idx[-1] + 1)] if contiguous else [(c, c + 1) for c in idx]
        if code == nat.ACT_SOFTMAX and not contiguous:
            raise NotImplementedError("softmax over a non-contiguous channel list is not supported on device")
        for a, b in groups:
            ops.channel_activation(pred_cl, a, b, code, scale, channels_last=True)
    sel = get_inference_select_channel(cfg)
    if sel is not None:
        idx = resolve_channel_indices(sel, num_channels=C, context="inference.model.select_channel")
        if idx != list(range(C)):
            pred_cl = pred_cl[..., idx].contiguous()
    return pred_cl


def _require_window_shape(prediction: torch.Tensor, read, roi, ctx, *, channels_last: bool = False) -> None:
    """The window contract of the lazy loop (reference lazy.py:389-419): the network keeps the spatial shape of what it was given --
    the ROI, or ROI + 2 * target_context, of which the loop then keeps the centre.  Messages in the caller's (N, C, Z, Y, X) order."""
    shape = tuple(int(v) for v in prediction.shape)
    if channels_last and len(shape) == 5:
        shape = (shape[0], shape[4]) + shape[1:4]
    if shape[2:] == tuple(read):
        return
    scope = "Lazy sliding-window inference"
    if any(ctx):
        raise RuntimeError(f"{scope} with target_context={tuple(ctx)} expected prediction spatial shape {tuple(read)}, got {shape[2:]}.")
    raise RuntimeError(f"{scope} requires model predictions to have the same spatial shape as the sliding-window ROI. "
                       f"Got prediction.shape={shape} and roi_size={tuple(roi)}.")


def _crop_prediction_to_roi(prediction: torch.Tensor, *, roi_size, target_context, scope: str = "Lazy sliding-window inference") -> torch.Tensor:
    """The centre ROI of a window prediction (N, C, Z, Y, X) made with `target_context` voxels of extra context per side -- the
    reference's helper under its name (lazy.py:389-419); the lazy loop itself does the same on channels-last batches."""
    roi = tuple(int(v) for v in roi_size)
    ctx = tuple(int(v) for v in target_context)
    read = tuple(r + 2 * c for r, c in zip(roi, ctx)) if any(ctx) else roi
    shape = tuple(int(v) for v in prediction.shape)
    if shape[2:] != read:
        if any(ctx):
            raise RuntimeError(f"{scope} with target_context={ctx} expected prediction spatial shape {read}, got {shape[2:]}.")
        raise RuntimeError(f"{scope} requires model predictions to have the same spatial shape as the sliding-window ROI. "
                           f"Got prediction.shape={shape} and roi_size={roi}.")
    if not any(ctx):
        return prediction
    return prediction[(slice(None), slice(None)) + tuple(slice(c, c + r) for c, r in zip(ctx, roi))]


def _lazy_tta_views(cfg) -> int:
    """How many test-time-augmentation views the configuration asks for per window (1 = none)."""
    tta = getattr(getattr(cfg, "inference", None), "test_time_augmentation", None)
    if tta is None or not getattr(tta, "enabled", True):
        return 1
    from .tta_combinations import resolve_tta_augmentation_combinations
    return len(resolve_tta_augmentation_combinations(tta, spatial_dims=3))


def _open_mask(cfg, mask):
    """mask volume of a lazy run: a path (-> accessor under the config's mask transforms), an accessor, or an array-like
    (Z,Y,X) / (C,Z,Y,X) volume.  -> (accessor | None, array | None, owned accessor | None)"""
    if mask is None:
        return None, None, None
    if isinstance(mask, LazyVolumeAccessor):
        return mask, None, None
    if isinstance(mask, (str, bytes)) or hasattr(mask, "__fspath__"):
        acc = build_accessor(_DefaultDataCfg(cfg), str(mask if not isinstance(mask, bytes) else mask.decode()), kind="mask", mode="test")
        return acc, None, acc
    return None, _as_channel_first(mask), None


_PINNED_SLABS: dict = {}
_STAGE_MIN_BYTES = 32 << 20
_STAGE_SLAB_BYTES = 16 << 20
_STAGE_THREADS = 4
_STAGE_POOL = None
_STAGE_STREAMS: dict = {}


def _stage_host_box(box: np.ndarray, dev) -> Optional[torch.Tensor]:
    """fp32 device copy of a host array view (C, z, y, x) -- the bounding box of a region's windows cut out of an in-memory volume --
    moved in z-slabs through pinned host buffers by a few host threads: each gathers its slab (strided read + dtype conversion, numpy
    releases the GIL) and issues the non-blocking H2D copy on a copy stream, instead of one single-threaded contiguous host copy of the
    whole box followed by one pageable H2D copy (a 320 x 320 x 400 fp32 box: ~40 ms + ~20 ms of a 0.8 s MedNeXt-L chunk; the gather
    alone bounds a one-thread pipeline at ~40 ms).  Same values as np.ascontiguousarray(box, float32).
    None: not worth it (small box, not a CUDA device, stream capture, PYTC_LAZY_STAGE_PINNED=0) -- the caller takes the plain path."""
    import os
    if dev.type != "cuda" or box.ndim != 4 or box.nbytes < _STAGE_MIN_BYTES or torch.cuda.is_current_stream_capturing():
        return None
    if os.environ.get("PYTC_LAZY_STAGE_PINNED", "1") == "0":
        return None
    global _STAGE_POOL
    C_, Z, Y, X = (int(v) for v in box.shape)
    plane = C_ * Y * X * 4
    zs = max(1, min(Z, _STAGE_SLAB_BYTES // max(plane, 1)))
    nbuf = 2 * _STAGE_THREADS
    key = (str(dev), C_, zs, Y, X)
    bufs = _PINNED_SLABS.get(key)
    if bufs is None:
        if len(_PINNED_SLABS) >= 4:                          # a handful of box shapes per job; do not hoard pinned memory
            _PINNED_SLABS.clear()
        bufs = [[torch.empty((C_, zs, Y, X), dtype=torch.float32).pin_memory(), None] for _ in range(nbuf)]
        _PINNED_SLABS[key] = bufs
    if _STAGE_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _STAGE_POOL = ThreadPoolExecutor(max_workers=_STAGE_THREADS, thread_name_prefix="pytc-stage")
    copy_stream = _STAGE_STREAMS.get(str(dev))
    if copy_stream is None:
        copy_stream = _STAGE_STREAMS[str(dev)] = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream(dev)
    out = torch.empty((C_, Z, Y, X), dtype=torch.float32, device=dev)
    copy_stream.wait_stream(main)                            # `out` may reuse memory whose last use is still queued on the caller's stream
    slabs = [(z0, min(Z, z0 + zs)) for z0 in range(0, Z, zs)]

    def work(lane: int):
        with torch.cuda.device(dev):
            for k in range(lane, len(slabs), _STAGE_THREADS):
                z0, z1 = slabs[k]
                buf = bufs[(k // _STAGE_THREADS & 1) * _STAGE_THREADS + lane]      # two buffers per thread
                if buf[1] is not None:
                    buf[1].synchronize()                     # the copy out of this buffer, two of this thread's slabs ago
                np.copyto(buf[0].numpy()[:, :z1 - z0], box[:, z0:z1], casting="unsafe")
                with torch.cuda.stream(copy_stream):
                    for c in range(C_):                      # per channel: source and destination are both dense (one DMA each)
                        out[c, z0:z1].copy_(buf[0][c, :z1 - z0], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(copy_stream)
                buf[1] = ev

    for f in [_STAGE_POOL.submit(work, lane) for lane in range(min(_STAGE_THREADS, len(slabs)))]:
        f.result()
    main.wait_stream(copy_stream)
    return out


def _window_lanes(dev, n_chunks: int, *, pipelined_ok: bool):
    """Side streams for the window batches of the lazy loop ([] = the caller's stream only).  PYTC_LAZY_SW_STREAMS (default 4; 1 = off;
    measured on the MedNeXt-L 160^3 chunked leg, steady state: 1: 6.4e8, 2: 7.3e8, 3: 6.9e8, 4: 7.6e8, 6: 7.5e8 window-voxels/s).
    Only the channels-last fast path of this package's models is pipelined by default (stream-safe by construction, see
    EagerSlidingWindowEngine._lanes); setting the variable explicitly also covers the per-window TTA predictor path."""
    import os
    asked = os.environ.get("PYTC_LAZY_SW_STREAMS")
    n = min(int(asked) if asked else 4, int(n_chunks))
    if n < 2 or dev.type != "cuda" or ops.PROFILER.enabled or torch.cuda.is_current_stream_capturing():
        return []
    if not pipelined_ok and not asked:
        return []
    return pipeline_streams_for(dev, n)


@torch.no_grad()
def _lazy_sliding_window(cfg, forward_fn, volume, *, region_start, region_stop, device, requested_head=None,
                         window_filter: Optional[Callable[[int, int], bool]] = None,
                         return_accumulators: bool = False, preloaded=None, mask=None, mask_align_to_image: bool = False):
    roi = resolve_inferer_roi_size(cfg)
    if roi is None:
        raise ValueError("Lazy sliding-window inference requires inference.sliding_window.window_size "
                         "or model.output_size to be configured.")
    if len(roi) != 3:
        raise ValueError(f"Lazy sliding-window inference currently supports 3D only, got {roi}.")
    dev = torch.device(device)
    ops.require_device(dev, "lazy sliding-window inference")
    overlap = _coerce_overlap(resolve_inferer_overlap(cfg, roi), 3)
    sw = getattr(getattr(cfg, "inference", None), "sliding_window", None)
    dl = getattr(getattr(cfg, "data", None), "dataloader", None)
    swb = max(1, int(getattr(sw, "sw_batch_size", None) or getattr(dl, "batch_size", 1)))
    blend = str(getattr(sw, "blending", "bump")).strip().lower()
    pad_mode = getattr(sw, "padding_mode", "constant")
    cval = float(getattr(sw, "cval", 0.0))
    snap = bool(getattr(sw, "snap_to_edge", False))
    ctx = _resolve_target_context(sw, roi)
    border = resolve_border_mask(cfg, 3) or None
    primary = getattr(getattr(cfg, "model", None), "primary_head", None)

    owned = None
    if isinstance(volume, (str, bytes)) or hasattr(volume, "__fspath__"):
        volume = owned = open_lazy_source(cfg, volume)
    accessor = volume if isinstance(volume, LazyVolumeAccessor) else None
    vol = None if accessor is not None else _as_channel_first(volume)
    bounds = tuple(int(v) for v in (accessor.padded_spatial_shape if accessor is not None else vol.shape[1:]))
    if any(bounds[a] < int(roi[a]) for a in range(3)):
        raise ValueError("Lazy sliding-window inference requires the transformed test volume to be at least as "
                         f"large as the ROI in every axis. Got bounds_shape={bounds}, roi_size={tuple(roi)}.")
    start = (0, 0, 0) if region_start is None else tuple(max(0, int(v)) for v in region_start)
    stop = bounds if region_stop is None else tuple(min(bounds[a], int(region_stop[a])) for a in range(3))
    if any(stop[a] <= start[a] for a in range(3)):
        raise ValueError(f"Empty lazy inference region: start={start}, stop={stop}")
    out_size = tuple(stop[a] - start[a] for a in range(3))

    wins = [tuple(int(s.start) for s in sl) for sl in _build_intersecting_window_slices(
        bounds, roi, overlap, region_start=start, region_stop=stop, snap_to_edge=snap)]
    if not wins:
        raise RuntimeError("No lazy sliding-window patches were generated for this region.")
    if window_filter is not None:
        total = len(wins)
        wins = [w for i, w in enumerate(wins) if window_filter(i, total)]
        # an empty shard on ANY rank fails on EVERY rank before anything is read or reduced (lazy_distributed.py:110-129)
        validate_distributed_patch_shard(local_count=len(wins), total_count=total, device=dev)

    # bring the bounding box of everything the windows read (incl. context) into HBM once
    read = tuple(int(roi[a]) + 2 * ctx[a] for a in range(3))
    lo = tuple(max(0, min(w[a] for w in wins) - ctx[a]) for a in range(3))
    hi = tuple(min(bounds[a], max(w[a] for w in wins) + int(roi[a]) + ctx[a]) for a in range(3))
    if preloaded is not None and tuple(preloaded[0]) == (lo, hi):
        sub = preloaded[1]                                   # staged ahead by the chunked runner (raw bytes in pinned host memory)
    elif accessor is not None:
        sub = accessor.stage_region(lo, hi)
    else:
        sub = vol[:, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
        if isinstance(sub, np.ndarray):
            staged = _stage_host_box(sub, dev)               # z-slabs through two pinned buffers: host gather and H2D copy overlap
            sub = staged if staged is not None else torch.from_numpy(np.ascontiguousarray(sub, dtype=np.float32))
    if hasattr(sub, "to_device"):                            # StagedRegion: one H2D copy of the stored bytes + one resample kernel
        sub = sub.to_device(dev)
    else:
        sub = sub.to(device=dev, dtype=torch.float32, non_blocking=True).contiguous()

    # per-window test-time augmentation and / or a mask volume: every window batch goes through the predictor the way the
    # reference's loop does (lazy.py:1193-1198); without either, the channels-last fast path below
    mask_acc, mask_vol, mask_owned = _open_mask(cfg, mask)
    predictor = None
    if _lazy_tta_views(cfg) > 1 or mask_acc is not None or mask_vol is not None:
        from .tta import TTAPredictor
        predictor = TTAPredictor(cfg=cfg, sliding_inferer=None, forward_fn=forward_fn)
    mask_sub = None
    if mask_acc is not None:
        if tuple(mask_acc.padded_spatial_shape) != bounds:
            raise ValueError(f"mask volume shape {tuple(mask_acc.padded_spatial_shape)} does not match the image volume {bounds}")
        mask_sub = mask_acc.stage_region(lo, hi).to_device(dev)
    elif mask_vol is not None:
        if tuple(int(v) for v in mask_vol.shape[1:]) != bounds:
            raise ValueError(f"mask volume shape {tuple(mask_vol.shape[1:])} does not match the image volume {bounds}")
        mask_sub = mask_vol[:, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
        if isinstance(mask_sub, np.ndarray):
            mask_sub = torch.from_numpy(np.ascontiguousarray(mask_sub, dtype=np.float32))
        mask_sub = mask_sub.to(device=dev, dtype=torch.float32).contiguous()

    ks, combine = _axis_kernels(roi, blend, torch.float32)
    wz, wy, wx = (k.to(dev).contiguous() for k in ks)
    value = None
    weight = torch.zeros(out_size, dtype=torch.float32, device=dev)
    fwd_cl = getattr(getattr(forward_fn, "__self__", None), "forward_cl", None) if requested_head is None else None

    def predict_batch(chunk):
        # windows overhang the box only where the box touches the volume border; the kernel's periodic
        # reflect / replicate / circular index math equals the np.pad semantics of the reference reader
        rel = [tuple(w[a] - ctx[a] - lo[a] for a in range(3)) for w in chunk]
        x = ops.gather_windows(sub, rel, read, pad_mode=pad_mode, cval=cval)
        if accessor is not None:
            # the reader's per-window tail (reference read_patch: binarise, percentile clip, normalisation of THAT window,
            # outer padding included) on the gathered batch
            x = accessor.finish_windows(x)
        if predictor is not None:
            m = None
            if mask_sub is not None:                         # the mask window of the same box, zero outside the volume
                m = ops.gather_windows(mask_sub, rel, read, pad_mode="constant", cval=0.0)
                if mask_acc is not None:
                    m = mask_acc.finish_windows(m)
                m = m.permute(0, 4, 1, 2, 3).contiguous()
            out = predictor.predict_windows(x.permute(0, 4, 1, 2, 3).contiguous(), mask=m, mask_align_to_image=mask_align_to_image,
                                            requested_head=requested_head)
            _require_window_shape(out, read, roi, ctx)
            pred = out.permute(0, 2, 3, 4, 1).contiguous()
        elif fwd_cl is not None:
            pred = fwd_cl(x)
            _require_window_shape(pred, read, roi, ctx, channels_last=True)
        else:
            xin = x.permute(0, 4, 1, 2, 3)
            out = forward_fn(xin if x.shape[-1] == 1 else xin.contiguous())
            out, _ = select_output_tensor(out, requested_head=requested_head, primary_head=primary,
                                          purpose="inference output selection")
            _require_window_shape(out, read, roi, ctx)
            pred = out.permute(0, 2, 3, 4, 1).contiguous()
        if any(ctx):
            pred = pred[:, ctx[0]:ctx[0] + roi[0], ctx[1]:ctx[1] + roi[1], ctx[2]:ctx[2] + roi[2]].contiguous()
        # activations / channel selection: the predictor applied them per view (before the ensemble, as the reference does)
        pred = (pred.float() if pred.dtype != torch.float32 else pred).contiguous() if predictor is not None else \
            _window_preprocess(cfg, pred.contiguous())
        return pred

    def blend_batch(pred, chunk):
        nonlocal value
        if value is None:
            value = torch.zeros((int(pred.shape[-1]),) + out_size, dtype=torch.float32, device=dev)
        rel_starts = [tuple(w[a] - start[a] for a in range(3)) for w in chunk]
        ops.blend_accumulate(pred, rel_starts, value, weight, wz, wy, wx, combine=combine, floor_w=1e-5, border=border)

    chunks = [wins[b0:b0 + swb] for b0 in range(0, len(wins), swb)]
    # the first batch on the caller's stream (it allocates the accumulators there); the rest travel on the window pipeline's side
    # streams like the eager engine's batches (window.py: EagerSlidingWindowEngine.accumulate) -- the latency-bound deep levels of
    # one batch run under the HBM-bound full-resolution launches of its neighbours; the accumulators still see the batches in
    # window order (blend k waits for the event recorded after blend k-1), so the result is bit-identical to the one-stream loop
    blend_batch(predict_batch(chunks[0]), chunks[0])
    lanes = _window_lanes(dev, len(chunks) - 1, pipelined_ok=(fwd_cl is not None and predictor is None))
    if not lanes:
        for chunk in chunks[1:]:
            blend_batch(predict_batch(chunk), chunk)
    else:
        main = torch.cuda.current_stream(dev)
        for s_ in lanes:
            s_.wait_stream(main)
        order_ev = None
        try:
            for i, chunk in enumerate(chunks[1:]):
                s_ = lanes[i % len(lanes)]
                with torch.cuda.stream(s_):
                    pred = predict_batch(chunk)
                    if order_ev is not None:
                        s_.wait_event(order_ev)
                    blend_batch(pred, chunk)
                    order_ev = torch.cuda.Event()
                    order_ev.record(s_)
                del pred
        finally:
            for s_ in lanes:          # always join: the side streams may still be writing the accumulators
                main.wait_stream(s_)
    if owned is not None:
        owned.close()
    if mask_owned is not None:
        mask_owned.close()
    if return_accumulators:
        return value, weight
    ops.blend_finalize(value, weight, clamp=1e-4, act=nat.ACT_NONE)
    out = value.unsqueeze(0)
    odt = resolve_model_output_dtype(cfg)
    return out if odt == torch.float32 else out.to(odt)


def lazy_region_read_box(cfg, bounds, region_start, region_stop):
    """(lo, hi) of the box `_lazy_sliding_window` reads for a region: the bounding box of the global-grid windows that
    intersect it (plus target_context), clipped to the volume.  The chunked runner prefetches exactly this box."""
    roi = resolve_inferer_roi_size(cfg)
    overlap = _coerce_overlap(resolve_inferer_overlap(cfg, roi), 3)
    sw = getattr(getattr(cfg, "inference", None), "sliding_window", None)
    ctx = _resolve_target_context(sw, roi)
    start = tuple(max(0, int(v)) for v in region_start)
    stop = tuple(min(int(bounds[a]), int(region_stop[a])) for a in range(3))
    wins = [tuple(int(s.start) for s in sl) for sl in _build_intersecting_window_slices(
        bounds, roi, overlap, region_start=start, region_stop=stop, snap_to_edge=bool(getattr(sw, "snap_to_edge", False)))]
    lo = tuple(max(0, min(w[a] for w in wins) - ctx[a]) for a in range(3))
    hi = tuple(min(int(bounds[a]), max(w[a] for w in wins) + int(roi[a]) + ctx[a]) for a in range(3))
    return lo, hi


def _source_argument(image_path, volume):
    """The reference names the third argument `image_path`; this engine also takes an accessor or an array there.  `volume=`
    (this package's earlier keyword) keeps working."""
    if (image_path is None) == (volume is None):
        raise TypeError("pass the test volume once: as `image_path` (path, accessor or array), or as `volume`")
    return image_path if image_path is not None else volume


def lazy_predict_region(cfg, forward_fn, image_path=None, *, region_start: Sequence[int], region_stop: Sequence[int],
                        mask_path=None, mask_align_to_image: bool = False, device="cuda", requested_head: Optional[str] = None,
                        preloaded=None, volume=None) -> torch.Tensor:
    """Predict one bounded ZYX region (transformed / padded coordinates) of the test volume; windows come from the full-volume
    grid, so region borders see real neighbouring data wherever they are not true volume borders.  `mask_path`: a mask volume
    of the same (transformed) shape whose windows multiply the per-window prediction after test-time augmentation, as in the
    reference (lazy.py:1261-1292).  Returns (1, C, *region) on the device."""
    return _lazy_sliding_window(cfg, forward_fn, _source_argument(image_path, volume), region_start=region_start,
                                region_stop=region_stop, device=device, requested_head=requested_head, preloaded=preloaded,
                                mask=mask_path, mask_align_to_image=mask_align_to_image)


def lazy_predict_volume(cfg, forward_fn, image_path=None, *, mask_path=None, mask_align_to_image: bool = False, device="cuda",
                        requested_head: Optional[str] = None, volume=None) -> torch.Tensor:
    """Whole-volume variant.  With inference.sliding_window.distributed_sharding, a lazy data path and an initialised
    process group (lazy_distributed.is_distributed_window_sharding_enabled), the windows are sharded [rank::world]
    (reference lazy.py:1104-1110), every rank's shard is validated non-empty, and the HBM-resident value / weight
    accumulators are summed onto rank 0 in place, `distributed_reduce_chunk_mb` per collective
    (lazy_distributed.py:78-169); rank 0 normalises, the other ranks get an empty tensor.  TTA-view sharding cannot be
    combined with it (lazy.py:1039-1043)."""
    volume = _source_argument(image_path, volume)
    masked = dict(mask=mask_path, mask_align_to_image=mask_align_to_image)
    if not is_distributed_window_sharding_enabled(cfg):
        return _lazy_sliding_window(cfg, forward_fn, volume, region_start=None, region_stop=None, device=device,
                                    requested_head=requested_head, **masked)
    tta = getattr(getattr(cfg, "inference", None), "test_time_augmentation", None)
    if tta is not None and getattr(tta, "enabled", False) and getattr(tta, "distributed_sharding", False):
        raise RuntimeError("Lazy sliding-window inference does not support "
                           "`inference.test_time_augmentation.distributed_sharding`. Disable it first.")
    sw = cfg.inference.sliding_window
    _is_dist, rank, world = distributed_context()
    value, weight = _lazy_sliding_window(cfg, forward_fn, volume, region_start=None, region_stop=None, device=device,
                                         requested_head=requested_head, return_accumulators=True,
                                         window_filter=lambda i, n: i % world == rank, **masked)
    hook = make_accumulator_reduce_hook(chunk_mb=int(getattr(sw, "distributed_reduce_chunk_mb", 128) or 128))
    reduced = hook(value, weight)
    if reduced is None:
        return torch.empty(0, device=value.device)
    value, weight = reduced
    ops.blend_finalize(value, weight, clamp=1e-4, act=nat.ACT_NONE)
    out = value.unsqueeze(0)
    odt = resolve_model_output_dtype(cfg)
    return out if odt == torch.float32 else out.to(odt)


__all__ = ["lazy_predict_region", "lazy_predict_volume", "load_lazy_volume", "LazyVolumeAccessor", "get_lazy_image_reference_shape", "open_lazy_source", "lazy_region_read_box",
           "_build_window_axis_offsets", "_build_window_slices", "_build_intersecting_window_slices",
           "_snap_offsets", "_resolve_target_context"]
