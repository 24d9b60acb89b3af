"""Whole-volume intensity normalisation of an image loaded for the eager test / training path of `main.py`.

The reference's data pipeline applies `smart_normalize` (connectomics/data/augmentation/augment_ops.py:552-611) to every loaded image under
`data.image_transform` -- and its schema default is mode "0-1" (config/schema/data.py:129), which tutorials such as mito_lucchi++ rely on
without naming it.  The lazy / chunked path does this per window on the device (`LazyVolumeAccessor.finish_windows`); this is the same
arithmetic for a whole host volume, in numpy, before the one upload: percentile clip first, then `none` | `normal` (z-score, skipped for
a flat volume) | `0-1` (min-max, skipped for a flat volume) | `divide` / `divide-K`.  Pinned by tests/golden/smart_normalize.npz."""
from __future__ import annotations

from typing import Any, Optional

import numpy as np

_MODES = "'none', 'normal', '0-1', 'divide', or 'divide-K'"


def _divisor(mode: str, divide_value: Optional[float]) -> tuple[str, Optional[float]]:
    """('divide-255', _) -> ('divide', 255.0); every other mode unchanged."""
    head, dash, tail = mode.partition("-")
    if head != "divide" or not dash:
        return mode, divide_value
    try:
        return "divide", float(tail)
    except ValueError as exc:
        raise ValueError(f"Invalid divide mode '{mode}'. Format should be 'divide-K' where K is a number (e.g., 'divide-255').") from exc


def normalize_volume(volume: np.ndarray, mode: str = "0-1", *, divide_value: Optional[float] = None, clip_percentile_low: float = 0.0,
                     clip_percentile_high: float = 1.0) -> np.ndarray:
    """A normalised COPY of `volume` (numpy's own type promotion, like the reference: integer volumes come back as float64)."""
    mode, divide_value = _divisor(str(mode), divide_value)
    out = np.array(volume, copy=True)
    if clip_percentile_low > 0.0 or clip_percentile_high < 1.0:
        lo, hi = (np.percentile(out, 100.0 * q) for q in (clip_percentile_low, clip_percentile_high))
        out = np.clip(out, lo, hi)
    if mode == "none":
        return out
    if mode == "normal":
        centre, spread = out.mean(), out.std()
        return (out - centre) / spread if spread > 1e-8 else out
    if mode == "0-1":
        lo, hi = out.min(), out.max()
        return (out - lo) / (hi - lo) if hi > lo else out
    if mode == "divide":
        if divide_value is None or float(divide_value) == 0.0:
            raise ValueError("smart_normalize mode='divide' requires a non-zero divide_value (or use 'divide-K' form to embed the divisor "
                             "in the mode string).")
        return out / float(divide_value)
    raise ValueError(f"Unknown smart_normalize mode '{mode}'. Expected {_MODES}.")


def normalize_image_for_config(volume: np.ndarray, cfg: Any) -> np.ndarray:
    """`normalize_volume` under `cfg.data.image_transform` (normalize / clip_percentile_low / clip_percentile_high); a configuration
    without that section means mode "none" (array-driven callers and tests that hand over prepared data)."""
    section = getattr(getattr(cfg, "data", None), "image_transform", None)
    mode = getattr(section, "normalize", None) if section is not None else None
    if mode is None:
        return volume
    low = float(getattr(section, "clip_percentile_low", 0.0) or 0.0)
    high = float(getattr(section, "clip_percentile_high", 1.0) or 1.0)
    if str(mode) == "none" and low <= 0.0 and high >= 1.0:
        return volume
    return normalize_volume(volume, str(mode), clip_percentile_low=low, clip_percentile_high=high)


def _spatial_pad_widths(pad_size, ndim: int):
    """`data.data_transform.pad_size` (1, 3 or 6 ints, ZYX) -> np.pad widths for a (Z, Y, X) or (C, Z, Y, X) volume."""
    values = [int(v) for v in (pad_size or [])]
    if not values or not any(values):
        return None
    if len(values) == 1:
        values = values * 3
    if len(values) == 3:
        pairs = [(v, v) for v in values]
    elif len(values) == 6:
        pairs = [(values[2 * i], values[2 * i + 1]) for i in range(3)]
    else:
        raise ValueError(f"pad_size length must be 1, 3, or 6, got {len(values)}")
    return [(0, 0)] * (ndim - 3) + pairs


def prepare_test_image(volume: np.ndarray, cfg: Any) -> np.ndarray:
    """What the reference's TEST transforms do to a loaded image before sliding-window inference, in their order
    (data/augmentation/build.py:416-655): `data_transform.val_transpose` on the spatial axes, the explicit context border
    `data_transform.pad_size` in `pad_mode` (reflect by default -- the border that `inference.crop_pad` removes again from the
    prediction, e.g. tutorials/neuron_snemi), then the intensity normalisation over the padded volume.  `data_transform.resize` is
    refused by name: resampling lives in the lazy reader of this package (`data.dataloader.use_lazy_*`)."""
    shared = getattr(getattr(cfg, "data", None), "data_transform", None)
    out = volume
    if shared is not None:
        if getattr(shared, "resize", None):
            raise NotImplementedError("data.data_transform.resize is honoured by the lazy reader only (set data.dataloader.use_lazy_h5 / "
                                      "use_lazy_zarr with inference.chunking), not by the eager test path of main.py")
        axes = [int(a) for a in (getattr(shared, "val_transpose", None) or [])]
        if axes:
            if sorted(axes) != [0, 1, 2]:
                raise ValueError(f"data.data_transform.val_transpose must be a permutation of (0, 1, 2), got {axes}")
            lead = out.ndim - 3
            out = np.transpose(out, list(range(lead)) + [lead + a for a in axes])
        widths = _spatial_pad_widths(getattr(shared, "pad_size", None), out.ndim)
        if widths is not None:
            mode = str(getattr(shared, "pad_mode", "reflect") or "reflect")
            mode = {"replicate": "edge", "circular": "wrap"}.get(mode, mode)
            out = np.pad(out, widths, mode="constant", constant_values=0) if mode == "constant" else np.pad(out, widths, mode=mode)
    return normalize_image_for_config(out, cfg)


def _mask_section(cfg: Any):
    data = getattr(cfg, "data", None)
    return getattr(data, "mask_transform", None) or getattr(data, "data_transform", None)


def mask_align_to_image(cfg: Any) -> bool:
    """`align_to_image` of `data.mask_transform` (else `data.data_transform`): allow the small centre pad / crop of a mask onto the
    prediction (reference training/lightning/test_pipeline.py:282-288)."""
    return bool(getattr(_mask_section(cfg), "align_to_image", False))


def prepare_test_mask(mask: np.ndarray, cfg: Any) -> np.ndarray:
    """The test transforms of a MASK volume (reference data/augmentation/build.py:566-615): the image's val_transpose, strict
    binarisation `mask > threshold` when `mask_transform.binarize` (dtype kept), and the context border filled with zeros -- outside the
    source field of view nothing is kept."""
    shared = getattr(getattr(cfg, "data", None), "data_transform", None)
    section = _mask_section(cfg)
    out = mask
    axes = [int(a) for a in (getattr(shared, "val_transpose", None) or [])]
    if axes:
        lead = out.ndim - 3
        out = np.transpose(out, list(range(lead)) + [lead + a for a in axes])
    if bool(getattr(section, "binarize", False)):
        out = (out > float(getattr(section, "threshold", 0.0))).astype(out.dtype, copy=False)
    widths = _spatial_pad_widths(getattr(shared, "pad_size", None), out.ndim)
    return out if widths is None else np.pad(out, widths, mode="constant", constant_values=0)


__all__ = ["normalize_volume", "normalize_image_for_config", "prepare_test_image", "prepare_test_mask", "mask_align_to_image"]
