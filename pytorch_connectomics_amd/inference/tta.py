"""Test-time augmentation + activation + channel selection on the device -- counterpart of the reference's
connectomics/inference/tta.py (TTAPredictor :67, apply_preprocessing :312-402, _run_ensemble :691-769,
_predict_prepared_tensor :806-878, _predict_patch_first_local :880-1314, predict :1619-1666).

MI355X design: a TTA view is index math inside the gather / blend kernels (no flipped copies of the volume
or of the predictions); each view gets one overlap-add pass over the SAME window grid into an HBM-resident
accumulator (the reference's patch-first-local semantics), then normalise -> per-channel activation ->
channel selection -> streaming mean/min/max ensemble, all as device kernels.

Quarter turns in every plane (round 6: the planes that contain z exchange z with y / x inside the gather and blend kernels,
PYTC_VIEW_SWAP_ZY / _ZX); an odd turn needs a window -- and, patch-first, an image -- of equal size along the plane's axes, as in the
reference (tta.py:1316-1340).
"""
from __future__ import annotations

import logging
from typing import Any, Optional

import torch

from .. import _native as nat
from .. import hip_ops as ops
from ..utils.channel_slices import resolve_channel_indices
from ..utils.model_outputs import (get_inference_channel_activations, get_inference_select_channel,
                                   select_output_tensor)
from .lazy_distributed import reduce_view_ensemble, validate_view_shards
from .tta_combinations import (_resolve_ensemble_mode_map, _resolve_spatial_dims, apply_view,
                               resolve_tta_augmentation_combinations)
from .window import is_2d_inference_mode, resolve_inferer_roi_size, resolve_model_output_dtype
from ..utils.model_outputs import resolve_output_channels, resolve_output_heads

logger = logging.getLogger(__name__)
tqdm = None          # the device engine reports no per-batch progress; the name exists because callers of the reference module patch it

_MODE_CODE = {"mean": 0, "min": 1, "max": 2}


def view_code(flip_axes, rotation_plane, k: int) -> int:
    """Map a reference view (flips, then rot90^k in `rotation_plane`; spatial axes 0=z,1=y,2=x) onto the
    kernels' PYTC_VIEW_* encoding  out[z,y,x] = win[T(F(z,y,x))]  (F: per-axis flips, T: exchange of the two axes of the plane)
    by matching its action on a cubic probe.  Any plane (tta_combinations.py:90-119): an odd turn exchanges the plane's two window
    axes, which must then have equal length -- checked where the window size is known (EagerSlidingWindowEngine.accumulate, the
    kernels)."""
    probe = torch.arange(3 * 3 * 3).reshape(3, 3, 3)
    want = apply_view(probe, list(flip_axes or []), rotation_plane, int(k), first_spatial_dim=0)
    swaps = [0] + sorted(nat.VIEW_SWAPS)
    for swap in swaps:
        t = probe.transpose(*nat.VIEW_SWAPS[swap]) if swap else probe
        for flips in range(8):
            dims = [d for d, bit in enumerate((nat.VIEW_FLIP_Z, nat.VIEW_FLIP_Y, nat.VIEW_FLIP_X)) if flips & bit]
            mine = torch.flip(t, dims) if dims else t
            if torch.equal(mine, want):
                return swap | flips
    raise NotImplementedError(f"TTA view (flip={flip_axes}, plane={rotation_plane}, k={k}) is not expressible as "
                              "flips + one exchange of two axes")


class TTAPredictor:
    """``predict(images)`` -> ensembled, activated, channel-selected prediction (1, C_sel, Z, Y, X)."""

    def __init__(self, cfg, sliding_inferer, forward_fn, model=None):
        self.cfg = cfg
        self.sliding_inferer = sliding_inferer
        self.forward_fn = forward_fn
        self.model = model
        self.channel_activation_types = None
        self._requested_output_head_override: Optional[str] = None
        self._last_distributed_sharding_active = False
        self._last_skip_postprocess_on_rank = False
        self._parse_channel_activations()

    def _parse_channel_activations(self) -> None:
        """Activation name per output channel from the configuration alone (the channel count the config declares for the
        selected output, else model.out_channels) -- so a bad `channel_activations` entry is refused when the predictor is built,
        not in the middle of a volume (reference tta.py:202-230).  `apply_preprocessing` refreshes it from the real tensor."""
        if not hasattr(self.cfg, "inference") or not get_inference_channel_activations(self.cfg):
            return
        hint = resolve_output_channels(self.cfg, requested_head=self._requested_output_head_override,
                                       purpose="TTA channel activation parsing", allow_ambiguous=True)
        if hint is None:
            hint = int(getattr(getattr(self.cfg, "model", None), "out_channels", 0) or 0)
        if hint <= 0:
            self.channel_activation_types = None
            return
        names: list = [None] * hint
        for idx, act in self._resolve_channel_activation_specs(hint):
            for c in idx:
                names[c] = act
        self.channel_activation_types = names if any(n is not None for n in names) else None

    # ------------------------------------------------------------------ config helpers
    @staticmethod
    def _distributed_context():
        """(initialised?, rank, world size) -- lazy_distributed.distributed_context under the reference predictor's name."""
        from .lazy_distributed import distributed_context
        return distributed_context()

    def _build_augmentation_combinations(self, tta_cfg, ndim: int):
        """The (flip axes, rotation plane, k) views of a configuration for tensors of rank `ndim`, in the reference predictor's
        convention (tta.py:603-624): flip axes as configured (0 = z, 1 = y, 2 = x), rotation planes in TENSOR dims (+2 for batch and
        channel).  The one place the predictor enumerates views, so it is also the seam for a caller that wants its own list."""
        return [(flips, None if plane is None else tuple(int(a) + 2 for a in plane), k)
                for flips, plane, k in resolve_tta_augmentation_combinations(tta_cfg, spatial_dims=_resolve_spatial_dims(ndim))]

    @staticmethod
    def _spatial_augmentation_combinations(augmentation_combinations):
        """The same views with rotation planes as SPATIAL axes -- what the engine and the affinity plans index (tta.py:626-636)."""
        return [(flips, None if plane is None else (int(plane[0]) - 2, int(plane[1]) - 2), k)
                for flips, plane, k in augmentation_combinations]

    def _get_tta_cfg(self):
        return getattr(getattr(self.cfg, "inference", None), "test_time_augmentation", None)

    def is_distributed_sharding_enabled(self) -> bool:
        tta = self._get_tta_cfg()
        is_dist = torch.distributed.is_available() and torch.distributed.is_initialized()
        return bool(tta is not None and getattr(tta, "enabled", False) and getattr(tta, "distributed_sharding", False)
                    and is_dist and torch.distributed.get_world_size() > 1)

    def should_skip_postprocess_on_rank(self) -> bool:
        return bool(self._last_skip_postprocess_on_rank)

    def _select_channel_indices(self, num_channels: int):
        sel = get_inference_select_channel(self.cfg)
        if sel is None:
            return None
        return resolve_channel_indices(sel, num_channels=int(num_channels), context="inference.model.select_channel")

    def _merged_head_window(self):
        """(offset, width, merged width) of the head being predicted inside the MERGED output, or None.  With
        `inference.model.head: "aff,sdt"` every head is predicted on its own and the results are concatenated, but
        `channel_activations` is written against the concatenation ("0:6" aff, "6:7" sdt): the entries have to be resolved there
        and shifted into the head's own channel numbering (reference tta.py:94-139).  None for single-tensor models, for a
        request that is itself a list, and for a head that is not part of the configured merged output."""
        head = self._requested_output_head_override
        if not isinstance(head, str) or "," in head:
            return None
        merged = resolve_output_heads(self.cfg, purpose="channel activation scoping")
        if head not in merged:
            return None
        def width(names):
            return resolve_output_channels(self.cfg, requested_head=",".join(names), purpose="channel activation scoping",
                                           allow_ambiguous=False)
        total, own = width(merged), width([head])
        if total is None or own is None:
            return None
        before = merged[:merged.index(head)]
        return (int(width(before) or 0) if before else 0), int(own), int(total)

    def _resolve_channel_activation_specs(self, num_channels: int):
        """`inference.model.channel_activations` -> [(channel indices of THIS tensor, activation)]: every entry a mapping with
        `channels` and `activation`, no channel claimed twice; under merged-head inference the selectors are resolved in the merged
        numbering, entries of other heads dropped and the rest shifted to local indices (reference tta.py:141-200)."""
        window = self._merged_head_window()
        specs, taken = [], set()
        for pos, entry in enumerate(get_inference_channel_activations(self.cfg)):
            if not isinstance(entry, dict):
                raise ValueError("inference.model.channel_activations entries must be mappings with keys "
                                 f"'channels' and 'activation', got {type(entry).__name__}.")
            if "channels" not in entry or "activation" not in entry:
                raise ValueError(f"inference.model.channel_activations[{pos}] must define both 'channels' and 'activation'.")
            where = f"inference.model.channel_activations[{pos}].channels"
            if window is None:
                idx = resolve_channel_indices(entry["channels"], num_channels=num_channels, context=where)
            else:
                first, own, total = window
                idx = [c - first for c in resolve_channel_indices(entry["channels"], num_channels=total, context=where)
                       if first <= c < first + own]
                if not idx:
                    continue                              # an entry of another head of the merged output
            twice = sorted(taken.intersection(idx))
            if twice:
                raise ValueError(f"inference.model.channel_activations[{pos}] overlaps already assigned channels: "
                                 f"{', '.join(map(str, twice))}.")
            taken.update(idx)
            specs.append((idx, entry["activation"]))
        return specs

    _activation_specs = _resolve_channel_activation_specs

    # ------------------------------------------------------------------ network plumbing
    def _network_tensor(self, x: torch.Tensor) -> torch.Tensor:
        outputs = self.forward_fn(x)
        primary = getattr(getattr(self.cfg, "model", None), "primary_head", None)
        out, _ = select_output_tensor(outputs, requested_head=self._requested_output_head_override,
                                      primary_head=primary, purpose="inference output selection")
        return out

    def _engine_network(self):
        """Callable handed to the window engine: the channels-last fast path when the wrapped model offers
        one and no named head has to be selected, the generic NCDHW callable otherwise."""
        m = self.model
        if (m is not None and hasattr(m, "forward_cl") and self._requested_output_head_override is None
                and getattr(self.forward_fn, "__self__", m) is m):
            return m
        return self._network_tensor

    # ------------------------------------------------------------------ activation / selection (device)
    def apply_preprocessing(self, tensor: torch.Tensor) -> torch.Tensor:
        """volume (N, C, Z, Y, X) fp32 on the device (N = 1 for a whole volume, a window batch in the lazy path): per-channel
        activations in place, then channel selection and the output dtype cast (reference tta.py:312-402)."""
        volume = tensor
        if not hasattr(self.cfg, "inference"):
            return volume
        C = int(volume.shape[1])
        types: list[Optional[str]] = [None] * C
        for idx, act in self._activation_specs(C):
            for c in idx:
                types[c] = act
            contiguous = idx == list(range(idx[0], idx[-1] + 1))
            groups = [(idx[0], idx[-1] + 1)] if contiguous else [(c, c + 1) for c in idx]
            if act is None or (isinstance(act, str) and act.lower() == "none"):
                continue
            if act == "sigmoid":
                code, scale = nat.ACT_SIGMOID, 1.0
            elif isinstance(act, str) and (act == "scale_sigmoid" or act.startswith("scale_sigmoid:")):
                scale = 0.2
                if ":" in act:
                    try:
                        scale = float(act.split(":", 1)[1])
                    except ValueError as exc:
                        raise ValueError(f"Invalid scale_sigmoid scale in '{act}'. "
                                         "Expected 'scale_sigmoid:<float>'.") from exc
                code = nat.ACT_SIGMOID
            elif act == "tanh":
                code, scale = nat.ACT_TANH, 1.0
            elif act == "softmax":
                if len(idx) <= 1:
                    logger.warning(f"Softmax activation for single channel ({idx[0]}) is not meaningful. Skipping.")
                    continue
                if not contiguous:
                    raise NotImplementedError("softmax over a non-contiguous channel list is not supported on device")
                code, scale = nat.ACT_SOFTMAX, 1.0
            else:
                raise ValueError(f"Unknown activation '{act}' for channels {idx}. Supported: 'sigmoid', "
                                 "'scale_sigmoid' (or 'scale_sigmoid:<float>'), 'softmax', 'tanh', None")
            for sample in volume:                       # (C, Z, Y, X) slabs of the batch
                for a, b in groups:
                    ops.channel_activation(sample, a, b, code, scale)
        self.channel_activation_types = types if any(t is not None for t in types) else None
        sel = self._select_channel_indices(C)
        if sel is not None:
            if sel != list(range(C)):
                volume = volume[:, sel].contiguous()
            if self.channel_activation_types is not None:
                self.channel_activation_types = [self.channel_activation_types[i] for i in sel]
        out_dtype = resolve_model_output_dtype(self.cfg)
        if volume.dtype != out_dtype:
            volume = volume.to(out_dtype)
        return volume

    # ------------------------------------------------------------------ masks
    @classmethod
    def _coerce_mask_to_tensor(cls, mask) -> torch.Tensor:
        """What a dataloader's collation can hand over as "the mask" -> one tensor: singleton lists / tuples are unwrapped, arrays
        wrapped, longer lists stacked (reference tta.py:550-574)."""
        while isinstance(mask, (list, tuple)) and len(mask) == 1:
            mask = mask[0]
        if torch.is_tensor(mask):
            return mask
        try:
            import numpy as np
            if isinstance(mask, np.ndarray):
                return torch.from_numpy(mask)
        except ImportError:      # pragma: no cover
            pass
        if not isinstance(mask, (list, tuple)):
            raise TypeError(f"Unsupported mask type: {type(mask).__name__}")
        parts = [cls._coerce_mask_to_tensor(item) for item in mask]
        if not parts:
            raise ValueError("Mask list is empty after collation.")
        if len({tuple(t.shape) for t in parts}) > 1:
            raise ValueError(f"Mask list contains tensors with incompatible shapes for stacking: {[tuple(t.shape) for t in parts]}")
        return torch.stack(parts)

    def _validate_and_prepare_mask(self, mask, prediction: torch.Tensor, align_to_image: bool = False) -> torch.Tensor:
        """-> the BINARY mask (mask > 0) in the prediction's dtype, rank, batch and spatial shape (reference tta.py:465-548): a
        missing channel axis (or batch and channel axes) is added, a depth-1 mask meets a 2-D prediction, a single mask serves the
        whole batch, one mask channel serves all prediction channels; spatial shapes must agree unless `align_to_image` allows the
        centre crop / zero pad."""
        if mask is None:
            raise ValueError("Mask is None while mask application is enabled.")
        mask = self._coerce_mask_to_tensor(mask).to(device=prediction.device)
        missing = prediction.dim() - mask.dim()
        if missing == 1:
            mask = mask[:, None]
        elif missing == 2:
            mask = mask[None, None]
        elif missing == -1 and prediction.dim() == 4 and mask.shape[2] == 1:
            mask = mask[:, :, 0]
        if mask.dim() != prediction.dim():
            raise ValueError(f"Mask rank {mask.dim()} does not match prediction rank {prediction.dim()}. "
                             f"mask.shape={tuple(mask.shape)}, prediction.shape={tuple(prediction.shape)}")
        if mask.shape[0] != prediction.shape[0]:
            if mask.shape[0] != 1:
                raise ValueError(f"Mask batch {mask.shape[0]} does not match prediction batch {prediction.shape[0]}.")
            mask = mask.expand(prediction.shape[0], *mask.shape[1:])
        if mask.shape[1] not in (1, prediction.shape[1]):
            raise ValueError(f"Mask channels {mask.shape[1]} incompatible with prediction channels {prediction.shape[1]}. "
                             f"Expected C=1 or C={prediction.shape[1]}.")
        if mask.shape[2:] != prediction.shape[2:]:
            if not align_to_image:
                raise ValueError("Mask spatial shape must exactly match prediction spatial shape. "
                                 f"Got mask.shape={tuple(mask.shape)} and prediction.shape={tuple(prediction.shape)}. "
                                 "Fix test/tune mask preprocessing so they produce identical spatial dimensions.")
            for dim in range(2, prediction.dim()):
                extra = int(mask.shape[dim]) - int(prediction.shape[dim])
                if extra > 0:                                   # centre crop
                    mask = mask.narrow(dim, extra // 2, int(prediction.shape[dim]))
                elif extra < 0:                                 # centre zero pad
                    before = (-extra) // 2
                    widths = [0, 0] * (prediction.dim() - 1 - dim) + [before, -extra - before]
                    mask = torch.nn.functional.pad(mask, widths, mode="constant", value=0)
        return (mask > 0).to(prediction.dtype)

    def _apply_mask_to_result(self, result: torch.Tensor, mask, mask_align_to_image: bool) -> torch.Tensor:
        """result * mask, channel by channel; a `tanh` channel is filled with -1 (its background value) outside the mask instead
        of 0 (reference tta.py:1568-1617)."""
        tta = self._get_tta_cfg()
        if mask is None or not (getattr(tta, "apply_mask", True) if tta is not None else True):
            return result
        try:
            mask = self._validate_and_prepare_mask(mask, result, align_to_image=mask_align_to_image)
        except TypeError as exc:
            logger.warning("Skipping mask application because the provided mask payload is not a tensor-like volume: %s", exc)
            return result
        types = self.channel_activation_types
        if types is None or len(types) != result.shape[1]:
            return result * mask
        for c, name in enumerate(types):
            m = mask[:, c:c + 1] if mask.shape[1] == result.shape[1] else mask[:, :1]
            kept = m * result[:, c:c + 1]
            result[:, c:c + 1] = kept + (m - 1) if name == "tanh" else kept
        return result

    # ------------------------------------------------------------------ view sharding
    def _reduce_views(self, acc, n_local, total, modes, *, skip=(), stats=None, counts=None, partial_modes=()):
        """Per-rank ensembles -> rank 0 (tta.py:1341-1519), in place in HBM; None on the other ranks."""
        tta = self._get_tta_cfg()
        return_value = reduce_view_ensemble(acc.contiguous(), n_local, total, modes,
                                            chunk_mb=int(getattr(tta, "distributed_reduce_chunk_mb", 128) or 128),
                                            skip_channels=skip, stats=stats, counts=counts, partial_modes=partial_modes)
        if return_value is None:
            return None
        return return_value[0] if stats is None else return_value

    def _finish(self, result, mask, mask_align_to_image):
        """Mask + return on the rank that holds the ensemble; the contributing ranks of a sharded run return an empty
        tensor and skip post-processing (tta.py:868-873)."""
        if result is None:
            self._last_skip_postprocess_on_rank = True
            return torch.empty(0, device=self._last_device)
        return self._apply_mask_to_result(result, mask, mask_align_to_image)

    # ------------------------------------------------------------------ predict
    def _normalize_input(self, images: torch.Tensor) -> torch.Tensor:
        if images.ndim == 3:
            images = images.unsqueeze(0).unsqueeze(0)
        elif images.ndim == 4:
            images = images.unsqueeze(1)
        elif images.ndim != 5:
            raise ValueError(f"TTA requires 3D, 4D, or 5D input tensor. Got {images.ndim}D tensor with shape "
                             f"{images.shape}. Expected shapes: (D, H, W), (B, D, H, W), or (B, C, D, H, W)")
        # 2-D mode (data.*.do_2d with a depth-1 batch): the reference squeezes the depth axis here (tta.py:598-599) and runs
        # everything in 2-D; the device engine keeps the depth-1 volume (its kernels index three axes) and predict() drops
        # the axis from the result instead
        return images

    def _is_flat_2d(self, images: torch.Tensor) -> bool:
        return bool(is_2d_inference_mode(self.cfg) and images.dim() == 5 and images.size(2) == 1)

    def _foreign_inferer(self) -> bool:
        """A sliding inferer that is not this package's engine: a plain callable without the `accumulate` entry point."""
        return self.sliding_inferer is not None and not hasattr(self.sliding_inferer, "accumulate")

    def _engine_for(self, images: torch.Tensor):
        """The configured sliding engine (for a caller's own inferer: the engine the configuration describes), or a single-window
        engine covering the whole image."""
        if self.sliding_inferer is not None and not self._foreign_inferer():
            return self.sliding_inferer
        from .window import EagerSlidingWindowEngine, build_sliding_inferer
        if self._foreign_inferer():
            own = build_sliding_inferer(self.cfg)
            if own is not None:
                return own
        return EagerSlidingWindowEngine(roi_size=tuple(images.shape[2:]), sw_batch_size=1, overlap=0.0,
                                        mode="constant", padding_mode="constant", cval=0.0)

    @torch.no_grad()
    def predict(self, images: torch.Tensor, mask=None, mask_align_to_image: bool = False,
                requested_head: Optional[str] = None) -> torch.Tensor:
        images = self._normalize_input(images)
        if images.shape[0] > 1:
            # a batch of volumes: one after the other through the single-volume engine (the reference's patch-first loop also walks
            # the batch sample by sample, tta.py:942-944); a mask with a matching batch axis is split along with it
            whole = None
            if mask is not None:
                try:
                    whole = self._coerce_mask_to_tensor(mask)
                except TypeError:
                    whole = None
            def mask_of(b):
                if whole is None:
                    return mask
                batched = whole.dim() >= images.dim() - 1 and whole.shape[0] == images.shape[0]
                return whole[b:b + 1] if batched else whole
            parts = [self.predict(images[b:b + 1], mask=mask_of(b), mask_align_to_image=mask_align_to_image,
                                  requested_head=requested_head) for b in range(images.shape[0])]
            return torch.cat(parts, 0) if all(p.numel() for p in parts) else parts[0]
        flat2d = self._is_flat_2d(images)
        if flat2d and isinstance(mask, torch.Tensor) and mask.dim() == 4:
            mask = mask.unsqueeze(2)                        # (B, C, H, W) -> the depth-1 volume the result is masked as
        out = self._predict_volume(images, mask, mask_align_to_image, requested_head, flat2d)
        return out.squeeze(2) if (flat2d and out.dim() == 5) else out      # (B, C, H, W) like the reference's 2-D mode

    @torch.no_grad()
    def predict_windows(self, windows: torch.Tensor, mask=None, mask_align_to_image: bool = False,
                        requested_head: Optional[str] = None, run=None) -> torch.Tensor:
        """A BATCH of windows (B, C, *roi) through the network without a sliding engine -- what the reference's lazy loop asks
        its predictor for (lazy.py:1193-1198 -> tta.py:806-878 with `use_sliding=False`): every configured view of the batch is
        predicted, mapped back (`invert_view`: inverse quarter turns / flips, affinity channels re-anchored), activated and
        channel-selected, and streamed into a `TTAEnsembleAccumulator` (mean / min / max per channel, validity-aware); the
        ensemble is masked last.  -> (B, C_sel, *roi) in the configured output dtype, on the device.  `run` replaces the direct network
        call per view (a caller-supplied sliding inferer: `predict`)."""
        from .tta_affinity import ViewValidity, build_affinity_tta_plan, invert_view, resolve_affinity_channel_groups_from_cfg, \
            validate_affinity_output
        from .tta_ensemble import TTAEnsembleAccumulator
        prev, self._requested_output_head_override = self._requested_output_head_override, requested_head
        try:
            ops.require_device(windows.device, "TTAPredictor")
            x = windows if windows.dtype == torch.float32 else windows.float()
            tta = self._get_tta_cfg()
            combos = [([], None, 0)]
            if tta is not None and getattr(tta, "enabled", True):
                combos = self._spatial_augmentation_combinations(self._build_augmentation_combinations(tta, x.dim()))
            has_affinity = bool(resolve_affinity_channel_groups_from_cfg(self.cfg))
            plan = None
            acc = None
            for i, (flips, plane, k) in enumerate(combos):
                pred = (run or self._network_tensor)(apply_view(x, flips, plane, k, first_spatial_dim=2).contiguous())
                pred = pred if pred.dtype == torch.float32 else pred.float()
                if has_affinity and plan is None:
                    plan = build_affinity_tta_plan(self.cfg, augmentation_combinations=combos, num_raw=int(pred.shape[1]),
                                                   requested_head=requested_head)
                if len(combos) == 1:                      # no augmentation: nothing to invert or to ensemble
                    validate_affinity_output(plan, pred)
                    return self._apply_mask_to_result(self.apply_preprocessing(pred.contiguous()), mask, mask_align_to_image)
                pred, validity = invert_view(pred, flip_axes=flips, rotation_plane_spatial=plane, k=k,
                                             view_plan=None if plan is None else plan.views[i], tta_plan=plan)
                raw_channels = int(pred.shape[1])
                sel = self._select_channel_indices(raw_channels)
                done = self.apply_preprocessing(pred.contiguous())
                kept = list(range(raw_channels)) if sel is None else list(sel)
                if acc is None:
                    partial = [] if plan is None else [j for j, c in enumerate(kept) if c in plan.partial_channels]
                    acc = TTAEnsembleAccumulator(tuple(done.shape), dtype=resolve_model_output_dtype(self.cfg), device=done.device,
                                                 mode_map=_resolve_ensemble_mode_map(getattr(tta, "ensemble_mode", "mean"),
                                                                                     int(done.shape[1])),
                                                 partial_channels=partial, distributed_sharding=False, max_views=len(combos))
                acc.add(done, ViewValidity(tuple(validity.channels[c] for c in kept)))
            return self._apply_mask_to_result(acc.finalize(), mask, mask_align_to_image)
        finally:
            self._requested_output_head_override = prev

    def _predict_volume(self, images: torch.Tensor, mask, mask_align_to_image: bool, requested_head: Optional[str],
                        flat2d: bool) -> torch.Tensor:
        prev = self._requested_output_head_override
        self._requested_output_head_override = requested_head
        try:
            ops.require_device(images.device, "TTAPredictor")
            if images.shape[0] != 1:
                raise ValueError(f"device inference expects batch size 1; got batch {images.shape[0]}.")
            self._last_distributed_sharding_active = False
            self._last_skip_postprocess_on_rank = False
            if self.sliding_inferer is None and not flat2d and not self.is_distributed_sharding_enabled():
                # no sliding engine: the network sees every whole view directly, as in the reference (`_run_network` without an
                # inferer, tta.py:415-433) -- so it may change the spatial shape (a view that transposes unequal axes, a network
                # that pads its output; the mask check then speaks), which a window engine could not allow
                return self.predict_windows(images, mask=mask, mask_align_to_image=mask_align_to_image, requested_head=requested_head)
            tta = self._get_tta_cfg()
            enabled = tta is not None and getattr(tta, "enabled", True)
            if self._foreign_inferer() and not (enabled and bool(getattr(tta, "patch_first_local", False))):
                # a caller's own sliding inferer -- any `inferer(inputs=(1, C, *spatial), network=callable)` (MONAI's
                # SlidingWindowInferer, a wrapper around this package's engine): every whole view goes through it, as in the reference
                # (`_run_network`, tta.py:415-433).  Patch-first-local TTA never calls it (reference tta.py:880-1000 builds its own
                # window loop from the configuration); `_engine_for` does the same below.
                if flat2d or self.is_distributed_sharding_enabled():
                    raise TypeError("a sliding inferer without `accumulate` (not this package's EagerSlidingWindowEngine) supports neither "
                                    "2-D mode nor view sharding here")
                return self.predict_windows(images, mask=mask, mask_align_to_image=mask_align_to_image, requested_head=requested_head,
                                            run=lambda view: self.sliding_inferer(inputs=view, network=self._network_tensor))
            engine = self._engine_for(images)
            network = self._engine_network()
            combos = [([], None, 0)]
            if enabled:
                if flat2d:
                    # the configuration speaks of the 2 image axes (0 = y, 1 = x; tta_combinations.py:29-34): resolve them
                    # there, then name the same axes in the depth-1 volume (1 = y, 2 = x)
                    combos = [([int(a) + 1 for a in f], None if pl is None else tuple(int(a) + 1 for a in pl), k)
                              for f, pl, k in resolve_tta_augmentation_combinations(tta, spatial_dims=2)]
                else:
                    combos = self._spatial_augmentation_combinations(self._build_augmentation_combinations(tta, images.dim()))
            vol = images[0].to(torch.float32).contiguous()
            orig = tuple(int(v) for v in vol.shape[1:])
            self._last_device = vol.device

            def one_view(code, weight):
                value, weight = engine.accumulate(vol, network, view=code, weight=weight, add_weight=weight is None)
                ops.blend_finalize(value, weight, clamp=1e-4, act=nat.ACT_NONE)
                out = value
                if tuple(out.shape[1:]) != orig:
                    out = out[:, :orig[0], :orig[1], :orig[2]].contiguous()
                return self.apply_preprocessing(out.unsqueeze(0)), weight

            if len(combos) == 1 and combos[0] == ([], None, 0):
                result, _ = one_view(0, None)
                return self._apply_mask_to_result(result, mask, mask_align_to_image)

            # view sharding (tta.py:771-792): this rank runs views [rank::world]; the per-rank ensembles meet on rank 0
            sharded = self.is_distributed_sharding_enabled()
            self._last_distributed_sharding_active = sharded
            local = validate_view_shards(len(combos))[2] if sharded else list(range(len(combos)))
            if not bool(getattr(tta, "patch_first_local", False)) and self.sliding_inferer is not None:
                # reference predict() :1652-1660: without patch-first-local every view is a whole-volume pass
                result = self._predict_whole_volume_views(vol, engine, network, combos, getattr(tta, "ensemble_mode", "mean"),
                                                          local, sharded)
                return self._finish(result, mask, mask_align_to_image)
            for _f, pl, k in combos:   # same restriction (and message) as the reference, tta.py:1316-1340
                if pl is not None and k % 2:
                    img = tuple(int(v) for v in images.shape[2:])
                    roi = tuple(getattr(engine, "roi_size", None) or resolve_inferer_roi_size(self.cfg) or img)
                    if len({img[a] for a in pl}) != 1 or len({roi[a] for a in pl}) != 1:
                        raise ValueError(
                            "Patch-first local TTA only supports odd 90-degree rotations when the rotated axes "
                            f"have equal image and ROI sizes. Got rotation_plane={tuple(a + 2 for a in pl)}, "
                            f"image_size={img}, roi_size={roi}. Use flip-only TTA, constrain "
                            "rotations to equal-sized axes such as square XY inputs, or disable "
                            "`inference.test_time_augmentation.patch_first_local`.")
            codes = [view_code(f, pl, k) for f, pl, k in combos]
            ensemble_mode = getattr(tta, "ensemble_mode", "mean")
            from .tta_affinity import resolve_affinity_channel_groups_from_cfg
            if resolve_affinity_channel_groups_from_cfg(self.cfg):
                result = self._predict_affinity_views(vol, orig, engine, network, combos, codes, ensemble_mode, local, sharded)
                return self._finish(result, mask, mask_align_to_image)
            acc = None
            weight = None
            for n, i in enumerate(local):
                pred, weight = one_view(codes[i], weight)
                pred32 = pred if pred.dtype == torch.float32 else pred.float()
                if acc is None:
                    modes = _resolve_ensemble_mode_map(ensemble_mode, int(pred32.shape[1]))
                    bad = sorted(set(modes) - set(_MODE_CODE))
                    if bad:
                        raise ValueError(f"Unknown TTA ensemble modes: {bad}.")
                    acc = pred32.clone()
                    continue
                for c, mode in enumerate(modes):   # contiguous per-channel slabs of the (1,C,Z,Y,X) volume
                    ops.ensemble_update(acc[0, c], pred32[0, c].contiguous(), _MODE_CODE[mode], n + 1)
            if sharded:
                acc = self._reduce_views(acc, len(local), len(combos), modes)
            result = None if acc is None else acc.to(resolve_model_output_dtype(self.cfg))
            return self._finish(result, mask, mask_align_to_image)
        finally:
            self._requested_output_head_override = prev


def _predict_whole_volume_views(self, vol, engine, network, combos, ensemble_mode, local=None, sharded=False):
    """`patch_first_local: false` (reference _predict_prepared_tensor :806-878 + _run_ensemble :691-769): each view flips /
    rotates the WHOLE volume, runs its own sliding-window pass over the augmented geometry (its own window grid and weight
    map, so non-square rotations are fine), and the blended prediction is rotated / flipped back before activation and
    the ensemble.  The augmented volume is one device copy per view; the windows still gather from HBM as usual."""
    from .tta_affinity import ViewValidity, build_affinity_tta_plan, invert_view, resolve_affinity_channel_groups_from_cfg
    from .tta_ensemble import TTAEnsembleAccumulator
    has_affinity = bool(resolve_affinity_channel_groups_from_cfg(self.cfg))
    if has_affinity and sharded:
        raise NotImplementedError("whole-volume TTA (patch_first_local: false) of directional-affinity outputs cannot be sharded over "
                                  "ranks here; use patch_first_local: true (the reference default) with distributed_sharding")
    plan = ens = None
    acc = modes = None
    weights = {}
    local = list(range(len(combos))) if local is None else local
    for n, (flips, pl, k) in enumerate(combos[i] for i in local):
        x = vol
        if flips:
            x = torch.flip(x, dims=[int(a) + 1 for a in flips])
        if pl is not None and k > 0:
            x = torch.rot90(x, k=int(k), dims=(int(pl[0]) + 1, int(pl[1]) + 1))
        x = x.contiguous()
        shape = tuple(int(v) for v in x.shape[1:])
        w = weights.get(shape)
        value, w = engine.accumulate(x, network, view=0, weight=w, add_weight=w is None)
        weights[shape] = w
        ops.blend_finalize(value, w, clamp=1e-4, act=nat.ACT_NONE)
        if tuple(value.shape[1:]) != shape:
            value = value[:, :shape[0], :shape[1], :shape[2]]
        if has_affinity:
            # directional-affinity outputs: the inverse view also re-anchors the affinity channels and says where each channel
            # received real values; the streaming accumulator counts only those (reference tta.py:691-769, tta_affinity.py:350-393)
            raw = value.contiguous().unsqueeze(0)
            if plan is None:
                plan = build_affinity_tta_plan(self.cfg, augmentation_combinations=combos, num_raw=int(raw.shape[1]),
                                               requested_head=self._requested_output_head_override)
            view_index = local[n]
            inv, validity = invert_view(raw, flip_axes=flips, rotation_plane_spatial=pl, k=k,
                                        view_plan=None if plan is None else plan.views[view_index], tta_plan=plan)
            channels = int(inv.shape[1])
            sel = self._select_channel_indices(channels)
            kept = list(range(channels)) if sel is None else list(sel)
            done = self.apply_preprocessing(inv.contiguous())
            done = done if done.dtype == torch.float32 else done.float()
            if ens is None:
                partial = [] if plan is None else [j for j, c in enumerate(kept) if c in plan.partial_channels]
                ens = TTAEnsembleAccumulator(tuple(done.shape), dtype=resolve_model_output_dtype(self.cfg), device=done.device,
                                             mode_map=_resolve_ensemble_mode_map(ensemble_mode, int(done.shape[1])),
                                             partial_channels=partial, distributed_sharding=False, max_views=len(combos))
            ens.add(done, ViewValidity(tuple(validity.channels[c] for c in kept)))
            continue
        if pl is not None and k > 0:
            value = torch.rot90(value, k=-int(k), dims=(int(pl[0]) + 1, int(pl[1]) + 1))
        if flips:
            value = torch.flip(value, dims=[int(a) + 1 for a in flips])
        pred = self.apply_preprocessing(value.contiguous().unsqueeze(0))
        pred32 = pred if pred.dtype == torch.float32 else pred.float()
        if acc is None:
            modes = _resolve_ensemble_mode_map(ensemble_mode, int(pred32.shape[1]))
            bad = sorted(set(modes) - set(_MODE_CODE))
            if bad:
                raise ValueError(f"Unknown TTA ensemble modes: {bad}.")
            acc = pred32.clone()
            continue
        for c, mode in enumerate(modes):
            ops.ensemble_update(acc[0, c], pred32[0, c].contiguous(), _MODE_CODE[mode], n + 1)
    if ens is not None:
        return ens.finalize()
    if sharded:
        acc = self._reduce_views(acc, len(local), len(combos), modes)
    return None if acc is None else acc.to(resolve_model_output_dtype(self.cfg))


def _predict_affinity_views(self, vol, orig, engine, network, combos, codes, ensemble_mode, local=None, sharded=False):
    """Directional-affinity outputs (reference tta.py:1036-1275 + tta_affinity.py + tta_ensemble.py): every view is
    blended with its channel map (channels re-anchored inside each window by the blending kernel), fully valid
    channels are normalised by the shared weight and ensembled as usual, shifted ("partial") channels by the weight of
    their own shift -- whose support is also their validity in the mean / min / max."""
    from .tta_affinity import build_affinity_tta_plan
    dev = vol.device
    plan = None
    acc = None
    w_full = None
    w_shift = {}
    stats = counts = None
    modes = sel = partial_sel = None
    local = list(range(len(combos))) if local is None else local
    for n, i in enumerate(local):
        code = codes[i]
        if plan is None:
            # the plan needs the raw channel count: probe one window of the identity view
            probe = engine._run_network(network, ops.gather_windows(vol, engine.plan(orig)[1][:1], engine.roi_size,
                                                                    pad_mode="constant", cval=engine.cval))
            plan = build_affinity_tta_plan(self.cfg, augmentation_combinations=combos, num_raw=int(probe.shape[-1]),
                                           requested_head=self._requested_output_head_override)
            for sh in sorted(plan.shifts):
                w_shift[tuple(sh)] = engine.shifted_weight(orig, sh, dev)
        vp = plan.views[i]
        value, w_full = engine.accumulate(vol, network, view=code, weight=w_full, add_weight=w_full is None,
                                          chan_map=vp.channel_map(plan.num_channels))
        covers = [None] * plan.num_channels
        partial = sorted(plan.partial_channels)
        if partial:
            # normalise the shared-weight channels in place (1e-4 clamp), then overwrite the partial ones
            raw_partial = {c: value[c].clone() for c in partial}
        ops.blend_finalize(value, w_full, clamp=1e-4, act=nat.ACT_NONE)
        for c in partial:
            sh = vp.shift_for_channel(c)
            wk = w_full if sh is None else w_shift[tuple(sh)]
            v = raw_partial[c]
            ops.normalize_covered(v, wk)
            value[c].copy_(v)
            covers[c] = wk
        out = value
        crop = tuple(out.shape[1:]) != orig
        if crop:
            out = out[:, :orig[0], :orig[1], :orig[2]].contiguous()
        nraw = int(out.shape[0])
        pred = self.apply_preprocessing(out.unsqueeze(0))
        pred = pred if pred.dtype == torch.float32 else pred.float()
        if acc is None:
            sel = self._select_channel_indices(nraw)
            sel = list(range(nraw)) if sel is None else [int(v) for v in sel]
            modes = _resolve_ensemble_mode_map(ensemble_mode, int(pred.shape[1]))
            bad = sorted(set(modes) - set(_MODE_CODE))
            if bad:
                raise ValueError(f"Unknown TTA ensemble modes: {bad}.")
            partial_sel = [j for j, c in enumerate(sel) if c in plan.partial_channels]
            acc = pred.clone()
            if partial_sel:
                shape = (len(partial_sel),) + tuple(pred.shape[2:])
                stats = torch.empty(shape, dtype=torch.float32, device=dev)
                for pi, j in enumerate(partial_sel):
                    stats[pi].fill_(0.0 if modes[j] == "mean" else (float("inf") if modes[j] == "min" else float("-inf")))
                counts = torch.zeros(shape, dtype=torch.float32, device=dev)
        else:
            for j, mode in enumerate(modes):
                if j not in partial_sel:
                    ops.ensemble_update(acc[0, j], pred[0, j].contiguous(), _MODE_CODE[mode], n + 1)
        for pi, j in enumerate(partial_sel):
            cov = covers[sel[j]]
            if cov is not None and crop:
                cov = cov[:orig[0], :orig[1], :orig[2]].contiguous()
            ops.ensemble_update_masked(stats[pi], counts[pi], pred[0, j].contiguous(), cov, _MODE_CODE[modes[j]])
    if sharded:
        red = self._reduce_views(acc, len(local), len(combos), modes, skip=partial_sel, stats=stats, counts=counts,
                                 partial_modes=[modes[j] for j in partial_sel]) if partial_sel else \
            self._reduce_views(acc, len(local), len(combos), modes)
        if red is None:
            return None
        if partial_sel:
            acc, stats, counts = red
        else:
            acc = red
    for pi, j in enumerate(partial_sel or []):
        if bool((counts[pi] == 0).any()):
            first = tuple(int(v) for v in torch.nonzero(counts[pi] == 0)[0])
            raise RuntimeError(f"TTA ensemble has zero valid contributions for channel {j} at voxel index {(0,) + first}.")
        ops.ensemble_finalize_masked(stats[pi], counts[pi], acc[0, j], _MODE_CODE[modes[j]])
    return acc.to(resolve_model_output_dtype(self.cfg))


TTAPredictor._predict_affinity_views = _predict_affinity_views
TTAPredictor._predict_whole_volume_views = _predict_whole_volume_views

__all__ = ["TTAPredictor", "view_code"]
