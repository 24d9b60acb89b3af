"""`TTAEnsembleAccumulator` -- the reference's streaming, validity-aware ensemble of TTA views
(connectomics/inference/tta_ensemble.py:13-211) as a public, device-resident object.

`TTAPredictor` does not go through this class (its ensemble is fused with per-view normalisation, and the affinity channel moves
are index math inside the blending kernel); the class exists for callers of the reference API that hold whole canonical
predictions.  Same constructor / `add` / `finalize` contract; the statistics live in HBM and every update is one of the
ensemble kernels of csrc/window_kernels.hip (`pytc_ensemble_update`, `pytc_ensemble_update_masked`,
`pytc_ensemble_finalize_masked`).  Differences from the reference, both inside its contract: partial-channel counts are fp32
(exact to 2^24 views) instead of uint8 / int16, and a box validity is expanded to a cover mask on the device.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from .. import hip_ops as ops
from .tta_affinity import ViewValidity

_MODE = {"mean": 0, "min": 1, "max": 2}
_IDENTITY = {"mean": 0.0, "min": float("inf"), "max": float("-inf")}       # the statistic before any view has contributed


def _channel_plan(shape, mode_map, partial_channels):
    """Validated (modes, partial channel tuple, full channel tuple) of an (N, C, *spatial) ensemble."""
    modes = tuple(str(m) for m in mode_map)
    channels = shape[1] if len(shape) >= 3 else -1
    if channels != len(modes):
        raise ValueError(f"Invalid TTA accumulator shape/modes: shape={shape}, modes={len(modes)}.")
    strangers = sorted(set(modes).difference(_MODE))
    if strangers:
        raise ValueError(f"Unknown TTA ensemble modes: {strangers}.")
    partial = tuple(sorted(set(map(int, partial_channels))))
    if partial and not (0 <= partial[0] and partial[-1] < channels):
        raise ValueError(f"Partial TTA channels {partial} are invalid for {channels} output channels.")
    return modes, partial, tuple(c for c in range(channels) if c not in partial)


class TTAEnsembleAccumulator:
    def __init__(self, shape: Sequence[int], *, dtype: torch.dtype, device, mode_map: Sequence[str],
                 partial_channels: Sequence[int], distributed_sharding: bool, max_views: int) -> None:
        self.device = torch.device(device)
        ops.require_device(self.device, "TTAEnsembleAccumulator")
        self.shape, self.dtype = tuple(map(int, shape)), dtype
        self.mode_map, self.partial_channels, self.full_channels = _channel_plan(self.shape, mode_map, partial_channels)
        self.distributed_sharding, self.max_views = bool(distributed_sharding), int(max_views)
        self.num_predictions = 0
        on_device = dict(device=self.device, dtype=torch.float32)
        self.legacy_result = torch.zeros(self.shape, **on_device)
        slabs = (self.shape[0], len(self.partial_channels), *self.shape[2:])
        self.partial_counts = torch.zeros(slabs, **on_device)
        self.partial_statistics = torch.empty(slabs, **on_device)
        for slot, channel in enumerate(self.partial_channels):
            self.partial_statistics[:, slot] = _IDENTITY[self.mode_map[channel]]

    @property
    def has_partial_channels(self) -> bool:
        return len(self.partial_channels) > 0

    def _cover(self, validity, like: torch.Tensor) -> Optional[torch.Tensor]:
        """None (valid everywhere) | tuple of slices (valid box) | bool tensor  ->  fp32 cover mask shaped like `like` or None."""
        if validity is None:
            return None
        if isinstance(validity, tuple):                               # a box: ones inside, zeros outside, built on the device
            cover = torch.zeros_like(like)
            cover[(slice(None), *validity)] = 1.0
            return cover
        mask = validity.to(device=like.device)
        if mask.dim() + 1 == like.dim():                              # no batch axis: one mask for every sample
            mask = mask[None]
        if mask.dim() != like.dim():
            raise ValueError(f"TTA validity tensor rank {mask.dim()} does not match channel value rank {like.dim()}.")
        if mask.shape[0] == 1 and like.shape[0] > 1:
            mask = mask.expand(like.shape[0], *mask.shape[1:])
        if mask.shape != like.shape:
            raise ValueError(f"TTA validity shape {tuple(mask.shape)} does not match channel value shape {tuple(like.shape)}.")
        return mask.float().contiguous()

    def add(self, prediction: torch.Tensor, validity: ViewValidity) -> None:
        """Stream one canonical prediction (N, C, *spatial) into the accumulator."""
        if tuple(prediction.shape) != self.shape:
            raise ValueError(f"TTA prediction shape {tuple(prediction.shape)} does not match accumulator shape {self.shape}.")
        if len(validity.channels) != self.shape[1]:
            raise ValueError(f"TTA validity describes {len(validity.channels)} channels, expected {self.shape[1]}.")
        pred = prediction.to(device=self.device, dtype=torch.float32)
        for c in self.full_channels:
            incoming = pred[:, c].contiguous()
            if self.num_predictions == 0:
                self.legacy_result[:, c].copy_(incoming)
            elif self.mode_map[c] == "mean" and self.distributed_sharding:
                self.legacy_result[:, c] += incoming               # shards sum, the reduce divides (reference :95-97)
            else:
                acc = self.legacy_result[:, c].contiguous()
                ops.ensemble_update(acc, incoming, _MODE[self.mode_map[c]], self.num_predictions + 1)
                self.legacy_result[:, c].copy_(acc)
        for pi, c in enumerate(self.partial_channels):
            values = pred[:, c].contiguous()
            stat, count = self.partial_statistics[:, pi].contiguous(), self.partial_counts[:, pi].contiguous()
            ops.ensemble_update_masked(stat, count, values, self._cover(validity.channels[c], values), _MODE[self.mode_map[c]])
            self.partial_statistics[:, pi].copy_(stat)
            self.partial_counts[:, pi].copy_(count)
        self.num_predictions += 1

    def finalize(self, *, legacy_result: Optional[torch.Tensor] = None, partial_statistics: Optional[torch.Tensor] = None,
                 partial_counts: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The aggregate in `dtype`; a partial channel without any valid contribution somewhere is an error."""
        result = (self.legacy_result if legacy_result is None else legacy_result).to(torch.float32).clone()
        stats = self.partial_statistics if partial_statistics is None else partial_statistics
        counts = self.partial_counts if partial_counts is None else partial_counts
        for pi, c in enumerate(self.partial_channels):
            cnt = counts[:, pi].to(torch.float32).contiguous()
            empty = cnt == 0
            if bool(empty.any()):
                first = tuple(int(v) for v in torch.nonzero(empty)[0])
                raise RuntimeError(f"TTA ensemble has zero valid contributions for channel {c} at voxel index {first}.")
            out = torch.empty_like(cnt)
            ops.ensemble_finalize_masked(stats[:, pi].to(torch.float32).contiguous(), cnt, out, _MODE[self.mode_map[c]])
            result[:, c].copy_(out)
        return result.to(self.dtype)


__all__ = ["TTAEnsembleAccumulator"]
