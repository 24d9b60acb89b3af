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


class TTAEnsembleAccumulator:
    def __init__(self, shape: Sequence[int], *, dtype: torch.dtype, device, mode_map: Sequence[str],
                 partial_channels: Sequence[int], distributed_sharding: bool, max_views: int) -> None:
        self.shape = tuple(int(v) for v in shape)
        self.dtype = dtype
        self.device = torch.device(device)
        ops.require_device(self.device, "TTAEnsembleAccumulator")
        self.mode_map = tuple(str(m) for m in mode_map)
        if len(self.shape) < 3 or len(self.mode_map) != self.shape[1]:
            raise ValueError(f"Invalid TTA accumulator shape/modes: shape={self.shape}, modes={len(self.mode_map)}.")
        unknown = sorted(set(self.mode_map) - set(_MODE))
        if unknown:
            raise ValueError(f"Unknown TTA ensemble modes: {unknown}.")
        self.partial_channels = tuple(sorted({int(c) for c in partial_channels}))
        if any(c < 0 or c >= self.shape[1] for c in self.partial_channels):
            raise ValueError(f"Partial TTA channels {self.partial_channels} are invalid for {self.shape[1]} output channels.")
        self.full_channels = tuple(c for c in range(self.shape[1]) if c not in set(self.partial_channels))
        self.distributed_sharding = bool(distributed_sharding)
        self.max_views = int(max_views)
        self.num_predictions = 0
        self.legacy_result = torch.zeros(self.shape, device=self.device, dtype=torch.float32)
        pshape = (self.shape[0], len(self.partial_channels)) + self.shape[2:]
        self.partial_statistics = torch.empty(pshape, device=self.device, dtype=torch.float32)
        for pi, c in enumerate(self.partial_channels):
            mode = self.mode_map[c]
            self.partial_statistics[:, pi].fill_(0.0 if mode == "mean" else float("inf") if mode == "min" else float("-inf"))
        self.partial_counts = torch.zeros(pshape, device=self.device, dtype=torch.float32)

    @property
    def has_partial_channels(self) -> bool:
        return bool(self.partial_channels)

    def _cover(self, validity, like: torch.Tensor) -> Optional[torch.Tensor]:
        """None (valid everywhere) | tuple of slices (valid box) | bool tensor  ->  fp32 cover mask shaped like `like` or None."""
        if validity is None:
            return None
        if isinstance(validity, tuple):
            cover = torch.zeros_like(like)
            cover[(slice(None),) + tuple(validity)] = 1.0
            return cover
        mask = validity.to(device=like.device)
        if mask.dim() == like.dim() - 1:
            mask = mask.unsqueeze(0)
        if mask.dim() != like.dim():
            raise ValueError(f"TTA validity tensor rank {mask.dim()} does not match channel value rank {like.dim()}.")
        if mask.shape[0] == 1 and like.shape[0] != 1:
            mask = mask.expand(like.shape[0], *mask.shape[1:])
        if tuple(mask.shape) != tuple(like.shape):
            raise ValueError(f"TTA validity shape {tuple(mask.shape)} does not match channel value shape {tuple(like.shape)}.")
        return mask.to(torch.float32).contiguous()

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
