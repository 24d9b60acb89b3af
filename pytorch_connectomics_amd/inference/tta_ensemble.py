"""`TTAEnsembleAccumulator` -- the reference's streaming, validity-aware ensemble of TTA views
(connectomics/inference/tta_ensemble.py:13-211) as a public, device-resident object.

`TTAPredictor.predict` (whole volumes) does not go through this class (its ensemble is fused with per-view normalisation, and the
affinity channel moves are index math inside the blending kernel); `TTAPredictor.predict_windows` -- the lazy / chunked path, one
accumulator per window batch -- and callers of the reference API that hold whole canonical predictions do.  Same constructor / `add` / `finalize` contract; the statistics live in HBM and every update is one of the
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
    """Statistics are stored CHANNEL-MAJOR, (C, N, *spatial): a channel -- and a run of consecutive channels that share an ensemble
    mode -- is one contiguous block, so a view is streamed in with one kernel launch per run directly on the stored statistics
    (round 3 kept (N, C, ...) and paid three strided-to-contiguous copies per channel and view: ADVICE r03).  `legacy_result`,
    `partial_statistics`, `partial_counts` are the reference's attribute names and shapes, as views."""

    def __init__(self, shape: Sequence[int], *, dtype: torch.dtype, device, mode_map: Sequence[str],
                 partial_channels: Sequence[int], distributed_sharding: bool, max_views: int) -> None:
        self.device = torch.device(device)
        ops.require_device(self.device, "TTAEnsembleAccumulator")
        self.shape, self.dtype = tuple(map(int, shape)), dtype
        self.mode_map, self.partial_channels, self.full_channels = _channel_plan(self.shape, mode_map, partial_channels)
        self.distributed_sharding, self.max_views = bool(distributed_sharding), int(max_views)
        self.num_predictions = 0
        on_device = dict(device=self.device, dtype=torch.float32)
        n, c = self.shape[0], self.shape[1]
        self._stat = torch.zeros((c, n, *self.shape[2:]), **on_device)
        slabs = (len(self.partial_channels), n, *self.shape[2:])
        self._pcount = torch.zeros(slabs, **on_device)
        self._pstat = torch.empty(slabs, **on_device)
        for slot, channel in enumerate(self.partial_channels):
            self._pstat[slot] = _IDENTITY[self.mode_map[channel]]
        # runs of consecutive fully valid channels with one ensemble mode: (first, stop, mode)
        self._runs = []
        for ch in self.full_channels:
            if self._runs and self._runs[-1][1] == ch and self._runs[-1][2] == self.mode_map[ch]:
                self._runs[-1][1] = ch + 1
            else:
                self._runs.append([ch, ch + 1, self.mode_map[ch]])

    # The reference keeps these three as plain (N, C, ...) tensor attributes that its distributed TTA reduction reads, reduces and
    # assigns back.  Here they are VIEWS of the channel-major stores (non-contiguous: an in-place collective needs `.contiguous()`
    # first, or the stores themselves -- inference/tta.py reduces `_stat` / `_pstat` / `_pcount` directly); assignment copies the
    # given (N, C, ...) tensor into the store, so `acc.legacy_result = reduced` works as it does on the reference's class.
    def _store_property(name):            # noqa: N805
        def get(self):
            return getattr(self, name).transpose(0, 1)

        def set_(self, value):
            store = getattr(self, name)
            value = torch.as_tensor(value, device=store.device)
            if tuple(value.shape) != tuple(store.transpose(0, 1).shape):
                raise ValueError(f"expected shape {tuple(store.transpose(0, 1).shape)}, got {tuple(value.shape)}")
            store.copy_(value.transpose(0, 1))
        return property(get, set_)

    legacy_result = _store_property("_stat")
    partial_statistics = _store_property("_pstat")
    partial_counts = _store_property("_pcount")
    del _store_property

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
        pcn = prediction.to(device=self.device, dtype=torch.float32).transpose(0, 1)         # (C, N, *spatial) view
        for c0, c1, mode in self._runs:
            acc = self._stat[c0:c1]                                  # contiguous block: updated in place
            if self.num_predictions == 0:
                acc.copy_(pcn[c0:c1])
            elif mode == "mean" and self.distributed_sharding:
                acc += pcn[c0:c1]                                    # shards sum, the reduce divides (reference :95-97)
            else:
                ops.ensemble_update(acc, pcn[c0:c1].contiguous(), _MODE[mode], self.num_predictions + 1)
        for pi, c in enumerate(self.partial_channels):
            values = pcn[c].contiguous()
            ops.ensemble_update_masked(self._pstat[pi], self._pcount[pi], values, self._cover(validity.channels[c], values),
                                       _MODE[self.mode_map[c]])
        self.num_predictions += 1

    def finalize(self, *, legacy_result: Optional[torch.Tensor] = None, partial_statistics: Optional[torch.Tensor] = None,
                 partial_counts: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The aggregate in `dtype`; a partial channel without any valid contribution somewhere is an error.  The optional arguments
        are (N, C, ...)-shaped replacements of the stored statistics (a distributed reduce hands back the reduced tensors)."""
        cn = lambda t: t.to(device=self.device, dtype=torch.float32).transpose(0, 1)      # noqa: E731
        result = (self._stat if legacy_result is None else cn(legacy_result)).clone(memory_format=torch.contiguous_format)
        stats = self._pstat if partial_statistics is None else cn(partial_statistics).contiguous()
        counts = self._pcount if partial_counts is None else cn(partial_counts).contiguous()
        if self.partial_channels:
            empty = counts == 0
            if bool(empty.any()):                                    # ONE device -> host check for all partial channels
                pi, *where = (int(v) for v in torch.nonzero(empty)[0])
                raise RuntimeError(f"TTA ensemble has zero valid contributions for channel {self.partial_channels[pi]} at voxel index "
                                   f"{tuple(where)}.")
            for pi, c in enumerate(self.partial_channels):
                ops.ensemble_finalize_masked(stats[pi], counts[pi], result[c], _MODE[self.mode_map[c]])
        return result.transpose(0, 1).contiguous().to(self.dtype)


__all__ = ["TTAEnsembleAccumulator"]
