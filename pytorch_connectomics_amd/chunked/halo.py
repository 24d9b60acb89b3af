"""Halo coordinate math (contract of the reference's connectomics/chunked/halo.py:12-40)."""
from __future__ import annotations

from typing import Sequence

from .chunk_grid import ChunkRef


def resolve_halo_region(chunk: ChunkRef, input_shape: Sequence[int], *, halo: Sequence[int] = (0, 0, 0),
                        crop_before: Sequence[int] = (0, 0, 0)):
    """-> (read_start, read_stop, local_core_slices): the chunk core shifted by `crop_before`, grown by `halo`
    and clipped to the input volume, plus the slices selecting the core inside that read window."""
    shape = [int(v) for v in input_shape]
    core_lo = [int(chunk.start[a]) + int(crop_before[a]) for a in range(3)]
    core_hi = [int(chunk.stop[a]) + int(crop_before[a]) for a in range(3)]
    read_lo = tuple(max(0, core_lo[a] - int(halo[a])) for a in range(3))
    read_hi = tuple(min(shape[a], core_hi[a] + int(halo[a])) for a in range(3))
    local = tuple(slice(core_lo[a] - read_lo[a], core_hi[a] - read_lo[a]) for a in range(3))
    return read_lo, read_hi, local


__all__ = ["resolve_halo_region"]
