"""Chunk grid (contract of the reference's connectomics/chunked/chunk_grid.py:13-43): ceil-div chunk counts,
row-major (z outermost) order, the last chunk per axis clipped to the volume; key = "z{z}_y{y}_x{x}"."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Sequence


@dataclass(frozen=True)
class ChunkRef:
    index: tuple[int, int, int]
    start: tuple[int, int, int]
    stop: tuple[int, int, int]

    @property
    def key(self) -> str:
        return "z{}_y{}_x{}".format(*self.index)

    @property
    def shape(self) -> tuple[int, int, int]:
        return tuple(b - a for a, b in zip(self.start, self.stop))

    @property
    def slices(self) -> tuple[slice, slice, slice]:
        return tuple(slice(a, b) for a, b in zip(self.start, self.stop))


def build_chunk_grid(volume_shape: Sequence[int], chunk_shape: Sequence[int]) -> list[ChunkRef]:
    vol = tuple(int(v) for v in volume_shape)
    ch = tuple(int(v) for v in chunk_shape)
    if len(vol) != 3 or len(ch) != 3:
        raise ValueError("volume_shape and chunk_shape must both be length-3 tuples.")
    if any(c <= 0 for c in ch):
        raise ValueError(f"chunk_shape must be positive, got {ch}")
    nz, ny, nx = (-(-v // c) for v, c in zip(vol, ch))
    out = []
    for iz in range(nz):
        for iy in range(ny):
            for ix in range(nx):
                idx = (iz, iy, ix)
                start = tuple(i * c for i, c in zip(idx, ch))
                stop = tuple(min(s + c, v) for s, c, v in zip(start, ch, vol))
                out.append(ChunkRef(index=idx, start=start, stop=stop))
    return out


__all__ = ["ChunkRef", "build_chunk_grid"]
