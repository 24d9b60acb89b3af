"""Chunk grid (contract of the reference's connectomics/chunked/chunk_grid.py:13-43): ceil-div chunk counts,
row-major (z outermost) order, the last chunk per axis clipped to the volume; key = "z{z}_y{y}_x{x}"."""
from __future__ import annotations

import itertools
from dataclasses import dataclass
from typing import List, Sequence, Tuple

Int3 = Tuple[int, int, int]


@dataclass(frozen=True)
class ChunkRef:
    """One cell of the grid: its (z, y, x) grid index and the half-open voxel box [start, stop) it covers."""
    index: Int3
    start: Int3
    stop: Int3

    @property
    def key(self) -> str:
        iz, iy, ix = self.index
        return f"z{iz}_y{iy}_x{ix}"

    @property
    def shape(self) -> Int3:
        return (self.stop[0] - self.start[0], self.stop[1] - self.start[1], self.stop[2] - self.start[2])

    @property
    def slices(self):
        return tuple(slice(lo, hi) for lo, hi in zip(self.start, self.stop))


def build_chunk_grid(volume_shape: Sequence[int], chunk_shape: Sequence[int]) -> List[ChunkRef]:
    dims = [int(v) for v in volume_shape]
    cell = [int(v) for v in chunk_shape]
    if len(dims) != 3 or len(cell) != 3:
        raise ValueError("volume_shape and chunk_shape must both be length-3 tuples.")
    if min(cell) <= 0:
        raise ValueError(f"chunk_shape must be positive, got {tuple(cell)}")
    counts = [(d + c - 1) // c for d, c in zip(dims, cell)]
    grid = []
    for idx in itertools.product(*(range(n) for n in counts)):        # z outermost, x fastest
        lo = tuple(i * c for i, c in zip(idx, cell))
        hi = tuple(min(a + c, d) for a, c, d in zip(lo, cell, dims))
        grid.append(ChunkRef(index=tuple(idx), start=lo, stop=hi))
    return grid


__all__ = ["ChunkRef", "build_chunk_grid"]
