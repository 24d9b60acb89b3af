"""Local neuroglancer "precomputed" image layer, raw encoding -- what the reference reaches through `cloudvolume` when
`inference.chunking.precomputed` streams chunk predictions straight into a layer that ABISS / Seuron read
(connectomics/inference/chunked.py:68-130 `_open_precomputed_layer`, :590-612 the per-chunk write).  cloudvolume is not in this
image; the on-disk format is small and public, so it is written directly:

  <layer>/info                       JSON: type "image", data_type, num_channels, one scale {key, size (xyz), resolution (xyz),
                                     voxel_offset, chunk_sizes [[cx, cy, cz]], encoding "raw"} -- the fields of
                                     CloudVolume.create_new_info as the reference calls it
  <layer>/<key>/x0-x1_y0-y1_z0-z1.gz one gzip-compressed file per storage chunk (compress=True in the reference's CloudVolume
                                     handle); payload = the chunk's voxels as little-endian `data_type`, Fortran order over
                                     (x, y, z, channel), i.e. the bytes of a C-contiguous (C, Z, Y, X) array.  Chunks on the high
                                     faces are clipped to the volume.

Parity note: written from the format specification, not checked against cloudvolume itself (absent here); the test reads the layer
back with an independent decoder of the same specification and compares it with the HDF5 chunk path."""
from __future__ import annotations

import gzip
import json
import os
import time
from pathlib import Path
from typing import Sequence

import numpy as np

__all__ = ["PrecomputedLayer", "open_precomputed_layer", "validate_precomputed_alignment"]

_DTYPES = {"uint8", "uint16", "uint32", "uint64", "int8", "int16", "int32", "float32"}


def validate_precomputed_alignment(chunk_shape_zyx: Sequence[int], chunk_size_xyz: Sequence[int]) -> None:
    """Inference chunks must tile the layer's storage chunks: ranks write disjoint inference chunks concurrently, and a chunk that
    does not end on storage-chunk boundaries would make two ranks touch one file (reference chunked.py:168-190, same message)."""
    shape_xyz = [int(v) for v in reversed(list(chunk_shape_zyx))]
    bad = [f"{axis}: inference chunk {shape_xyz[i]} is not a multiple of storage chunk {int(chunk_size_xyz[i])}"
           for i, axis in enumerate("xyz") if int(chunk_size_xyz[i]) <= 0 or shape_xyz[i] % int(chunk_size_xyz[i]) != 0]
    if bad:
        raise ValueError("chunking.precomputed_chunk_size must divide the inference chunk on every axis so concurrent chunk "
                         "writes never straddle a storage chunk. " + "; ".join(bad))


class PrecomputedLayer:
    """One-scale raw-encoded image layer on the local filesystem."""

    def __init__(self, layer_dir):
        self.dir = Path(layer_dir)
        info = json.loads((self.dir / "info").read_text())
        scale = info["scales"][0]
        if info.get("type") != "image" or scale.get("encoding") != "raw":
            raise ValueError(f"{self.dir}: only raw-encoded image layers are supported")
        self.info = info
        self.dtype = np.dtype(info["data_type"]).newbyteorder("<")
        self.num_channels = int(info["num_channels"])
        self.key = str(scale["key"])
        self.size_xyz = tuple(int(v) for v in scale["size"])
        self.offset_xyz = tuple(int(v) for v in scale.get("voxel_offset", (0, 0, 0)))
        self.chunk_xyz = tuple(int(v) for v in scale["chunk_sizes"][0])

    # ---- naming
    def _chunk_path(self, lo, hi) -> Path:
        name = "_".join(f"{lo[a] + self.offset_xyz[a]}-{hi[a] + self.offset_xyz[a]}" for a in range(3))
        return self.dir / self.key / (name + ".gz")

    def _cells(self, lo, hi):
        """Storage chunks (clipped to the volume) that intersect the box [lo, hi) given in xyz voxels from the layer origin."""
        rng = [range(lo[a] // self.chunk_xyz[a], -(-hi[a] // self.chunk_xyz[a])) for a in range(3)]
        for ix in rng[0]:
            for iy in rng[1]:
                for iz in rng[2]:
                    c_lo = (ix * self.chunk_xyz[0], iy * self.chunk_xyz[1], iz * self.chunk_xyz[2])
                    c_hi = tuple(min(c_lo[a] + self.chunk_xyz[a], self.size_xyz[a]) for a in range(3))
                    yield c_lo, c_hi

    # ---- write / read
    def write_czyx(self, start_zyx: Sequence[int], block: np.ndarray) -> list:
        """Write a (C, Z, Y, X) block whose origin is `start_zyx`; the block has to cover whole storage chunks (or end at the
        volume face).  -> the files written."""
        if block.ndim != 4 or block.shape[0] != self.num_channels:
            raise ValueError(f"block must be (C={self.num_channels}, Z, Y, X), got {tuple(block.shape)}")
        lo = tuple(int(start_zyx[2 - a]) for a in range(3))                               # xyz
        hi = tuple(lo[a] + int(block.shape[3 - a]) for a in range(3))
        if any(lo[a] < 0 or hi[a] > self.size_xyz[a] for a in range(3)):
            raise ValueError(f"block [{lo}, {hi}) lies outside the layer of size {self.size_xyz}")
        written = []
        (self.dir / self.key).mkdir(parents=True, exist_ok=True)
        for c_lo, c_hi in self._cells(lo, hi):
            if any(c_lo[a] < lo[a] or c_hi[a] > hi[a] for a in range(3)):
                raise ValueError(f"block [{lo}, {hi}) does not cover storage chunk [{c_lo}, {c_hi}) completely "
                                 f"(chunk size {self.chunk_xyz}): writes must be storage-chunk aligned")
            sub = block[:, c_lo[2] - lo[2]:c_hi[2] - lo[2], c_lo[1] - lo[1]:c_hi[1] - lo[1], c_lo[0] - lo[0]:c_hi[0] - lo[0]]
            payload = np.ascontiguousarray(sub, dtype=self.dtype).tobytes()              # C-order (C,Z,Y,X) == F-order (x,y,z,c)
            path = self._chunk_path(c_lo, c_hi)
            tmp = path.with_name(path.name + f".tmp{os.getpid()}")
            with open(tmp, "wb") as fh:
                fh.write(gzip.compress(payload, compresslevel=6, mtime=0))
            os.replace(tmp, path)
            written.append(path)
        return written

    def read_czyx(self, start_zyx: Sequence[int], stop_zyx: Sequence[int], *, fill_missing: bool = True) -> np.ndarray:
        """(C, Z, Y, X) array of the box; storage chunks without a file read as zeros (fill_missing, as the reference opens the
        layer) or raise."""
        lo = tuple(int(start_zyx[2 - a]) for a in range(3))
        hi = tuple(int(stop_zyx[2 - a]) for a in range(3))
        out = np.zeros((self.num_channels, hi[2] - lo[2], hi[1] - lo[1], hi[0] - lo[0]), dtype=self.dtype.newbyteorder("="))
        for c_lo, c_hi in self._cells(lo, hi):
            path = self._chunk_path(c_lo, c_hi)
            if not path.exists():
                if fill_missing:
                    continue
                raise FileNotFoundError(path)
            shape = (self.num_channels, c_hi[2] - c_lo[2], c_hi[1] - c_lo[1], c_hi[0] - c_lo[0])
            cell = np.frombuffer(gzip.decompress(path.read_bytes()), dtype=self.dtype).reshape(shape)
            i_lo = tuple(max(c_lo[a], lo[a]) for a in range(3))
            i_hi = tuple(min(c_hi[a], hi[a]) for a in range(3))
            out[:, i_lo[2] - lo[2]:i_hi[2] - lo[2], i_lo[1] - lo[1]:i_hi[1] - lo[1], i_lo[0] - lo[0]:i_hi[0] - lo[0]] = \
                cell[:, i_lo[2] - c_lo[2]:i_hi[2] - c_lo[2], i_lo[1] - c_lo[1]:i_hi[1] - c_lo[1], i_lo[0] - c_lo[0]:i_hi[0] - c_lo[0]]
        return out


def open_precomputed_layer(layer_dir, *, volume_size_xyz, num_channels: int, data_type: str, resolution_xyz,
                           chunk_size_xyz) -> PrecomputedLayer:
    """Open the output layer, creating its `info` exactly once: several ranks arrive here together, so creation is guarded by an
    O_EXCL lock file -- the winner commits `info`, the others wait for it (reference chunked.py:68-130)."""
    layer_dir = Path(layer_dir)
    layer_dir.mkdir(parents=True, exist_ok=True)
    info_path = layer_dir / "info"
    if str(data_type) not in _DTYPES:
        raise ValueError(f"precomputed layers store one of {sorted(_DTYPES)}, got {data_type!r}")
    res = [int(v) for v in resolution_xyz]
    info = {"@type": "neuroglancer_multiscale_volume", "type": "image", "data_type": str(data_type),
            "num_channels": int(num_channels),
            "scales": [{"key": "_".join(str(v) for v in res), "size": [int(v) for v in volume_size_xyz],
                        "resolution": res, "voxel_offset": [0, 0, 0],
                        "chunk_sizes": [[int(v) for v in chunk_size_xyz]], "encoding": "raw"}]}
    if not info_path.exists():
        lock = layer_dir / ".info.lock"
        try:
            # a lock older than the wait below belongs to a run that died between taking it and committing `info`
            if time.time() - lock.stat().st_mtime > 120.0:
                lock.unlink()
        except OSError:
            pass
        try:
            fd = os.open(str(lock), os.O_CREAT | os.O_EXCL | os.O_WRONLY)
        except FileExistsError:
            fd = None
        if fd is not None:
            try:
                tmp = layer_dir / f".info.tmp{os.getpid()}"
                tmp.write_text(json.dumps(info))
                os.replace(tmp, info_path)
            finally:
                os.close(fd)
                try:
                    lock.unlink()           # the lock only guards the creation of `info`
                except OSError:
                    pass
        else:
            for _ in range(600):            # ~60 s; the writer only has one small file to commit
                if info_path.exists():
                    break
                time.sleep(0.1)
            if not info_path.exists():
                raise RuntimeError(f"Timed out waiting for precomputed info at {info_path}")
    # an existing layer must describe THIS run: chunks of another volume size / dtype / channel count / chunking would be
    # written mis-shaped or mis-typed into it
    have = json.loads(info_path.read_text())
    hs, ws = (have.get("scales") or [{}])[0], info["scales"][0]
    diffs = [f"{k}: layer has {a!r}, this run needs {b!r}" for k, a, b in
             [("data_type", have.get("data_type"), info["data_type"]), ("num_channels", have.get("num_channels"), info["num_channels"]),
              ("size", hs.get("size"), ws["size"]), ("chunk_sizes", hs.get("chunk_sizes"), ws["chunk_sizes"]),
              ("encoding", hs.get("encoding"), ws["encoding"])] if a != b]
    if diffs:
        raise ValueError(f"precomputed layer {layer_dir} was created with different parameters ({'; '.join(diffs)}); "
                         "write to a fresh directory or remove the old layer")
    return PrecomputedLayer(layer_dir)
