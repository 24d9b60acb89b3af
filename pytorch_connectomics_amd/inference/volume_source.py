"""Storage back ends of disk-backed test volumes: HDF5, `.npy` (memory-mapped), zarr v2 directories, multi-page TIFF.

A `VolumeSource` knows the stored array (shape, dtype, which axis -- if any -- is the channel axis) and does exactly one thing:
`read_box(lo, hi)` returns the RAW stored values of a spatial box, in storage axis order and storage dtype, as one contiguous
numpy array.  No transpose, no float conversion, no padding: those run on the device (csrc/volume_kernels.hip), so the bytes that
cross PCIe are the bytes on disk (uint8 EM volumes: a quarter of the fp32 volume the reference's reader hands over).

Formats and the channel-axis rule follow the reference's reader (connectomics/data/io/io.py:33-58 `_detect_format`,
inference/lazy.py:567-596 layout inference): a 3-D array has no channel axis; in a 4-D array the SMALLEST axis is the channel
axis when it is axis 0, 1 or 3, and axis 0 otherwise.
"""
from __future__ import annotations

import itertools
import json
from pathlib import Path
from typing import Optional, Sequence, Tuple

import numpy as np

from ..utils.h5lite import get_h5_backend

_SUFFIX_FORMAT = {"h5": "h5", "hdf5": "h5", "tif": "tiff", "tiff": "tiff", "png": "png", "nii": "nifti", "npy": "npy"}


def detect_format(filename: str) -> str:
    if filename.endswith(".nii.gz"):
        return "nifti"
    fmt = _SUFFIX_FORMAT.get(Path(filename).suffix.lower().lstrip("."))
    if fmt is not None:
        return fmt
    if ".zarr" in filename:
        return "zarr"
    raise ValueError(f"Unrecognizable file format for {filename}. Expected: h5, hdf5, tif, tiff, png, nii, nii.gz, zarr")


# ------------------------------------------------------------------------------------------------ zarr v2 (read only)
class ZarrV2Array:
    """Minimal zarr v2 array reader (directory store, C order, '.' or '/' chunk keys; compressor None / zlib / gzip / bz2 / lzma).
    zarr / numcodecs are not part of the image; blosc- or zstd-compressed stores need them and are refused."""

    _CODECS = {None: lambda b: b, "zlib": None, "gzip": None, "bz2": None, "lzma": None}

    def __init__(self, path: str):
        p = str(path)
        cut = p.index(".zarr") + len(".zarr")
        store, key = Path(p[:cut]), p[cut:].strip("/")
        self.root = store / key if key else store
        if not (self.root / ".zarray").exists():
            arrays = sorted(q.name for q in self.root.iterdir() if (q / ".zarray").exists()) if self.root.is_dir() else []
            if not arrays:
                raise FileNotFoundError(f"{self.root}: no .zarray (zarr v2 array) found")
            self.root = self.root / arrays[0]
        meta = json.loads((self.root / ".zarray").read_text())
        if meta.get("zarr_format") != 2:
            raise ValueError(f"{self.root}: only zarr v2 is supported, got format {meta.get('zarr_format')}")
        if meta.get("order", "C") != "C" or meta.get("filters"):
            raise NotImplementedError(f"{self.root}: zarr arrays with order='F' or filters are not supported")
        self.shape = tuple(int(v) for v in meta["shape"])
        self.chunks = tuple(int(v) for v in meta["chunks"])
        self.dtype = np.dtype(meta["dtype"])
        self.fill = meta.get("fill_value") or 0
        self.sep = meta.get("dimension_separator", ".")
        comp = meta.get("compressor")
        self.codec = None if comp is None else str(comp.get("id"))
        if self.codec not in self._CODECS:
            raise NotImplementedError(f"{self.root}: zarr compressor {self.codec!r} needs numcodecs (not in this image); "
                                      "re-encode with zlib / gzip or use HDF5")

    def _decode(self, raw: bytes) -> bytes:
        if self.codec is None:
            return raw
        import importlib
        return importlib.import_module(self.codec).decompress(raw)

    def _chunk(self, index) -> np.ndarray:
        f = self.root / self.sep.join(str(i) for i in index)
        if not f.exists():
            return np.full(self.chunks, self.fill, dtype=self.dtype)
        return np.frombuffer(self._decode(f.read_bytes()), dtype=self.dtype).reshape(self.chunks)

    def __getitem__(self, key) -> np.ndarray:
        key = key if isinstance(key, tuple) else (key,)
        key = key + (slice(None),) * (len(self.shape) - len(key))
        lo, hi = [], []
        for k, n in zip(key, self.shape):
            start, stop, step = k.indices(n)
            if step != 1:
                raise NotImplementedError("unit-step slices only")
            lo.append(start)
            hi.append(max(start, stop))
        out = np.empty([h - l for l, h in zip(lo, hi)], dtype=self.dtype)
        if out.size == 0:
            return out
        touched = [range(l // c, (h - 1) // c + 1) for l, h, c in zip(lo, hi, self.chunks)]
        for index in itertools.product(*touched):
            chunk = self._chunk(index)
            src, dst = [], []
            for a, ci in enumerate(index):
                origin = ci * self.chunks[a]
                s, e = max(lo[a], origin), min(hi[a], origin + self.chunks[a], self.shape[a])
                src.append(slice(s - origin, e - origin))
                dst.append(slice(s - lo[a], e - lo[a]))
            out[tuple(dst)] = chunk[tuple(src)]
        return out


# ------------------------------------------------------------------------------------------------ the source
class VolumeSource:
    """An opened stored array + its channel axis.  `spatial_axes` are the storage axes of the three spatial dimensions in
    storage order; `read_box` slices them and keeps every channel."""

    def __init__(self, path: str):
        self.path = str(path)
        self.fmt = detect_format(self.path)
        self._owner = None
        if self.fmt == "h5":
            backend = get_h5_backend()
            if backend is None:
                raise RuntimeError(f"{self.path}: HDF5 needs h5py or the in-repo libpytc_h5.so (csrc/host/h5io.c); neither loads")
            self._owner = backend.File(self.path, "r")
            self.array = self._owner[list(self._owner.keys())[0]]
        elif self.fmt == "zarr":
            self.array = ZarrV2Array(self.path)
        elif self.fmt == "npy":
            self.array = np.load(self.path, mmap_mode="r")
        elif self.fmt == "tiff":
            from ..utils.tiffstack import TiffStack
            self._owner = self.array = TiffStack(self.path)          # page-range reads
            if self.array.ndim == 2:
                raise ValueError(f"{self.path}: a single-page TIFF is not a volume")
        else:
            raise ValueError(f"Lazy sliding-window inference does not support format '{self.fmt}' for {self.path}.")
        self.shape = tuple(int(v) for v in self.array.shape)
        self.dtype = np.dtype(self.array.dtype)
        if len(self.shape) == 3:
            self.channel_axis: Optional[int] = None
        elif len(self.shape) == 4:
            smallest = int(np.argmin(self.shape))
            self.channel_axis = smallest if smallest in (0, 1, 3) else 0
        else:
            raise ValueError(f"Unsupported lazy volume rank {len(self.shape)} for shape {self.shape}.")
        self.spatial_axes = tuple(a for a in range(len(self.shape)) if a != self.channel_axis)
        self.channels = 1 if self.channel_axis is None else self.shape[self.channel_axis]
        self.spatial_shape = tuple(self.shape[a] for a in self.spatial_axes)

    def read_box(self, lo: Sequence[int], hi: Sequence[int]) -> np.ndarray:
        """Raw values of spatial box [lo, hi) (indices along `spatial_axes`), all channels, storage order, C-contiguous."""
        key = [slice(None)] * len(self.shape)
        for axis, l, h in zip(self.spatial_axes, lo, hi):
            key[axis] = slice(int(l), int(h))
        box = np.ascontiguousarray(self.array[tuple(key)])
        return box if box.flags.writeable else box.copy()          # a read-only memmap view cannot back a torch tensor

    def close(self) -> None:
        if self._owner is not None and hasattr(self._owner, "close"):
            self._owner.close()
        self._owner = None
        self.array = None


def box_strides(source: VolumeSource, box: np.ndarray, logical_to_stored: Tuple[int, int, int]) -> Tuple[int, int, int, int]:
    """Element strides (channel, z, y, x) of `box` = source.read_box(...), z / y / x being the LOGICAL axes: logical axis a is
    stored along spatial axis `logical_to_stored[a]` (the `val_transpose` permutation) -- the transpose costs an index, no copy."""
    es = box.strides
    item = box.itemsize
    channel = 0 if source.channel_axis is None else es[source.channel_axis] // item
    return (channel,) + tuple(es[source.spatial_axes[logical_to_stored[a]]] // item for a in range(3))


__all__ = ["VolumeSource", "ZarrV2Array", "detect_format", "box_strides"]
