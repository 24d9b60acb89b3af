"""Read-only multi-page TIFF stack with array-style region reads.

The reference reads TIFF volumes through `tifffile` (connectomics/data/io/io.py:199-237 whole volume and shape,
connectomics/inference/lazy.py:639-676 page-range reads of the lazy accessor).  tifffile is not part of this image;
Pillow is, and it decodes what the tutorial datasets ship (Lucchi++ / SNEMI3D: 8- and 16-bit grayscale page stacks,
uncompressed / LZW / deflate).  Pages are decoded one at a time, so a window read touches only its own z range -- the
property the lazy sliding-window path needs -- and the decoded rows are cropped before they are stacked.

Shape convention follows the reference: a single page is (Y, X); a stack of grayscale pages (Z, Y, X); multi-sample
pages (RGB) are (Z, Y, X, S), which the accessor's layout inference treats as channel-last."""
from __future__ import annotations

import threading

import numpy as np

__all__ = ["TiffStack", "read_tiff_volume", "tiff_volume_shape"]


def _pil():
    try:
        from PIL import Image
    except ImportError as exc:                                                       # pragma: no cover
        raise ImportError("TIFF volumes need Pillow (tifffile is not in this image)") from exc
    Image.MAX_IMAGE_PIXELS = None                # EM sections are routinely > 89 Mpx; these are local trusted files
    return Image


class TiffStack:
    """`stack[z0:z1, y0:y1, x0:x1]` -> ndarray; `.shape`, `.dtype`, `.ndim` like the h5 / zarr datasets."""

    def __init__(self, path: str):
        self.path = str(path)
        self._img = _pil().open(self.path)
        if getattr(self._img, "format", None) != "TIFF":
            raise ValueError(f"{self.path}: not a TIFF file (format={self._img.format!r})")
        self._lock = threading.Lock()            # seek + decode is stateful; tile_read_workers may share one accessor
        n = int(getattr(self._img, "n_frames", 1))
        first = self._page(0)
        self._page_shape = first.shape
        self.dtype = first.dtype
        self.shape = tuple(first.shape) if n == 1 else (n, *first.shape)
        self.ndim = len(self.shape)

    def _page(self, z: int) -> np.ndarray:
        self._img.seek(int(z))
        return np.asarray(self._img)             # 1-bit pages decode to bool, as they do through tifffile

    def close(self) -> None:
        if self._img is not None:
            self._img.close()
        self._img = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __len__(self) -> int:
        return self.shape[0]

    def __getitem__(self, key) -> np.ndarray:
        if not isinstance(key, tuple):
            key = (key,)
        if any(k is Ellipsis for k in key):
            i = key.index(Ellipsis)
            key = key[:i] + (slice(None),) * (self.ndim - len(key) + 1) + key[i + 1:]
        key = key + (slice(None),) * (self.ndim - len(key))
        if len(key) != self.ndim:
            raise IndexError(f"too many indices for a TIFF stack of shape {self.shape}")
        if len(self.shape) == len(self._page_shape):                                   # single page
            with self._lock:
                return self._page(0)[key]
        zk, rest = key[0], key[1:]
        if isinstance(zk, (int, np.integer)):
            z = int(zk) + (self.shape[0] if zk < 0 else 0)
            if not 0 <= z < self.shape[0]:
                raise IndexError(f"page {zk} out of range for {self.shape[0]} pages")
            with self._lock:
                return self._page(z)[rest]
        zs = range(*zk.indices(self.shape[0]))
        with self._lock:
            planes = [self._page(z)[rest] for z in zs]
        if not planes:
            probe = np.empty(self._page_shape, self.dtype)[rest]
            return np.empty((0, *probe.shape), self.dtype)
        return np.stack(planes, axis=0)

    def __array__(self, dtype=None, copy=None):
        a = self[(slice(None),) * self.ndim]
        return a if dtype is None else a.astype(dtype, copy=False)


def read_tiff_volume(path: str) -> np.ndarray:
    """The whole stack (io.py:199-216)."""
    with TiffStack(path) as st:
        return np.asarray(st)


def tiff_volume_shape(path: str) -> tuple:
    """Shape without decoding more than the first page (io.py:219-237)."""
    with TiffStack(path) as st:
        return st.shape
