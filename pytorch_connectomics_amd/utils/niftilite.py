"""Single-file NIfTI-1 volumes (.nii / .nii.gz) without nibabel -- the reference reads them through `nibabel`
(connectomics/data/io/io.py:267-306: `np.asarray(nib.load(f).dataobj)`, then (X, Y, Z) -> (Z, Y, X) and (X, Y, Z, C) -> (C, Z, Y, X));
nibabel is not part of this image.  The format is a 348-byte header followed by the voxels in Fortran order:

  offset  40  int16 dim[8]      dim[0] = rank, dim[1..] = sizes (x fastest)
          70  int16 datatype    2 u8 | 4 i16 | 8 i32 | 16 f32 | 64 f64 | 256 i8 | 512 u16 | 768 u32 | 1024 i64 | 1280 u64
         108  f32   vox_offset  byte offset of the data (352 for a plain single file)
         112  f32   scl_slope, 116 f32 scl_inter   stored = raw * slope + inter when slope is finite and non-zero
         344  char  magic "n+1\\0"
  sizeof_hdr (int32 at 0) == 348 tells the byte order.

Like `dataobj`, scaling is applied only when the header asks for it ((slope, inter) not (1, 0) / (0, *) / NaN) and then yields float64."""
from __future__ import annotations

import gzip
import struct

import numpy as np

__all__ = ["read_nifti", "nifti_shape", "write_nifti"]

_DTYPES = {2: "u1", 4: "i2", 8: "i4", 16: "f4", 64: "f8", 256: "i1", 512: "u2", 768: "u4", 1024: "i8", 1280: "u8"}
_CODES = {np.dtype(v).str[1:]: k for k, v in _DTYPES.items()}


def _open(path: str):
    return gzip.open(path, "rb") if str(path).endswith(".gz") else open(path, "rb")


def _header(raw: bytes, path: str):
    if len(raw) < 348:
        raise ValueError(f"{path}: shorter than a NIfTI-1 header")
    for end in ("<", ">"):
        if struct.unpack(end + "i", raw[:4])[0] == 348:
            break
    else:
        raise ValueError(f"{path}: not a NIfTI-1 file (sizeof_hdr != 348)")
    if raw[344:347] not in (b"n+1",):
        raise ValueError(f"{path}: only single-file NIfTI-1 ('n+1') is supported, magic={raw[344:348]!r}")
    dim = struct.unpack(end + "8h", raw[40:56])
    rank = int(dim[0])
    if not 1 <= rank <= 7:
        raise ValueError(f"{path}: bad NIfTI rank {rank}")
    shape = tuple(int(v) for v in dim[1:1 + rank])
    while len(shape) > 3 and shape[-1] == 1:            # trailing singleton time / vector axes
        shape = shape[:-1]
    code = struct.unpack(end + "h", raw[70:72])[0]
    if code not in _DTYPES:
        raise ValueError(f"{path}: unsupported NIfTI datatype code {code}")
    vox_offset = int(struct.unpack(end + "f", raw[108:112])[0])
    slope, inter = struct.unpack(end + "2f", raw[112:120])
    return end, shape, np.dtype(end + _DTYPES[code]), max(vox_offset, 352), float(slope), float(inter)


def _to_reference_axes(data: np.ndarray) -> np.ndarray:
    if data.ndim == 3:
        return data.transpose(2, 1, 0)                  # (X, Y, Z) -> (Z, Y, X) = (D, H, W)
    if data.ndim == 4:
        return data.transpose(3, 2, 1, 0)               # (X, Y, Z, C) -> (C, D, H, W)
    return data


def read_nifti(path: str) -> np.ndarray:
    """-> (D, H, W) or (C, D, H, W), like the reference's `_read_nifti`."""
    with _open(path) as fh:
        raw = fh.read()
    _end, shape, dtype, off, slope, inter = _header(raw, path)
    n = int(np.prod(shape))
    data = np.frombuffer(raw, dtype=dtype, count=n, offset=off).reshape(shape, order="F")
    if np.isfinite(slope) and slope != 0.0 and not (slope == 1.0 and inter == 0.0):
        data = data.astype(np.float64) * slope + inter
    else:
        data = data.astype(dtype.newbyteorder("="), copy=False)
    return np.ascontiguousarray(_to_reference_axes(data))


def nifti_shape(path: str) -> tuple:
    """Shape in the reference's convention without reading the voxels (io.py:297-306)."""
    with _open(path) as fh:
        raw = fh.read(352)
    shape = _header(raw, path)[1]
    return tuple(reversed(shape)) if len(shape) in (3, 4) else shape


def write_nifti(path: str, volume: np.ndarray) -> None:
    """(D, H, W) / (C, D, H, W) -> single-file NIfTI-1 with an identity affine (the reference's `_write_nifti`)."""
    vol = np.asarray(volume)
    key = vol.dtype.newbyteorder("=").str[1:]
    if key not in _CODES:
        raise ValueError(f"NIfTI cannot store dtype {vol.dtype}")
    data = vol.transpose(2, 1, 0) if vol.ndim == 3 else (vol.transpose(3, 2, 1, 0) if vol.ndim == 4 else vol)
    hdr = bytearray(348)
    struct.pack_into("<i", hdr, 0, 348)
    dim = [data.ndim, *data.shape] + [1] * (7 - data.ndim)
    struct.pack_into("<8h", hdr, 40, *dim)
    struct.pack_into("<h", hdr, 70, _CODES[key])
    struct.pack_into("<h", hdr, 72, vol.dtype.itemsize * 8)
    struct.pack_into("<8f", hdr, 76, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0)          # pixdim
    struct.pack_into("<f", hdr, 108, 352.0)
    struct.pack_into("<2f", hdr, 112, 1.0, 0.0)
    struct.pack_into("<h", hdr, 254, 2)                                                 # sform_code: aligned
    struct.pack_into("<4f", hdr, 280, 1.0, 0.0, 0.0, 0.0)
    struct.pack_into("<4f", hdr, 296, 0.0, 1.0, 0.0, 0.0)
    struct.pack_into("<4f", hdr, 312, 0.0, 0.0, 1.0, 0.0)
    hdr[344:348] = b"n+1\x00"
    payload = bytes(hdr) + b"\x00" * 4 + np.asarray(data, dtype=vol.dtype.newbyteorder("<")).tobytes(order="F")
    with (gzip.open(path, "wb") if str(path).endswith(".gz") else open(path, "wb")) as fh:
        fh.write(payload)
