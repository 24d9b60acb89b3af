"""h5lite -- the slice of the h5py API the hot path's readers / writers use, over libpytc_h5.so (csrc/host/h5io.c, a C shim
on the HDF5 C library of this image; h5py itself is not installed).

    with h5lite.File(path, "w") as f:
        d = f.create_dataset("main", data=arr, chunks=(1, 64, 64, 64), compression="gzip")
        d.attrs["crop_pad"] = "[[0, 0], [0, 0], [0, 0]]"
    with h5lite.File(path, "r") as f:
        sub = f["main"][:, 10:20]            # hyperslab read
        meta = dict(f["main"].attrs)

Files are ordinary HDF5 (readable by h5py / h5dump and the reference's decoders: `main` CZYX datasets, variable-length
UTF-8 string attributes, int64 / float64 scalars, h5py's bool enum), and HDF5 volumes written by the reference are
readable here.  Supported indexing: basic slices with step 1 and integers per axis (what artifact.py:141-240,
chunked.py:317-434 and lazy.py:852-904 do).  `get_h5_backend()` returns real h5py when it is importable, else this module,
else None (callers then keep their .npy layout and say so).
"""
from __future__ import annotations

import ctypes as C
import os
import functools
import threading
from pathlib import Path
from typing import Any, Iterator, Optional, Sequence

import numpy as np

LIB_PATH = Path(__file__).resolve().parent.parent / "lib" / "libpytc_h5.so"

_CODES = [("uint8", 0), ("int8", 1), ("uint16", 2), ("int16", 3), ("uint32", 4), ("int32", 5), ("uint64", 6), ("int64", 7),
          ("float16", 8), ("float32", 9), ("float64", 10), ("bool", 11)]
_CODE_OF = {np.dtype(n): c for n, c in _CODES}
_DTYPE_OF = {c: np.dtype(n) for n, c in _CODES}

_lib = None
_lib_err: Optional[str] = None
# ctypes releases the GIL inside a foreign call and libhdf5 is not built thread-safe: every entry below holds this lock
# (the chunk writer thread and a lazy volume reader on the main thread may both be inside HDF5 otherwise)
_LOCK = threading.RLock()


def _locked(fn):
    @functools.wraps(fn)
    def wrapper(*a, **k):
        with _LOCK:
            return fn(*a, **k)
    return wrapper


def _load():
    global _lib, _lib_err
    if _lib is not None or _lib_err is not None:
        return _lib
    try:
        lib = C.CDLL(str(LIB_PATH))
    except OSError as exc:
        _lib_err = f"{LIB_PATH.name} is not available ({exc}); build it with `python -m pytorch_connectomics_amd.csrc.build`"
        return None
    i64, p64 = C.c_int64, C.POINTER(C.c_int64)
    sig = {
        "pytc_h5_last_error": (C.c_char_p, []),
        "pytc_h5_init": (C.c_int, []),
        "pytc_h5_lzf_pack": (i64, [C.c_void_p, i64, C.c_void_p, i64]),
        "pytc_h5_lzf_unpack": (i64, [C.c_void_p, i64, C.c_void_p, i64]),
        "pytc_h5_file_open": (i64, [C.c_char_p, C.c_int]),
        "pytc_h5_file_close": (C.c_int, [i64]),
        "pytc_h5_list": (C.c_int, [i64, C.c_char_p, C.c_int]),
        "pytc_h5_exists": (C.c_int, [i64, C.c_char_p]),
        "pytc_h5_dset_create": (i64, [i64, C.c_char_p, C.c_int, C.c_int, p64, p64, C.c_int]),
        "pytc_h5_dset_open": (i64, [i64, C.c_char_p]),
        "pytc_h5_dset_close": (C.c_int, [i64]),
        "pytc_h5_dset_info": (C.c_int, [i64, C.POINTER(C.c_int), p64, C.POINTER(C.c_int), p64, C.POINTER(C.c_int)]),
        "pytc_h5_dset_filter": (C.c_int, [i64, C.POINTER(C.c_int)]),
        "pytc_h5_dset_write": (C.c_int, [i64, C.c_int, p64, p64, C.c_void_p, C.c_int]),
        "pytc_h5_dset_read": (C.c_int, [i64, C.c_int, p64, p64, C.c_void_p, C.c_int]),
        "pytc_h5_dset_write_parallel": (C.c_int, [i64, C.c_int, p64, p64, C.c_void_p, C.c_int, C.c_int]),
        "pytc_h5_write_parallel_stats": (None, [C.POINTER(C.c_double)]),
        "pytc_h5_deflate_backend": (C.c_int, []),
        "pytc_h5_attr_write": (C.c_int, [i64, C.c_char_p, C.c_int, C.c_char_p, i64, C.c_double]),
        "pytc_h5_attr_count": (C.c_int, [i64]),
        "pytc_h5_attr_name": (C.c_int, [i64, C.c_int, C.c_char_p, C.c_int]),
        "pytc_h5_attr_read": (C.c_int, [i64, C.c_char_p, C.POINTER(C.c_int), C.c_char_p, C.c_int, p64,
                                        C.POINTER(C.c_double)]),
        "pytc_h5_attr_write_array": (C.c_int, [i64, C.c_char_p, C.c_void_p, C.c_int, C.c_int]),
        "pytc_h5_attr_read_array": (C.c_int, [i64, C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int),
                                              C.POINTER(C.c_int)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if lib.pytc_h5_init() != 0:
        _lib_err = "HDF5 library initialisation failed"
        return None
    _lib = lib
    return lib


def available() -> bool:
    return _load() is not None


PARALLEL_WRITE_MIN_BYTES = 4 << 20


_UNSET = object()
_quota_cache = _UNSET
_last_write_was_parallel = False


def write_threads() -> int:
    """Host threads of the parallel deflate writer: PYTC_H5_THREADS, else the cores this process may run on, at most 128 (zlib level 4
    compresses ~15 MB/s per core on near-incompressible fp32 predictions; measured on the 256-core MI355X host: 64 threads 1.77 s for
    the 0.92 GB of a 7 x 320^3 chunk; the single-threaded library path took 63 s)."""
    v = os.environ.get("PYTC_H5_THREADS")
    if v is not None:
        return max(1, int(v))
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    global _quota_cache
    if _quota_cache is _UNSET:
        _quota_cache = _cgroup_cpu_quota()         # a property of the container: read once per process
    q = _quota_cache
    if q is not None:
        n = min(n, max(1, int(2 * q + 0.5)))          # two workers per granted core: the serialized H5Dwrite_chunk calls overlap with deflate
    return max(1, min(128, n))


def _cgroup_cpu_quota() -> Optional[float]:
    """CPUs the container may use at once (cgroup v2 cpu.max, v1 cfs quota / period), or None when unlimited / unknown.  The MI355X boxes
    show 256 cores in the affinity mask under a quota of 16: 128 deflate threads there run no faster than 16 and pay for the oversubscription
    (1.9 against 1.56 s per 0.92 GB chunk, tools/r05_h5_threads.py)."""
    try:
        txt = open("/sys/fs/cgroup/cpu.max").read().split()
        if txt and txt[0] != "max":
            return float(txt[0]) / float(txt[1])
        return None
    except (OSError, ValueError, IndexError):
        pass
    try:
        quota = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return quota / period if quota > 0 and period > 0 else None
    except (OSError, ValueError):
        return None


def last_parallel_write_stats() -> Optional[dict]:
    """Where the last parallel chunk write of this process spent its time (csrc/host/h5io.c): thread-seconds of gather and deflate summed over the
    workers, seconds inside the serialized H5Dwrite_chunk calls, wall seconds, workers; None without the library or when the LAST dataset write
    did not take the parallel path (small, contiguous-layout or declined by the library)."""
    lib = _load()
    if lib is None or not _last_write_was_parallel:          # the last write took the plain hyperslab path: the library's record is stale
        return None
    out = (C.c_double * 5)()
    lib.pytc_h5_write_parallel_stats(out)
    return {"gather_thread_s": out[0], "deflate_thread_s": out[1], "h5_write_serial_s": out[2], "wall_s": out[3], "threads": int(out[4]),
            "deflate": "libdeflate" if lib.pytc_h5_deflate_backend() else "zlib"}


def _need():
    lib = _load()
    if lib is None:
        raise RuntimeError(f"HDF5 support unavailable: {_lib_err}")
    return lib


def _err(lib) -> str:
    return (lib.pytc_h5_last_error() or b"").decode(errors="replace")


def _i64(seq):
    return (C.c_int64 * len(seq))(*[int(v) for v in seq])


def guess_chunk(shape: Sequence[int], typesize: int) -> tuple:
    """h5py's auto-chunk heuristic (chunk between 8 KiB and 1 MiB, halving axes round-robin)."""
    base, cmin, cmax = 16 * 1024, 8 * 1024, 1024 * 1024
    chunks = np.array([max(1, int(s)) for s in shape], dtype="=f8")
    dset_size = float(np.prod(chunks)) * typesize
    target = base * (2 ** np.log10(max(dset_size, 1.0) / (1024.0 * 1024)))
    target = min(max(target, cmin), cmax)
    idx, nd = 0, len(chunks)
    while True:
        cb = float(np.prod(chunks)) * typesize
        if (cb < target or abs(cb - target) / target < 0.5) and cb < cmax:
            break
        if np.prod(chunks) == 1:
            break
        chunks[idx % nd] = np.ceil(chunks[idx % nd] / 2.0)
        idx += 1
    return tuple(int(x) for x in chunks)


class AttributeManager:
    """dict-like view of the scalar attributes of a dataset."""

    def __init__(self, owner: "Dataset"):
        self._o = owner

    @_locked
    def __setitem__(self, key: str, value: Any) -> None:
        lib = _need()
        k = key.encode()
        if isinstance(value, (bool, np.bool_)):
            rc = lib.pytc_h5_attr_write(self._o._id, k, 3, None, int(value), 0.0)
        elif isinstance(value, (int, np.integer)):
            rc = lib.pytc_h5_attr_write(self._o._id, k, 1, None, int(value), 0.0)
        elif isinstance(value, (float, np.floating)):
            rc = lib.pytc_h5_attr_write(self._o._id, k, 2, None, 0, float(value))
        elif isinstance(value, (str, bytes)):
            sval = value if isinstance(value, bytes) else value.encode("utf-8")
            rc = lib.pytc_h5_attr_write(self._o._id, k, 0, sval, 0, 0.0)
        elif isinstance(value, (list, tuple, np.ndarray)) and np.asarray(value).dtype.kind in "iuf" and np.asarray(value).ndim == 1:
            arr = np.asarray(value)
            # integers travel as 8-byte integers (never through doubles: uint64 ids / offsets above 2^53 stay exact)
            is_int = 2 if (arr.dtype.kind == "u" and arr.dtype.itemsize == 8) else int(arr.dtype.kind in "iu")
            data = np.ascontiguousarray(arr, dtype=(np.float64, np.int64, np.uint64)[is_int])
            rc = lib.pytc_h5_attr_write_array(self._o._id, k, data.ctypes.data_as(C.c_void_p), int(arr.size), is_int)
        else:
            raise TypeError(f"h5lite attributes hold str / int / float / bool scalars or 1-D numeric arrays, got "
                            f"{type(value).__name__} for {key!r}")
        if rc != 0:
            raise OSError(_err(lib))

    @_locked
    def keys(self):
        lib = _need()
        n = lib.pytc_h5_attr_count(self._o._id)
        out = []
        for i in range(max(n, 0)):
            buf = C.create_string_buffer(512)
            if lib.pytc_h5_attr_name(self._o._id, i, buf, 512) == 0:
                out.append(buf.value.decode())
        return out

    @_locked
    def __getitem__(self, key: str) -> Any:
        lib = _need()
        kind, ival, dval = C.c_int(0), C.c_int64(0), C.c_double(0.0)
        sbuf = C.create_string_buffer(1 << 16)
        rc = lib.pytc_h5_attr_read(self._o._id, key.encode(), C.byref(kind), sbuf, len(sbuf), C.byref(ival), C.byref(dval))
        if rc == 3:     # array-valued attribute (e.g. `resolution`): numeric arrays come back as numpy arrays
            n, is_int = C.c_int(0), C.c_int(0)
            rc = lib.pytc_h5_attr_read_array(self._o._id, key.encode(), None, 0, C.byref(n), C.byref(is_int))      # size + type
            if rc == 2:
                raise TypeError(f"attribute {key!r} is an array of a type h5lite does not map (numeric arrays only: "
                                "string / compound / reference arrays are not supported)")
            if rc != 0:
                raise OSError(_err(lib))
            out = np.zeros(max(n.value, 0), dtype=(np.float64, np.int64, np.uint64)[is_int.value])
            if out.size and lib.pytc_h5_attr_read_array(self._o._id, key.encode(), out.ctypes.data_as(C.c_void_p), int(out.size),
                                                        C.byref(n), C.byref(is_int)) != 0:
                raise OSError(_err(lib))
            return out
        if rc != 0:
            raise KeyError(key)
        return {0: lambda: sbuf.value.decode("utf-8", errors="replace"), 1: lambda: int(ival.value),
                2: lambda: float(dval.value), 3: lambda: bool(ival.value)}[kind.value]()

    @_locked
    def __contains__(self, key) -> bool:
        return key in self.keys()

    def __iter__(self) -> Iterator[str]:
        return iter(self.keys())

    def __len__(self) -> int:
        return len(self.keys())

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default


class Dataset:
    @_locked
    def __init__(self, ds_id: int, name: str):
        self._id = ds_id
        self.name = name
        lib = _need()
        nd, dt, hc = C.c_int(0), C.c_int(0), C.c_int(0)
        dims, chunks = (C.c_int64 * 8)(), (C.c_int64 * 8)()
        if lib.pytc_h5_dset_info(ds_id, C.byref(nd), dims, C.byref(dt), chunks, C.byref(hc)) != 0:
            raise OSError(_err(lib))
        if dt.value not in _DTYPE_OF:
            raise TypeError(f"dataset {name!r} has an HDF5 type h5lite does not map (code {dt.value})")
        self.shape = tuple(int(dims[i]) for i in range(nd.value))
        self.dtype = _DTYPE_OF[dt.value]
        self.chunks = tuple(int(chunks[i]) for i in range(nd.value)) if hc.value else None
        lvl = C.c_int(-1)
        fid = lib.pytc_h5_dset_filter(ds_id, C.byref(lvl))
        self.compression = {0: None, 1: "gzip", 4: "szip", 32000: "lzf"}.get(int(fid))       # h5py's names
        self.compression_opts = int(lvl.value) if fid == 1 else None
        self.attrs = AttributeManager(self)

    ndim = property(lambda self: len(self.shape))
    size = property(lambda self: int(np.prod(self.shape)) if self.shape else 1)

    def __len__(self):
        return self.shape[0]

    def _select(self, key):
        if not isinstance(key, tuple):
            key = (key,)
        if any(k is Ellipsis for k in key):
            i = next(j for j, k in enumerate(key) if k is Ellipsis)
            key = key[:i] + (slice(None),) * (len(self.shape) - (len(key) - 1)) + key[i + 1:]
        key = key + (slice(None),) * (len(self.shape) - len(key))
        if len(key) != len(self.shape):
            raise IndexError(f"too many indices for a {len(self.shape)}-D dataset")
        start, count, squeeze = [], [], []
        for ax, (k, n) in enumerate(zip(key, self.shape)):
            if isinstance(k, (int, np.integer)):
                k = int(k) + (n if k < 0 else 0)
                if not 0 <= k < n:
                    raise IndexError(f"index {k} out of range for axis {ax} of size {n}")
                start.append(k); count.append(1); squeeze.append(ax)
            elif isinstance(k, slice):
                s, e, st = k.indices(n)
                if st != 1:
                    raise NotImplementedError("h5lite supports unit-step slices only")
                start.append(s); count.append(max(0, e - s))
            else:
                raise TypeError(f"unsupported index {k!r} (h5lite: ints and unit-step slices)")
        return start, count, squeeze

    @_locked
    def __getitem__(self, key) -> np.ndarray:
        lib = _need()
        start, count, squeeze = self._select(key)
        out = np.empty(count, dtype=self.dtype)
        if out.size:
            if lib.pytc_h5_dset_read(self._id, len(start), _i64(start), _i64(count), out.ctypes.data_as(C.c_void_p),
                                     _CODE_OF[self.dtype]) != 0:
                raise OSError(_err(lib))
        return out.reshape([c for ax, c in enumerate(count) if ax not in squeeze]) if squeeze else out

    @_locked
    def __setitem__(self, key, value) -> None:
        lib = _need()
        start, count, squeeze = self._select(key)
        tshape = [c for ax, c in enumerate(count) if ax not in squeeze]
        arr = np.ascontiguousarray(np.broadcast_to(np.asarray(value, dtype=self.dtype), tshape))
        if arr.size:
            # chunk-aligned writes into a chunked (gzip or unfiltered) dataset: N host threads deflate whole HDF5 chunks and the
            # compressed bytes go in through H5Dwrite_chunk (csrc/host/h5io.c) -- H5Dwrite would run zlib on ONE thread (63 s for a
            # 0.9 GB prediction chunk, 40x the GPU time that produced it).  rc 2 = not applicable here: the plain hyperslab write.
            global _last_write_was_parallel
            _last_write_was_parallel = False
            # (size and layout first: write_threads() reads the environment and, once per process, the cgroup files)
            nthreads = write_threads() if (self.chunks is not None and arr.nbytes >= PARALLEL_WRITE_MIN_BYTES) else 1
            if nthreads > 1:
                rc = lib.pytc_h5_dset_write_parallel(self._id, len(start), _i64(start), _i64(count), arr.ctypes.data_as(C.c_void_p),
                                                     _CODE_OF[self.dtype], nthreads)
                if rc == 0:
                    _last_write_was_parallel = True
                    return
                if rc == 1:
                    raise OSError(_err(lib))
            if lib.pytc_h5_dset_write(self._id, len(start), _i64(start), _i64(count), arr.ctypes.data_as(C.c_void_p),
                                      _CODE_OF[self.dtype]) != 0:
                raise OSError(_err(lib))

    def __array__(self, dtype=None, copy=None):
        a = self[...]
        return a if dtype is None else a.astype(dtype)

    @_locked
    def _close(self):
        if self._id is not None:
            _need().pytc_h5_dset_close(self._id)
            self._id = None


class File:
    """h5py.File subset: modes 'r', 'r+', 'a', 'w'; context manager; datasets in the root group."""

    @_locked
    def __init__(self, path, mode: str = "r"):
        lib = _need()
        p = Path(path)
        if mode == "a":
            mode = "r+" if p.exists() else "w"
        code = {"r": 0, "r+": 1, "w": 2}.get(mode)
        if code is None:
            raise ValueError(f"unsupported h5lite file mode {mode!r}")
        self._id = lib.pytc_h5_file_open(str(p).encode(), code)
        if self._id < 0:
            self._id = None
            if code == 0 and not p.exists():
                raise FileNotFoundError(str(p))
            raise OSError(_err(lib))
        self.filename = str(p)
        self.mode = mode
        self._open: list[Dataset] = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    @_locked
    def close(self):
        if self._id is not None:
            for d in self._open:
                d._close()
            self._open = []
            _need().pytc_h5_file_close(self._id)
            self._id = None

    def __del__(self):
        try:
            self.close()
        except Exception:    # noqa: BLE001 - interpreter shutdown
            pass

    @_locked
    def keys(self):
        lib = _need()
        buf = C.create_string_buffer(1 << 16)
        lib.pytc_h5_list(self._id, buf, len(buf))
        return [s for s in buf.value.decode().split("\n") if s]

    def __iter__(self):
        return iter(self.keys())

    @_locked
    def __contains__(self, name) -> bool:
        return bool(_need().pytc_h5_exists(self._id, str(name).encode()))

    @_locked
    def __getitem__(self, name: str) -> Dataset:
        lib = _need()
        ds = lib.pytc_h5_dset_open(self._id, name.encode())
        if ds < 0:
            raise KeyError(f"Unable to open object '{name}' in {self.filename}")
        d = Dataset(ds, name)
        self._open.append(d)
        return d

    @_locked
    def create_dataset(self, name: str, shape=None, dtype=None, data=None, chunks=None, compression=None,
                       compression_opts=None) -> Dataset:
        lib = _need()
        arr = None
        if data is not None:
            arr = np.ascontiguousarray(data if dtype is None else np.asarray(data, dtype=dtype))
            shape, dtype = arr.shape, arr.dtype
        if shape is None or dtype is None:
            raise TypeError("create_dataset needs data, or shape and dtype")
        dt = np.dtype(dtype)
        if dt not in _CODE_OF:
            raise TypeError(f"h5lite does not map dtype {dt}")
        shape = tuple(int(v) for v in shape)
        gz = -1
        if compression is not None:
            if isinstance(compression, str) and compression.lower() == "lzf":
                gz = -2                                  # h5py's fast filter (id 32000), registered by the shim (csrc/host/h5io.c)
            elif compression not in ("gzip", "GZIP") and not isinstance(compression, int):
                raise NotImplementedError(f"h5lite writes gzip (deflate) or lzf compression, got {compression!r}")
            else:
                gz = int(compression) if isinstance(compression, int) else int(4 if compression_opts is None else compression_opts)
            if chunks is None or chunks is True:
                chunks = guess_chunk(shape, dt.itemsize)
        if chunks is True:
            chunks = guess_chunk(shape, dt.itemsize)
        if chunks is not None:
            chunks = tuple(min(int(c), max(1, s)) for c, s in zip(chunks, shape))
            if len(chunks) != len(shape) or any(c <= 0 for c in chunks):
                raise ValueError(f"chunks {chunks} do not fit shape {shape}")
        ds = lib.pytc_h5_dset_create(self._id, name.encode(), _CODE_OF[dt], len(shape), _i64(shape),
                                     _i64(chunks) if chunks is not None else None, gz)
        if ds < 0:
            raise OSError(_err(lib))
        d = Dataset(ds, name)
        self._open.append(d)
        if arr is not None and arr.size:
            d[...] = arr
        return d


def get_h5_backend():
    """h5py if importable, else this module when libpytc_h5.so loads, else None."""
    try:
        import h5py  # type: ignore
        return h5py
    except ImportError:
        pass
    import sys
    return sys.modules[__name__] if available() else None


__all__ = ["File", "Dataset", "AttributeManager", "available", "get_h5_backend", "guess_chunk"]
