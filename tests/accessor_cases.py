"""Shared by tests/test_host_lazy_accessor.py and tests/test_gpu_lazy_accessor.py: the accessor configurations of
tests/golden/lazy_accessor.npz (make_golden.py --accessor) and the writer of the same volume in every storage format."""
import itertools
import json
import zlib

import numpy as np

from pytorch_connectomics_amd.utils import h5lite

CASES = {
    "plain": ("zyx", dict(kind="image"), [((0, 0, 0), (6, 7, 8)), ((-2, 3, 12), (6, 8, 10)), ((8, 10, 14), (8, 8, 8))], "reflect", 0.0),
    "transpose_pad_reflect_div": ("zyx", dict(kind="image", transpose_axes=(2, 0, 1), context_pad=((2, 1), (0, 3), (2, 2)),
                                              context_pad_mode="reflect", normalize_mode="divide-255"),
                                  [((0, 0, 0), (8, 8, 8)), ((-3, -1, 5), (10, 9, 12)), ((15, 6, 10), (8, 8, 8))], "constant", 0.25),
    "resize_bilinear_znorm": ("czyx", dict(kind="image", scale_factors=(1.5, 0.75, 1.25), context_pad=((1, 1), (1, 1), (1, 1)),
                                           context_pad_mode="constant", normalize_mode="normal", clip_percentile_low=0.05,
                                           clip_percentile_high=0.95),
                              [((0, 0, 0), (8, 6, 10)), ((5, 2, 8), (8, 8, 8)), ((-1, -2, 14), (6, 6, 10))], "replicate", 0.0),
    "channel_last_edge_01": ("zyxc", dict(kind="image", context_pad=((0, 2), (2, 0), (1, 1)), context_pad_mode="edge",
                                          normalize_mode="0-1"),
                             [((0, 0, 0), (6, 6, 6)), ((6, 8, 10), (6, 8, 8))], "reflect", 0.0),
    "mask_nearest_binarize": ("zyx", dict(kind="mask", scale_factors=(0.5, 2.0, 1.0), binarize=True, threshold=100.0),
                              [((0, 0, 0), (4, 10, 8)), ((2, 20, 10), (4, 8, 8))], "constant", 0.0),
}


def _write_sources(g, tmp_path, key):
    """the same volume as .h5, .npy and a zlib-compressed zarr v2 directory with ragged edge chunks"""
    vol = g[f"vol_{key}"]
    paths = {"npy": str(tmp_path / f"{key}.npy")}
    np.save(paths["npy"], vol)
    be = h5lite.get_h5_backend()
    if be is not None:
        paths["h5"] = str(tmp_path / f"{key}.h5")
        with be.File(paths["h5"], "w") as fh:
            fh.create_dataset("main", data=vol, compression="gzip")
    zdir = tmp_path / f"{key}.zarr"
    zdir.mkdir()
    chunks = tuple(max(1, (s + 2) // 3) for s in vol.shape)
    (zdir / ".zarray").write_text(json.dumps({"zarr_format": 2, "shape": list(vol.shape), "chunks": list(chunks),
                                               "dtype": vol.dtype.str, "compressor": {"id": "zlib", "level": 1},
                                               "fill_value": 0, "order": "C", "filters": None}))
    for idx in itertools.product(*[range((s + c - 1) // c) for s, c in zip(vol.shape, chunks)]):
        block = np.zeros(chunks, vol.dtype)
        sl = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, vol.shape))
        block[tuple(slice(0, s.stop - s.start) for s in sl)] = vol[sl]
        (zdir / ".".join(map(str, idx))).write_bytes(zlib.compress(block.tobytes(), 1))
    paths["zarr"] = str(zdir)
    return paths
