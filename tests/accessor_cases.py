"""Shared by tests/test_host_lazy_accessor.py and tests/test_gpu_lazy_accessor.py: the accessor configurations of
tests/golden/lazy_accessor.npz (make_golden.py --accessor) and the writer of the same volume in every storage format."""
import itertools
import json
import zlib
from pathlib import Path

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


# ---- tile-grid sources (tests/golden/lazy_accessor_tiles.npz, make_golden.py --accessor_tiles)
TILE_CASES = {
    # name: (layout, accessor kwargs, [(location, size) ...], outer pad mode, outer pad value)
    "tiles_json_plain": ("json", dict(kind="image"), [((0, 0, 0), (4, 12, 14)), ((1, 5, 7), (3, 16, 12)), ((-1, 14, 12), (4, 12, 10))], "reflect", 0.0),
    "tiles_json_transpose_pad_div": ("json", dict(kind="image", transpose_axes=(1, 2, 0), context_pad=((1, 2), (2, 0), (1, 1)),
                                                  context_pad_mode="reflect", normalize_mode="divide-255"),
                                     [((0, 0, 0), (10, 10, 4)), ((12, 9, 2), (12, 12, 4))], "constant", 0.5),
    "tiles_dir_inferred": ("dir", dict(kind="image", scale_factors=(1.0, 0.5, 1.5)), [((0, 0, 0), (4, 8, 16)), ((2, 4, 12), (2, 8, 18))], "replicate", 0.0),
    "tiles_json_ratio2_mask": ("json_ratio2", dict(kind="mask", binarize=True, threshold=120.0), [((0, 10, 10), (4, 30, 24)), ((2, 20, 0), (2, 28, 40))], "constant", 0.0),
    "tiles_json_rgb_label": ("json_rgb", dict(kind="label"), [((0, 0, 0), (3, 16, 20)), ((1, 7, 9), (3, 12, 10))], "constant", 0.0),
}


def write_tile_layout(root, layout, tiles, rgb_tiles):
    """tiles (Z, R, C, h, w) uint8 / rgb_tiles (Z, R, C, h, w, 3) uint8 -> a tile source under `root`; returns its path.
    json: sections sec<z>/, tile index origin (1, 2), tile (z=2, r=2, c=1) missing (stays background); dir: numeric section
    directories the metadata is inferred from; json_ratio2: tiles zoomed x2 at read time; json_rgb: VAST RGB label tiles."""
    from PIL import Image
    root = Path(root)
    Z, R, Cc, h, w = tiles.shape
    if layout == "dir":
        for z in range(Z):
            (root / "stack" / str(z)).mkdir(parents=True, exist_ok=True)
            for r in range(R):
                for c in range(Cc):
                    Image.fromarray(tiles[z, r, c]).save(root / "stack" / str(z) / f"{r + 3}_{c + 1}.png")
        return str(root / "stack")
    rgb = layout == "json_rgb"
    scale = 2 if layout == "json_ratio2" else 1
    name = f"tiles_{layout}"
    for z in range(Z):
        (root / name / f"sec{z}").mkdir(parents=True, exist_ok=True)
        for r in range(R):
            for c in range(Cc):
                if layout == "json" and (z, r, c) == (2, 2, 1):
                    continue
                Image.fromarray(rgb_tiles[z, r, c] if rgb else tiles[z, r, c]).save(root / name / f"sec{z}" / f"{r + 1}_{c + 2}.png")
    meta = {"image": [f"{name}/sec{z}/{{row}}_{{column}}.png" for z in range(Z)], "height": R * h * scale, "width": Cc * w * scale,
            "tile_size": [h * scale, w * scale], "tile_st": [1, 2]}
    if scale != 1:
        meta["tile_ratio"] = float(scale)
    if rgb:
        meta["dtype"] = "uint32"
    (root / f"{name}.json").write_text(json.dumps(meta))
    return str(root / f"{name}.json")
