"""CPU parity of the disk-backed volume reader (inference/lazy_accessor.py) against tests/golden/lazy_accessor.npz -- outputs of
the REFERENCE's LazyVolumeAccessor (connectomics/inference/lazy.py:456-917) reading HDF5 files through the same libhdf5
(make_golden.py --accessor) -- plus the other sources (.npy memmap, zarr v2 directory) and the region prefetcher."""
import itertools
import json
import zlib
from types import SimpleNamespace as NS

import numpy as np
import pytest

from pytorch_connectomics_amd.inference.lazy_accessor import (LazyVolumeAccessor, RegionPrefetcher, ZarrV2Array, build_accessor,
                                                              get_padsize, smart_normalize)
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


@pytest.mark.parametrize("name", list(CASES))
def test_accessor_matches_reference_fixture(name, golden_dir, tmp_path):
    g = np.load(golden_dir / "lazy_accessor.npz")
    vk, kw, reads, outer_mode, outer_val = CASES[name]
    for fmt, path in _write_sources(g, tmp_path, vk).items():
        with LazyVolumeAccessor(path, **kw) as acc:
            shapes = [acc.channel_count, *acc.raw_spatial_shape, *acc.logical_spatial_shape, *acc.transformed_spatial_shape,
                      *acc.padded_spatial_shape]
            assert shapes == list(g[f"{name}__shapes"]), fmt
            for i, (loc, size) in enumerate(reads):
                got = acc.read_patch(loc, size, outer_pad_mode=outer_mode, outer_pad_value=outer_val)
                want = g[f"{name}__patch{i}"]
                assert got.shape == want.shape and got.dtype == np.float32
                np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6, err_msg=f"{name} {fmt} patch {i}")
            np.testing.assert_allclose(acc.load_full(), g[f"{name}__full"], rtol=1e-6, atol=1e-6)
            assert acc.shape == (acc.channel_count, *acc.padded_spatial_shape)


def test_read_region_is_the_union_of_its_patches_and_prefetcher(golden_dir, tmp_path):
    """Pointwise pipelines: a region read once equals what per-window read_patch calls return inside it (the device engine
    gathers windows from the region); per-patch statistics modes refuse the region path."""
    g = np.load(golden_dir / "lazy_accessor.npz")
    path = _write_sources(g, tmp_path, "zyx")["npy"]
    kw = dict(kind="image", transpose_axes=(2, 0, 1), context_pad=((2, 1), (0, 3), (2, 2)), context_pad_mode="reflect",
              normalize_mode="divide-255")
    with LazyVolumeAccessor(path, **kw) as acc:
        assert not acc.needs_per_patch_host_path
        reg = acc.read_region((3, 2, 4), (15, 12, 16))
        assert reg.shape == (1, 12, 10, 12)
        p = acc.read_patch((5, 4, 6), (6, 6, 8), outer_pad_mode="constant", outer_pad_value=0.0)
        np.testing.assert_array_equal(reg[:, 2:8, 2:8, 2:10], p)
        clipped = acc.read_region((-4, 0, 0), (4, 100, 5))                 # clipped to the padded volume
        assert clipped.shape == (1, 4, acc.padded_spatial_shape[1], 5)
        regions = [((0, 0, 0), (8, 8, 8)), ((8, 0, 0), (16, 8, 8)), ((4, 4, 4), (12, 12, 12))]
        pf = RegionPrefetcher(acc, regions, pin=False)
        for r in regions:
            rr, t = pf.get()
            assert rr == r
            np.testing.assert_array_equal(t.numpy(), acc.read_region(*r))
        with pytest.raises(StopIteration):
            pf.get()
    with LazyVolumeAccessor(path, kind="image", normalize_mode="normal") as acc:
        assert acc.needs_per_patch_host_path
        with pytest.raises(RuntimeError, match="read_patch"):
            acc.read_region((0, 0, 0), (4, 4, 4))


def test_build_accessor_from_config_and_helpers(tmp_path):
    vol = (np.random.default_rng(0).random((9, 10, 11)) * 255).astype(np.uint8)
    np.save(tmp_path / "v.npy", vol)
    cfg = NS(system=NS(num_workers=2),
             data=NS(dataloader=NS(patch_size=[4, 4, 4]),
                     data_transform=NS(val_transpose=[0, 2, 1], pad_size=[2, 0, 1], pad_mode="reflect", resize=None),
                     image_transform=NS(normalize="divide-255", clip_percentile_low=0.0, clip_percentile_high=1.0, resize=None),
                     mask_transform=None))
    with build_accessor(cfg, str(tmp_path / "v.npy"), kind="image") as acc:
        assert acc.transpose_axes == (0, 2, 1) and acc.context_pad == ((2, 2), (0, 0), (1, 1))
        assert acc.padded_spatial_shape == (13, 11, 12) and acc.normalize_mode == "divide-255"
        p = acc.read_patch((2, 0, 1), (4, 4, 4), outer_pad_mode="constant", outer_pad_value=0.0)
        np.testing.assert_allclose(p[0], vol.transpose(0, 2, 1)[0:4, 0:4, 0:4].astype(np.float32) / 255.0, rtol=1e-6)
    with build_accessor(cfg, str(tmp_path / "v.npy"), kind="label") as acc:           # labels: no context pad, no normalisation
        assert acc.context_pad == ((0, 0), (0, 0), (0, 0)) and acc.normalize_mode == "none"
    cfg.data.data_transform.resize = [8, 8, 8]
    with build_accessor(cfg, str(tmp_path / "v.npy"), kind="image") as acc:
        assert acc.scale_factors == (2.0, 2.0, 2.0)
    cfg.data.dataloader.patch_size = None
    with pytest.raises(ValueError, match="patch_size"):
        build_accessor(cfg, str(tmp_path / "v.npy"), kind="image")
    assert get_padsize(3) == ((3, 3),) * 3 and get_padsize([1, 2, 3, 4, 5, 6]) == ((1, 2), (3, 4), (5, 6))
    with pytest.raises(ValueError):
        get_padsize([1, 2])
    x = np.linspace(-1, 3, 50, dtype=np.float32)
    np.testing.assert_allclose(smart_normalize(x, "divide-4"), x / 4)
    assert abs(float(smart_normalize(x, "normal").std()) - 1.0) < 1e-5 and float(smart_normalize(x, "0-1").max()) == 1.0
    with pytest.raises(ValueError, match="Unknown smart_normalize"):
        smart_normalize(x, "zscore")
    with pytest.raises(ValueError, match="Unrecognizable file format"):
        LazyVolumeAccessor(str(tmp_path / "v.raw"), kind="image")


def test_zarr_reader_rejects_what_it_cannot_decode(tmp_path):
    z = tmp_path / "b.zarr"
    z.mkdir()
    (z / ".zarray").write_text(json.dumps({"zarr_format": 2, "shape": [4, 4, 4], "chunks": [2, 2, 2], "dtype": "<f4",
                                           "compressor": {"id": "blosc"}, "fill_value": 0, "order": "C", "filters": None}))
    with pytest.raises(NotImplementedError, match="blosc"):
        ZarrV2Array(str(z))
    (z / ".zarray").write_text(json.dumps({"zarr_format": 2, "shape": [4, 4, 4], "chunks": [2, 2, 2], "dtype": "<f4",
                                           "compressor": None, "fill_value": 7, "order": "C", "filters": None}))
    a = ZarrV2Array(str(z))
    assert float(a[1:3, :, 0:2].mean()) == 7.0            # missing chunks read as fill_value


@pytest.mark.parametrize("dtype,compression", [(np.uint8, None), (np.uint16, "tiff_lzw"), (np.float32, "tiff_adobe_deflate")])
def test_tiff_stack_reader_and_accessor(tmp_path, dtype, compression):
    """Multi-page TIFF stacks (what Lucchi++ / SNEMI3D ship): whole-volume read, page-range region reads, and the lazy
    accessor over them give exactly what the same data gives from .npy (reference io.py:199-237, lazy.py:639-676)."""
    from PIL import Image
    from pytorch_connectomics_amd.main import read_volume
    from pytorch_connectomics_amd.utils.tiffstack import TiffStack, read_tiff_volume, tiff_volume_shape
    rng = np.random.default_rng(3)
    vol = (rng.random((7, 12, 10)) * (255 if dtype == np.uint8 else 4000)).astype(dtype)
    path = tmp_path / "stack.tif"
    pages = [Image.fromarray(p) for p in vol]
    pages[0].save(path, save_all=True, append_images=pages[1:], **({"compression": compression} if compression else {}))
    assert tiff_volume_shape(str(path)) == vol.shape
    got = read_tiff_volume(str(path))
    assert got.dtype == vol.dtype
    np.testing.assert_array_equal(got, vol)
    np.testing.assert_array_equal(read_volume(str(path)), vol)
    with TiffStack(str(path)) as st:
        assert st.shape == vol.shape and st.ndim == 3 and len(st) == 7
        np.testing.assert_array_equal(st[2:5, 3:9, 1:7], vol[2:5, 3:9, 1:7])
        np.testing.assert_array_equal(st[-1], vol[-1])
        np.testing.assert_array_equal(st[..., 4], vol[..., 4])
        assert st[3:3].shape == (0, 12, 10)
        with pytest.raises(IndexError):
            st[7]
    np.save(tmp_path / "stack.npy", vol)
    kw = dict(kind="image", transpose_axes=(1, 0, 2), context_pad=((1, 1), (2, 0), (0, 2)), context_pad_mode="reflect",
              normalize_mode="divide-255")
    with LazyVolumeAccessor(str(path), **kw) as a, LazyVolumeAccessor(str(tmp_path / "stack.npy"), **kw) as b:
        assert a.fmt == "tiff" and a.padded_spatial_shape == b.padded_spatial_shape
        np.testing.assert_array_equal(a.read_region((0, 0, 0), a.padded_spatial_shape), b.read_region((0, 0, 0), b.padded_spatial_shape))
        pa = a.read_patch((3, 2, 1), (6, 4, 8), outer_pad_mode="constant", outer_pad_value=0.0)
        np.testing.assert_array_equal(pa, b.read_patch((3, 2, 1), (6, 4, 8), outer_pad_mode="constant", outer_pad_value=0.0))
    one = tmp_path / "one.tif"
    Image.fromarray(vol[0]).save(one)
    assert tiff_volume_shape(str(one)) == vol.shape[1:]
    with pytest.raises(ValueError, match="single-page"):
        LazyVolumeAccessor(str(one), kind="image")
    (tmp_path / "fake.tif").write_bytes(b"not a tiff")
    with pytest.raises(Exception):
        TiffStack(str(tmp_path / "fake.tif"))
