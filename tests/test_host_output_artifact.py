"""Host logic of the prediction transforms and the artifact writer (reference semantics: inference/output.py:150-243,
inference/artifact.py:15-240); expectations restate the reference's numpy operations."""
import json
import os
from pathlib import Path
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from pytorch_connectomics_amd.inference.artifact import (PredictionArtifactMetadata, artifact_attrs,
                                                         build_prediction_artifact_metadata, read_prediction_artifact,
                                                         write_prediction_artifact)
from pytorch_connectomics_amd.inference.output import apply_prediction_transform, apply_storage_dtype_transform


def _cfg(**inf):
    return NS(inference=NS(**inf))


def test_prediction_transform_numpy_semantics():
    rng = np.random.default_rng(0)
    data = rng.uniform(-0.2, 1.3, size=(2, 5, 6, 7)).astype(np.float32)
    assert apply_prediction_transform(NS(), data) is data
    assert apply_prediction_transform(_cfg(), data) is data
    assert apply_prediction_transform(_cfg(prediction_transform=NS(enabled=False)), data) is data
    out = apply_prediction_transform(_cfg(prediction_transform=NS(enabled=True, intensity_scale=255.0, intensity_dtype="uint8")), data)
    want = np.clip(data.astype(np.float32) * 255.0, 0, 255).astype(np.uint8)
    assert out.dtype == np.uint8 and np.array_equal(out, want)
    # negative scale = disabled scaling, dtype still applied; unknown dtype keeps the data
    out = apply_prediction_transform(_cfg(prediction_transform=NS(enabled=True, intensity_scale=-1.0, intensity_dtype="int8")), data)
    assert np.array_equal(out, np.clip(data, -128, 127).astype(np.int8))
    out = apply_prediction_transform(_cfg(prediction_transform=NS(enabled=True, intensity_scale=2.0, intensity_dtype="float8")), data)
    assert out.dtype == np.float32 and np.allclose(out, data * 2.0)
    assert apply_storage_dtype_transform(_cfg(save_dtype=None), data) is data
    st = apply_storage_dtype_transform(_cfg(save_dtype="float16"), data)
    assert st.dtype == np.float16 and np.array_equal(st, data.astype(np.float16))


def test_artifact_roundtrip_and_attrs(tmp_path):
    data = (np.arange(2 * 3 * 4 * 5) % 251).astype(np.uint8).reshape(2, 3, 4, 5)
    cfg = NS(model=NS(arch=NS(type="mednext")), decoding=NS(enabled=False),
             data=NS(data_transform=NS(val_transpose=[2, 1, 0])))
    md = build_prediction_artifact_metadata(cfg, image_path="vol.h5", checkpoint_path="last.ckpt", output_head="aff",
                                            input_shape=(3, 4, 5), final_shape=(3, 4, 5), crop_pad=[(1, 1), (0, 0), (2, 2)],
                                            chunk_shape=(2, 2, 2), halo=(1, 1, 1), intensity_scale=255.0,
                                            intensity_dtype="uint8", extra={"note": "x", "views": [0, 1]})
    attrs = artifact_attrs(md)
    assert attrs["kind"] == "raw_prediction" and attrs["layout"] == "CZYX" and attrs["model_architecture"] == "mednext"
    assert attrs["final_shape"] == json.dumps([3, 4, 5]) and attrs["crop_pad"] == json.dumps([[1, 1], [0, 0], [2, 2]])
    assert attrs["decode_after_inference"] is False and attrs["transpose"] == json.dumps([2, 1, 0])
    assert attrs["note"] == "x" and attrs["views"] == json.dumps([0, 1]) and "activation" not in attrs
    p = write_prediction_artifact(tmp_path / "out" / "pred.h5", data, metadata=md)
    arr, got = read_prediction_artifact(p, return_metadata=True)
    assert np.array_equal(arr, data) and got == attrs
    with pytest.raises(ValueError, match="CZYX"):
        write_prediction_artifact(tmp_path / "bad.h5", data[0])
    with pytest.raises(ValueError, match="shape and dtype"):
        write_prediction_artifact(tmp_path / "bad2.h5", None)
    # streaming mode + default metadata
    p2 = write_prediction_artifact(tmp_path / "s.h5", None, shape=(1, 2, 3, 4), dtype="float32",
                                   writer=lambda d: d.__setitem__((slice(None),) * 4, 1.5))
    arr2, a2 = read_prediction_artifact(p2, return_metadata=True)
    assert arr2.shape == (1, 2, 3, 4) and float(arr2.mean()) == 1.5 and a2["final_shape"] == json.dumps([2, 3, 4])
    assert PredictionArtifactMetadata().kind == "raw_prediction"


def test_transforms_match_reference_fixture():
    """tests/golden/output_transforms.npz: outputs of the reference's apply_prediction_transform /
    apply_storage_dtype_transform (inference/output.py:146-243) for the dtype vocabulary and scale conventions."""
    from pathlib import Path
    z = np.load(Path(__file__).parent / "golden" / "output_transforms.npz")
    data = z["data"]
    cases = {"u8_255": (255.0, "uint8"), "i8_neg_scale": (-1.0, "int8"), "u16_1000": (1000.0, "uint16"), "f16_2": (2.0, "float16"),
             "unknown_dtype": (3.0, "float8"), "scale1_none": (1.0, None), "i32_big": (1.0e6, "int32")}
    for name, (scale, dt) in cases.items():
        cfg = _cfg(prediction_transform=NS(enabled=True, intensity_scale=scale, intensity_dtype=dt), save_dtype=None)
        got = apply_prediction_transform(cfg, data.copy())
        want = z["pt__" + name]
        assert got.dtype == want.dtype and np.array_equal(got, want), name
    for dt in ("uint8", "float16", "int16", "float64"):
        got = apply_storage_dtype_transform(_cfg(save_dtype=dt), data.copy() * 100.0)
        assert got.dtype == z["sd__" + dt].dtype and np.array_equal(got, z["sd__" + dt]), dt
    assert np.array_equal(apply_prediction_transform(_cfg(prediction_transform=NS(enabled=False)), data.copy()), z["pt__disabled"])


class _HeldManager:
    """reference tests/unit/test_inference_stage.py:13-33: a manager that returns a fixed prediction and records the call."""

    def __init__(self, cfg, prediction):
        self.cfg, self.prediction, self.observed = cfg, prediction, {}

    def predict_with_tta(self, images, *, mask=None, mask_align_to_image=False, requested_head=None):
        self.observed = {"images_shape": tuple(images.shape), "mask": mask, "mask_align_to_image": mask_align_to_image,
                         "requested_head": requested_head}
        return self.prediction


def test_run_prediction_inference_writes_raw_artifact_metadata(tmp_path):
    """reference tests/unit/test_inference_stage.py:36-71, same inputs and expectations (HDF5 read back through the same
    backend the reference's decoders would use)."""
    import json
    from pytorch_connectomics_amd.config import ConfigNode, schema_defaults
    from pytorch_connectomics_amd.inference import run_prediction_inference
    from pytorch_connectomics_amd.utils.h5lite import get_h5_backend
    h5 = get_h5_backend()
    if h5 is None:
        pytest.skip("no HDF5 backend")
    cfg = ConfigNode(schema_defaults())
    cfg.model.arch.type = "mednext"
    cfg.data.data_transform.val_transpose = [2, 1, 0]
    cfg.inference.save_dtype = "float16"
    cfg.inference.model.select_channel = [0, 1]
    prediction = torch.arange(1 * 2 * 3 * 4 * 5, dtype=torch.float32).reshape(1, 2, 3, 4, 5)
    manager = _HeldManager(cfg, prediction)
    out = tmp_path / "raw_prediction.h5"
    returned = run_prediction_inference(manager, torch.zeros(1, 1, 3, 4, 5), requested_head="affinity", output_path=out,
                                        image_path="input.h5", checkpoint_path="checkpoint.ckpt", input_shape=(3, 4, 5))
    assert returned is prediction and manager.observed["requested_head"] == "affinity"
    with h5.File(out, "r") as handle:
        d = handle["main"]
        assert tuple(d.shape) == (2, 3, 4, 5) and np.dtype(d.dtype) == np.dtype("float16")
        assert d.attrs["image_path"] == "input.h5" and d.attrs["checkpoint_path"] == "checkpoint.ckpt"
        assert d.attrs["output_head"] == "affinity" and d.attrs["model_architecture"] == "mednext"
        assert d.attrs["model_output_identity"] == "head=affinity;select_channel=[0, 1]"
        assert bool(d.attrs["decode_after_inference"]) is True
        assert json.loads(d.attrs["transpose"]) == [2, 1, 0] and json.loads(d.attrs["final_shape"]) == [3, 4, 5]
        assert np.array_equal(d[...], prediction[0].numpy().astype(np.float16))
    # no output path: prediction only; batch > 1 cannot become one artifact; a contributing rank of a sharded run writes nothing
    assert run_prediction_inference(manager, torch.zeros(1, 1, 3, 4, 5)) is prediction
    with pytest.raises(ValueError, match="one artifact per call"):
        run_prediction_inference(_HeldManager(cfg, torch.zeros(2, 1, 3, 4, 5)), torch.zeros(2, 1, 3, 4, 5), output_path=tmp_path / "b.h5")
    skipper = _HeldManager(cfg, torch.empty(0))
    skipper.should_skip_postprocess_on_rank = lambda: True
    assert run_prediction_inference(skipper, torch.zeros(1, 1, 3, 4, 5), output_path=tmp_path / "none.h5").numel() == 0
    assert not (tmp_path / "none.h5").exists()


def test_run_prediction_inference_honors_save_compression(tmp_path):
    """reference test_inference_stage.py:74-99: inference.save_compression reaches the HDF5 writer."""
    from pytorch_connectomics_amd.config import ConfigNode, schema_defaults
    from pytorch_connectomics_amd.inference import run_prediction_inference
    from pytorch_connectomics_amd.utils.h5lite import get_h5_backend
    h5 = get_h5_backend()
    if h5 is None:
        pytest.skip("no HDF5 backend")
    cfg = ConfigNode(schema_defaults())
    cfg.model.arch.type = "mednext"
    prediction = torch.arange(60, dtype=torch.float32).reshape(1, 1, 3, 4, 5)
    for comp, want in (("gzip", "gzip"), ("none", None)):
        cfg.inference.save_compression = comp
        out = tmp_path / f"raw_{comp}.h5"
        run_prediction_inference(_HeldManager(cfg, prediction), torch.zeros(1, 1, 3, 4, 5), output_path=out, image_path="input.h5",
                                 checkpoint_path="checkpoint.ckpt", input_shape=(3, 4, 5))
        with h5.File(out, "r") as handle:
            assert handle["main"].compression == want
            assert np.array_equal(handle["main"][...], prediction[0].numpy())


def test_array_valued_hdf5_attributes_do_not_overrun_the_scalar_reader(tmp_path):
    """third-party EM files carry array attributes (e.g. `resolution`); the scalar reader used to H5Aread them into
    one int64 / double.  They now come back as numpy arrays, and `dict(attrs)` over a mix of scalar and array attributes works."""
    from pytorch_connectomics_amd.utils import h5lite
    if not h5lite.available():
        pytest.skip("no libhdf5 in this image")
    path = tmp_path / "a.h5"
    with h5lite.File(str(path), "w") as f:
        ds = f.create_dataset("main", data=np.zeros((2, 3), np.float32))
        ds.attrs["resolution"] = [30, 8, 8]
        ds.attrs["scale"] = np.array([1.5, 2.5, -3.25])
        ds.attrs["n"] = 3
        ds.attrs["name"] = "vol"
        ds.attrs["flag"] = True
    with h5lite.File(str(path), "r") as f:
        a = dict(f["main"].attrs.items())
    assert a["n"] == 3 and a["name"] == "vol" and a["flag"] is True
    assert a["resolution"].dtype == np.int64 and a["resolution"].tolist() == [30, 8, 8]
    assert a["scale"].dtype == np.float64 and a["scale"].tolist() == [1.5, 2.5, -3.25]


def test_integer_array_attributes_are_exact_beyond_2_pow_53_and_long(tmp_path):
    """integer array attributes went through float64 (ids / offsets above 2^53 corrupted) and arrays longer than 4096
    elements raised KeyError; they travel as 8-byte integers now and have no length cap."""
    from pytorch_connectomics_amd.utils import h5lite
    if not h5lite.available():
        pytest.skip("no libhdf5 in this image")
    big = np.array([2 ** 53 + 1, 2 ** 63 - 1, -(2 ** 62) - 3], np.int64)
    ubig = np.array([2 ** 64 - 1, 2 ** 53 + 1], np.uint64)
    long = np.arange(6000, dtype=np.int32)            # 48 kB: under HDF5's own 64 kB object-header limit
    with h5lite.File(str(tmp_path / "b.h5"), "w") as f:
        ds = f.create_dataset("main", data=np.zeros((2,), np.uint8))
        ds.attrs["ids"], ds.attrs["uids"], ds.attrs["long"] = big, ubig, long
    with h5lite.File(str(tmp_path / "b.h5"), "r") as f:
        a = f["main"].attrs
        assert a["ids"].dtype == np.int64 and np.array_equal(a["ids"], big)
        assert a["uids"].dtype == np.uint64 and np.array_equal(a["uids"], ubig)
        assert np.array_equal(a["long"], long)


def test_output_file_names_and_hdf5_writer(tmp_path):
    """`resolve_output_filenames` / `write_outputs` (reference inference/output.py:19-83, :252-356; stem rule of
    runtime/output_naming.py:54-94; fuzzed against the reference in tools/diff_fuzz_reference.py): uninformative stems climb to the
    first informative directory, container directories are skipped, items without a name get `volume_<step>_<index>`; predictions
    land in `<save_path>/<stem>/<suffix stem>.h5` as dataset `main` in the storage dtype; other backends are refused by name."""
    from types import SimpleNamespace as NS
    from pytorch_connectomics_amd.inference import resolve_output_filenames, write_outputs
    from pytorch_connectomics_amd.inference.output import volume_stem_from_path
    from pytorch_connectomics_amd.utils import h5lite
    stems = {"/data/seed101/data.zarr/img": "seed101", "/data/seed101/img.h5": "seed101", "/data/sample.h5": "sample",
             "/data/seed101/raw_aff.h5": "raw_aff", "img.tif": "volume", "/a/b.n5/data/raw": "a", "/data/vol.ome.zarr/0": "0"}
    for path, want in stems.items():
        assert volume_stem_from_path(path) == want, path
    batch = {"image": np.zeros((3, 1, 2, 2, 2)), "image_meta_dict": [{"filename_or_obj": "/d/s1/img.h5"}, {"filename_or_obj": "/d/s2/em.h5"}]}
    assert resolve_output_filenames(None, batch, global_step=4) == ["s1", "s2", "volume_4_2"]
    assert resolve_output_filenames(None, {"image": "/d/s3/vol.h5"}) == ["vol"]
    assert resolve_output_filenames(None, {"image": np.zeros((2, 1, 2, 2, 2)), "image_meta_dict": {"filename_or_obj": ["/x/a.h5", None]}}, 1) == ["a", "volume_1_1"]
    if not h5lite.available():
        pytest.skip("libpytc_h5.so not built")
    cfg = NS(inference=NS(save_path=str(tmp_path / "out"), save_backend="h5", save_dtype="float16"), data=NS(nnunet_preprocessing=None))
    preds = np.random.default_rng(0).random((2, 2, 3, 4, 5)).astype(np.float32)
    write_outputs(cfg, preds, ["s1", "s2"], suffix="prediction.h5")
    for i, stem in enumerate(("s1", "s2")):
        with h5lite.File(str(tmp_path / "out" / stem / "prediction.h5"), "r") as fh:
            got = np.asarray(fh["main"][...])
        assert got.dtype == np.float16 and got.shape == (2, 3, 4, 5)
        np.testing.assert_array_equal(got, preds[i].astype(np.float16))
    cfg.inference.save_backend = "tiff"
    with pytest.raises(NotImplementedError, match="save_backend='tiff'"):
        write_outputs(cfg, preds, ["s1", "s2"])
    write_outputs(NS(inference=NS(save_path=None)), preds, ["s1", "s2"])          # no output directory: nothing to do


def test_lzf_filter_codec_and_hdf5_round_trip(tmp_path):
    """`compression="lzf"` (h5py's filter id 32000, which the shim registers on libhdf5): the codec against hand-built vectors of the LZF
    byte format (literal run; overlapping short match; long match with its extra length byte; malformed distance; short output),
    round trips of compressible / incompressible buffers, and datasets written and read through HDF5 -- an incompressible chunk is
    stored raw (optional filter) and still reads back (reference tests/unit/test_inference_stage.py:79-97 saves with lzf)."""
    import ctypes as C
    from pytorch_connectomics_amd.utils import h5lite
    if not h5lite.available():
        pytest.skip("libpytc_h5.so not built")
    lib = h5lite._need()

    def unpack(stream: bytes, cap: int):
        out = (C.c_ubyte * max(cap, 1))()
        n = lib.pytc_h5_lzf_unpack(stream, len(stream), out, cap)
        return n, bytes(out[:max(n, 0)])

    def pack(raw: bytes) -> bytes:
        out = (C.c_ubyte * (len(raw) + 64))()
        return bytes(out[:lib.pytc_h5_lzf_pack(raw, len(raw), out, len(raw) + 64)])
    assert unpack(bytes([4]) + b"hello", 16) == (5, b"hello")
    assert unpack(bytes([0]) + b"a" + bytes([(1 << 5) | 0, 0]), 16) == (4, b"aaaa")
    assert unpack(bytes([2]) + b"abc" + bytes([(7 << 5) | 0, 3, 2]), 64) == (15, b"abc" * 5)
    assert unpack(bytes([0]) + b"a" + bytes([(1 << 5) | 0, 5]), 16)[0] == 0
    assert unpack(bytes([4]) + b"hello", 3)[0] == -1
    rng = np.random.default_rng(0)
    for raw in (b"abc", b"aaaa", bytes(1000), bytes(range(256)) * 40, rng.integers(0, 4, 70000, dtype=np.uint8).tobytes(),
                (b"x" * 33 + b"y") * 500, np.repeat(rng.integers(0, 256, 300, dtype=np.uint8), 300).tobytes()):
        packed = pack(raw)
        assert packed and unpack(packed, len(raw)) == (len(raw), raw)
    assert len(pack(bytes(100000))) < 1500 and pack(rng.integers(0, 256, 5000, dtype=np.uint8).tobytes()) == b""     # noise: "does not shrink"
    vol = np.repeat(rng.integers(0, 255, (8, 16, 4), dtype=np.uint8), 8, axis=2)
    noise = rng.random((4, 8, 8)).astype(np.float32)
    with h5lite.File(tmp_path / "z.h5", "w") as f:
        f.create_dataset("main", data=vol, chunks=(4, 8, 16), compression="lzf")
        f.create_dataset("noise", data=noise, chunks=(2, 8, 8), compression="lzf")
    with h5lite.File(tmp_path / "z.h5", "r") as f:
        assert f["main"].compression == "lzf" and f["noise"].compression == "lzf"
        assert np.array_equal(f["main"][...], vol) and np.array_equal(f["noise"][...], noise)
        assert np.array_equal(f["main"][2:7, 3:11, 5:30], vol[2:7, 3:11, 5:30])
    with pytest.raises(NotImplementedError, match="gzip \\(deflate\\) or lzf"):
        with h5lite.File(tmp_path / "bad.h5", "w") as f:
            f.create_dataset("main", data=vol, compression="szip")


def test_parallel_deflate_writer_round_trips_and_matches_the_serial_writer(tmp_path, monkeypatch):
    """csrc/host/h5io.c pytc_h5_dset_write_parallel: N threads deflate whole HDF5 chunks, H5Dwrite_chunk stores them.  The file is an
    ordinary gzip-chunked dataset: read back through H5Dread (the library's own inflate) it equals the source, edge chunks included
    (C, 64, 64, 64 chunks over extents that are not multiples of 64), slab by slab as the chunked runner writes (chunk-aligned z slabs),
    and a region that is NOT chunk aligned silently takes the serial path."""
    import numpy as np
    from pytorch_connectomics_amd.utils import h5lite
    if not h5lite.available():
        pytest.skip("libpytc_h5.so not built")
    rng = np.random.default_rng(0)
    data = rng.standard_normal((3, 130, 70, 96)).astype(np.float32)
    data[:, 40:90] = 0.0                                   # compressible stretch
    monkeypatch.setattr(h5lite, "PARALLEL_WRITE_MIN_BYTES", 1)
    for threads, name in ((4, "par.h5"), (1, "ser.h5")):
        monkeypatch.setenv("PYTC_H5_THREADS", str(threads))
        with h5lite.File(tmp_path / name, "w") as f:
            d = f.create_dataset("main", shape=data.shape, dtype=np.float32, chunks=(3, 64, 64, 64), compression="gzip")
            d[:, 0:64] = data[:, 0:64]                     # aligned slab
            d[:, 64:130] = data[:, 64:130]                 # aligned start, ends at the dataset's end (edge chunks in z, y, x)
            d.attrs["k"] = 3
    for name in ("par.h5", "ser.h5"):
        with h5lite.File(tmp_path / name, "r") as f:
            assert f["main"].compression == "gzip" and f["main"].chunks == (3, 64, 64, 64)
            assert np.array_equal(f["main"][...], data)
            assert np.array_equal(f["main"][1, 60:70, 3:9, 90:96], data[1, 60:70, 3:9, 90:96])
    # unaligned region: falls back to H5Dwrite (rc 2), still correct
    monkeypatch.setenv("PYTC_H5_THREADS", "4")
    with h5lite.File(tmp_path / "un.h5", "w") as f:
        d = f.create_dataset("main", shape=(2, 100, 100), dtype=np.int16, chunks=(1, 32, 32), compression="gzip")
        src = rng.integers(-5, 5, size=(2, 100, 100)).astype(np.int16)
        d[:, 5:77, :] = src[:, 5:77, :]
        d[:, 0:5] = src[:, 0:5]
        d[:, 77:] = src[:, 77:]
    with h5lite.File(tmp_path / "un.h5", "r") as f:
        assert np.array_equal(f["main"][...], src)
    # an unfiltered chunked dataset takes the direct path too
    with h5lite.File(tmp_path / "raw.h5", "w") as f:
        d = f.create_dataset("main", shape=(65, 33), dtype=np.float64, chunks=(16, 16))
        src2 = rng.standard_normal((65, 33))
        d[...] = src2
    with h5lite.File(tmp_path / "raw.h5", "r") as f:
        assert np.array_equal(f["main"][...], src2)


def test_parallel_writer_deflate_backends_and_thread_policy(tmp_path, monkeypatch):
    """The deflate workers use libdeflate when the image has it (bound at run time, no header) and zlib otherwise or under PYTC_H5_DEFLATE=zlib:
    both produce zlib streams that HDF5's own gzip filter inflates back to the source.  The backend is chosen once per process, so the
    zlib arm runs in a child process.  write_threads() honours a cgroup CPU quota (two workers per granted core)."""
    import subprocess
    import sys
    import numpy as np
    from pytorch_connectomics_amd.utils import h5lite
    if not h5lite.available():
        pytest.skip("libpytc_h5.so not built")
    rng = np.random.default_rng(1)
    data = (1.0 / (1.0 + np.exp(-rng.standard_normal((2, 70, 64, 130)).astype(np.float32)))).astype(np.float32)
    np.save(tmp_path / "src.npy", data)
    monkeypatch.setattr(h5lite, "PARALLEL_WRITE_MIN_BYTES", 1)
    monkeypatch.setenv("PYTC_H5_THREADS", "3")
    with h5lite.File(tmp_path / "here.h5", "w") as f:
        f.create_dataset("main", data=data, chunks=(2, 64, 64, 64), compression="gzip")
    st = h5lite.last_parallel_write_stats()
    assert st["deflate"] in ("libdeflate", "zlib") and st["threads"] == 3 and st["deflate_thread_s"] > 0 and st["wall_s"] > 0
    child = (
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {str(Path(__file__).resolve().parents[1])!r})\n"
        "from pytorch_connectomics_amd.utils import h5lite\n"
        "h5lite.PARALLEL_WRITE_MIN_BYTES = 1\n"
        f"data = np.load({str(tmp_path / 'src.npy')!r})\n"
        f"with h5lite.File({str(tmp_path / 'zlib.h5')!r}, 'w') as f:\n"
        "    f.create_dataset('main', data=data, chunks=(2, 64, 64, 64), compression='gzip')\n"
        "print(h5lite.last_parallel_write_stats()['deflate'])\n")
    env = dict(os.environ, PYTC_H5_DEFLATE="zlib", PYTC_H5_THREADS="3")
    out = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1] == "zlib"
    for name in ("here.h5", "zlib.h5"):
        with h5lite.File(tmp_path / name, "r") as f:
            assert f["main"].compression == "gzip" and np.array_equal(f["main"][...], data), name
    # thread policy: PYTC_H5_THREADS wins; otherwise a quota of q CPUs gives 2 q workers (at most 128, at most the affinity mask)
    #  a write that does not take the parallel path leaves no stale record behind
    with h5lite.File(tmp_path / "here.h5", "r+") as f:
        f["main"][0, 0, 0, :4] = np.zeros(4, np.float32)                   # a few bytes below the threshold
    monkeypatch.setattr(h5lite, "PARALLEL_WRITE_MIN_BYTES", 4 << 20)
    with h5lite.File(tmp_path / "small.h5", "w") as f:
        f.create_dataset("main", data=data[:, :4], chunks=(2, 4, 64, 64), compression="gzip")
    assert h5lite.last_parallel_write_stats() is None
    monkeypatch.delenv("PYTC_H5_THREADS")
    monkeypatch.setattr(h5lite, "_quota_cache", h5lite._UNSET)
    monkeypatch.setattr(h5lite, "_cgroup_cpu_quota", lambda: 1.0)
    assert h5lite.write_threads() == min(2, len(os.sched_getaffinity(0)))
    calls = []
    monkeypatch.setattr(h5lite, "_cgroup_cpu_quota", lambda: calls.append(1))
    assert h5lite.write_threads() == min(2, len(os.sched_getaffinity(0))) and not calls       # the quota is read once per process
    monkeypatch.setattr(h5lite, "_quota_cache", h5lite._UNSET)
    monkeypatch.setattr(h5lite, "_cgroup_cpu_quota", lambda: None)
    assert h5lite.write_threads() == min(128, len(os.sched_getaffinity(0)))


def test_cgroup_cpu_quota_parsing(monkeypatch):
    """h5lite._cgroup_cpu_quota: cgroup v2 `cpu.max` ("<quota> <period>" or "max <period>"), cgroup v1 cfs quota / period (-1 = unlimited),
    neither file -> None.  The MI355X boxes grant 16 CPUs ("1600000 100000") under a 256-core affinity mask."""
    import builtins
    import io
    from pytorch_connectomics_amd.utils import h5lite
    real_open = builtins.open

    def fake(files):
        def _open(path, *a, **k):
            p = str(path)
            if p.startswith("/sys/fs/cgroup/"):
                if p in files:
                    return io.StringIO(files[p])
                raise FileNotFoundError(p)
            return real_open(path, *a, **k)
        return _open

    v2, q1, p1 = "/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"
    cases = [({v2: "1600000 100000\n"}, 16.0), ({v2: "max 100000\n"}, None), ({v2: "150000 100000"}, 1.5),
             ({q1: "400000\n", p1: "100000\n"}, 4.0), ({q1: "-1\n", p1: "100000\n"}, None), ({}, None), ({v2: "garbage"}, None)]
    for files, want in cases:
        monkeypatch.setattr(builtins, "open", fake(files))
        assert h5lite._cgroup_cpu_quota() == want, (files, want)
    monkeypatch.setattr(builtins, "open", fake({v2: "1600000 100000"}))
    monkeypatch.delenv("PYTC_H5_THREADS", raising=False)
    assert h5lite.write_threads() == min(32, len(os.sched_getaffinity(0)))
