"""CPU tests: chunk grid / halo / manifest (vs reference fixtures), lazy window grids, and the multi-rank
chunk-sharding protocol of chunked inference (2 gloo ranks, injected region predictor)."""
import json
import os
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from pytorch_connectomics_amd.chunked import (ManifestConfigMismatch, ResumeManifest, build_chunk_grid,
                                              resolve_halo_region)
from pytorch_connectomics_amd.inference import lazy as L
from pytorch_connectomics_amd.inference.chunked import (resolve_chunk_shape, run_chunked_prediction_inference,
                                                        stitch_chunk_prediction_files)


def test_chunk_grid_and_halo_match_reference(golden_dir):
    g = np.load(golden_dir / "chunk_grid.npz")
    for i in range(int(g["n"])):
        vol, ch, halo, crop = g[f"vol_{i}"], g[f"chunk_{i}"], g[f"halo_{i}"], g[f"crop_{i}"]
        in_shape = tuple(int(v) + 2 * int(c) for v, c in zip(vol, crop))
        rows, keys = [], []
        for r in build_chunk_grid(vol, ch):
            rs, re, sl = resolve_halo_region(r, in_shape, halo=halo, crop_before=crop)
            rows.append(list(r.index) + list(r.start) + list(r.stop) + list(rs) + list(re)
                        + [s.start for s in sl] + [s.stop for s in sl])
            keys.append(r.key)
        assert np.array_equal(np.asarray(rows, np.int64), g[f"rows_{i}"])
        assert keys == list(g[f"keys_{i}"])
    refs = build_chunk_grid((9, 9, 9), (4, 4, 4))
    # grid covers the volume without overlap (reference test_chunked_inference.py:33)
    cover = np.zeros((9, 9, 9), int)
    for r in refs:
        cover[r.slices] += 1
        assert r.shape == tuple(b - a for a, b in zip(r.start, r.stop))
    assert cover.min() == 1 and cover.max() == 1
    with pytest.raises(ValueError):
        build_chunk_grid((9, 9), (4, 4, 4))


def test_resume_manifest(tmp_path):
    p = tmp_path / "m.json"
    m = ResumeManifest.load_or_create(p, {"chunk_shape": [4, 4, 4], "overlap": 0.5})
    m.mark_completed("z0_y0_x0")
    m.mark_many(["z0_y0_x1", "z0_y0_x0"])
    assert json.loads(p.read_text())["completed"] == ["z0_y0_x0", "z0_y0_x1"]
    again = ResumeManifest.load_or_create(p, {"chunk_shape": [4, 4, 4], "overlap": 0.5})
    assert again.completed == {"z0_y0_x0", "z0_y0_x1"}
    with pytest.raises(ManifestConfigMismatch, match="chunk_shape"):
        ResumeManifest.load_or_create(p, {"chunk_shape": [8, 4, 4]})
    fresh = ResumeManifest.load_or_create(p, {"chunk_shape": [8, 4, 4]}, overwrite=True)
    assert fresh.completed == set() and not (tmp_path / "m.json.tmp").exists()


def test_lazy_window_grids(golden_dir):
    g = np.load(golden_dir / "lazy.npz")
    for i in range(4):
        m = g[f"grid{i}_meta"]
        offs = L._build_window_axis_offsets(m[:3], m[3:6], tuple(g[f"grid{i}_ov"]), snap_to_edge=bool(m[6]))
        for a in range(3):
            assert np.array_equal(np.asarray(offs[a]), g[f"grid{i}_axis{a}"])
    # region grid is a subset of the global grid (reference test_lazy_inference.py)
    full = set(tuple(s.start for s in sl) for sl in L._build_window_slices((20, 30, 34), (8, 12, 16), (0.5,) * 3, snap_to_edge=False))
    part = [tuple(s.start for s in sl) for sl in L._build_intersecting_window_slices(
        (20, 30, 34), (8, 12, 16), (0.5,) * 3, region_start=(3, 5, 7), region_stop=(17, 22, 30), snap_to_edge=False)]
    assert set(part) <= full and 0 < len(part) < len(full)
    assert L._snap_offsets(4, 8, 2) == [0]
    assert L._resolve_target_context(NS(target_context=[2]), (8, 8, 8)) == (2, 2, 2)
    with pytest.raises(ValueError, match="length 1 or 3"):
        L._resolve_target_context(NS(target_context=[1, 2]), (8, 8, 8))


def _cfg(chunk, halo=(0, 0, 0), axes="all", shard=None):
    return NS(inference=NS(chunking=NS(enabled=True, chunk_size=list(chunk), halo=list(halo), axes=axes,
                                       shard_id=None if shard is None else shard[0],
                                       num_shards=None if shard is None else shard[1])))


def _fake_predictor(vol):
    def fn(start, stop):      # position-dependent "prediction": 2 channels derived from the volume itself
        sl = tuple(slice(a, b) for a, b in zip(start, stop))
        v = torch.from_numpy(vol[sl])
        return torch.stack([v, v * 2 + 1], 0).unsqueeze(0)
    return fn


def test_chunked_single_process_and_external_shards(tmp_path):
    vol = np.random.default_rng(0).random((10, 13, 9)).astype(np.float32)
    assert resolve_chunk_shape(_cfg((4, 5, 6), axes="z"), vol.shape) == (4, 13, 9)
    out = run_chunked_prediction_inference(_cfg((4, 5, 6), halo=(1, 2, 1)), None, vol, output_path=tmp_path / "a.npy",
                                           predict_region_fn=_fake_predictor(vol))
    np.testing.assert_array_equal(out, np.stack([vol, vol * 2 + 1]))
    # external shards: same code run N times sequentially, then stitched (reference test_chunked_inference.py:328-380)
    for sid in range(3):
        r = run_chunked_prediction_inference(_cfg((4, 5, 6), shard=(sid, 3)), None, vol, output_path=tmp_path / "b.npy",
                                             predict_region_fn=_fake_predictor(vol))
        assert r is None
    chunks = build_chunk_grid(vol.shape, (4, 5, 6))
    np.testing.assert_array_equal(stitch_chunk_prediction_files(tmp_path / "b.npy", chunks, vol.shape),
                                  np.stack([vol, vol * 2 + 1]))
    with pytest.raises(ValueError, match="shard_id"):
        run_chunked_prediction_inference(_cfg((4, 5, 6), shard=(3, 3)), None, vol, output_path=tmp_path / "c.npy",
                                         predict_region_fn=_fake_predictor(vol))


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    vol = np.random.default_rng(0).random((10, 13, 9)).astype(np.float32)
    calls = []
    base = _fake_predictor(vol)

    def fn(start, stop):
        calls.append(tuple(start))
        return base(start, stop)

    out = run_chunked_prediction_inference(_cfg((4, 5, 6)), None, vol, output_path=os.path.join(tmp, "d.npy"),
                                           predict_region_fn=fn)
    chunks = build_chunk_grid(vol.shape, (4, 5, 6))
    mine = [c for i, c in enumerate(chunks) if i % world == rank]
    assert sorted(calls) == sorted(tuple(c.start) for c in mine)       # idx % world == rank (chunked.py:471)
    if rank == 0:
        np.testing.assert_array_equal(out, np.stack([vol, vol * 2 + 1]))
    else:
        assert out is None
    torch.distributed.destroy_process_group()


def test_chunked_two_ranks_gloo(tmp_path):
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)


def _worker_c4(rank, world, port, tmp):
    """BASELINE configs[3] geometry scaled by 1/40: volume 16^3, chunk 8^3 (2 x 2 x 2 = 8 chunks, one per rank at world 8), halo 2."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    vol = np.random.default_rng(4).random((16, 16, 16)).astype(np.float32)
    calls = []
    base = _fake_predictor(vol)

    def fn(start, stop):
        calls.append(tuple(start))
        return base(start, stop)

    out = run_chunked_prediction_inference(_cfg((8, 8, 8)), None, vol, output_path=os.path.join(tmp, "c4.npy"), predict_region_fn=fn)
    chunks = build_chunk_grid(vol.shape, (8, 8, 8))
    assert len(chunks) == 8 and len(calls) == 1 and calls[0] == tuple(chunks[rank].start)        # idx % world == rank: one chunk each
    if rank == 0:
        np.testing.assert_array_equal(out, np.stack([vol, vol * 2 + 1]))
    else:
        assert out is None
    torch.distributed.destroy_process_group()


def test_chunked_eight_ranks_gloo_one_chunk_per_rank(tmp_path):
    """The 8-GPU chunked configuration (MitoEM-R, 640^3 in 320^3 chunks): every rank predicts exactly its chunk, rank 0 stitches after
    the barrier (chunked.py:471, 666) -- at the rank count the protocol is meant for."""
    port = 31500 + os.getpid() % 2000
    mp.spawn(_worker_c4, args=(8, port, str(tmp_path)), nprocs=8, join=True)


def test_chunked_with_user_crop_and_deepem_affinity_border(tmp_path):
    """The chunk grid covers the CROPPED output space; the stitched result equals the cropped whole-volume prediction
    (reference chunked.py:743-755, chunk_grid.py:56-77)."""
    vol = np.random.default_rng(1).random((12, 13, 11)).astype(np.float32)
    cfg = _cfg((4, 5, 6), halo=(1, 1, 1))
    cfg.inference.model = NS(crop_pad=[1, 0, 0, 2, 0, 0], select_channel=None, head=None)
    cfg.model = NS(primary_head=None, heads=None, out_channels=2)
    cfg.data = NS(label_transform=NS(stack_outputs=True, targets=[
        {"name": "affinity", "kwargs": {"offsets": ["0-0-1", "0-2-0"], "affinity_mode": "deepem"}}]))
    out = run_chunked_prediction_inference(cfg, None, vol, output_path=tmp_path / "c.npy", predict_region_fn=_fake_predictor(vol))
    want = np.stack([vol, vol * 2 + 1])[:, 1:, 2:-2, 1:]         # crop_pad (1,0),(0,2),(0,0) + DeepEM border (0,0),(2,0),(1,0)
    assert out.shape == want.shape
    np.testing.assert_array_equal(out, want)
    cfg.inference.model.crop_pad = [6, 6, 0, 0, 0, 0]
    with pytest.raises(ValueError, match="too large"):
        run_chunked_prediction_inference(cfg, None, vol, output_path=tmp_path / "d.npy", predict_region_fn=_fake_predictor(vol))


def test_chunk_writer_orders_manifest_after_the_file_and_surfaces_errors(tmp_path):
    """The background writer marks a chunk completed only after its file is in place, and an IO error reaches the caller."""
    from pytorch_connectomics_amd.inference.chunked import _ChunkWriter

    class Manifest:
        def __init__(self):
            self.seen = []

        def mark_completed(self, key):
            self.seen.append((key, (tmp_path / f"{key}.npy").exists()))

    man = Manifest()
    w = _ChunkWriter(man)
    for i in range(5):
        w.submit(torch.full((2, 3, 4, 5), float(i)), tmp_path / f"c{i}.npy", f"c{i}")
    w.close()
    assert man.seen == [(f"c{i}", True) for i in range(5)]
    assert float(np.load(tmp_path / "c3.npy").mean()) == 3.0
    bad = _ChunkWriter(Manifest())
    bad.submit(np.zeros((1, 2, 2, 2), np.float32), tmp_path / "missing_dir" / "x.npy", "x")
    with pytest.raises(OSError):
        bad.close()


def test_minimal_rsunet_tutorial_config_parses():
    import os
    from pytorch_connectomics_amd.config import load_config
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tutorials", "minimal_rsunet.yaml")
    tr = load_config(path, mode="train")
    assert tr.model.arch.type == "rsunet" and list(tr.model.rsunet.width) == [8, 16] and tr.optimization.n_steps_per_epoch == 2
    te = load_config(path, mode="test", overrides=["inference.window.overlap=0.25"]) if "overrides" in load_config.__code__.co_varnames else load_config(path, mode="test")
    assert str(te.data.test.image).startswith("random://") and list(te.inference.sliding_window.window_size) == [32, 64, 64]


def test_resume_manifest_file_format_interchanges_with_reference(tmp_path, golden_dir):
    """tests/golden/resume_manifest_ref.json was written by the reference's ResumeManifest: this implementation resumes
    from it, refuses a mismatched config like the reference, and writes the same JSON document for the same history."""
    ref_text = (golden_dir / "resume_manifest_ref.json").read_text()
    cfg = {"chunk_shape": [4, 5, 6], "output_shape": [10, 13, 9], "halo": [1, 2, 1], "overlap": 0}
    p = tmp_path / "m.json"
    p.write_text(ref_text)
    m = ResumeManifest.load_or_create(p, cfg)
    assert m.completed == {"z0_y0_x0", "z1_y2_x1", "z2_y0_x1", "z2_y2_x1"}
    with pytest.raises(ManifestConfigMismatch):
        ResumeManifest.load_or_create(p, dict(cfg, chunk_shape=[4, 5, 7]))
    q = tmp_path / "own.json"
    own = ResumeManifest.load_or_create(q, cfg)
    for k in ("z0_y0_x0", "z1_y2_x1", "z0_y0_x0", "z2_y0_x1"):
        own.mark_completed(k)
    own.mark_many(["z2_y2_x1", "z1_y2_x1"])
    assert json.loads(q.read_text()) == json.loads(ref_text)
    assert ResumeManifest.load_or_create(q, cfg, overwrite=True).completed == set()


# ------------------------------------------------------------------------------------------------ HDF5 container / ROI
h5 = pytest.importorskip("pytorch_connectomics_amd.utils.h5lite")
needs_h5 = pytest.mark.skipif(h5.get_h5_backend() is None, reason="no HDF5 backend (h5py / libhdf5) on this box")


@needs_h5
def test_chunked_hdf5_layout_matches_reference_vocabulary(tmp_path):
    """chunk_{key}.h5 files (dataset `main` CZYX, gzip, (C, <=64^3) HDF5 chunks, the reference's attribute vocabulary),
    <output>.index.json in the reference's field names, stitched artifact streamed by z slabs (reference chunked.py:279-434)."""
    be = h5.get_h5_backend()
    vol = np.random.default_rng(3).random((10, 13, 9)).astype(np.float32)
    cfg = _cfg((4, 5, 6), halo=(1, 2, 1))
    cfg.model = NS(arch=NS(type="mednext"))
    cfg.inference.prediction_transform = NS(enabled=True, intensity_scale=255.0, intensity_dtype="uint8")
    out = run_chunked_prediction_inference(cfg, None, vol, output_path=tmp_path / "pred.h5", predict_region_fn=_fake_predictor(vol),
                                           image_path="vol.h5", checkpoint_path="last.ckpt")
    want = np.clip(np.stack([vol, vol * 2 + 1]) * np.float32(255.0), 0, 255).astype(np.uint8)
    assert out.dtype == np.uint8 and np.array_equal(out, want)
    files = sorted(p.name for p in (tmp_path / "pred.h5.chunks").glob("chunk_*.h5"))
    assert len(files) == 3 * 3 * 2 and files[0] == "chunk_z0_y0_x0.h5"
    with be.File(tmp_path / "pred.h5.chunks" / "chunk_z1_y2_x1.h5", "r") as fh:
        d = fh["main"]
        assert d.shape == (2, 4, 3, 3) and d.dtype == np.uint8 and d.chunks == (2, 4, 3, 3)
        a = dict(d.attrs.items())
        assert a["kind"] == "raw_prediction" and a["layout"] == "CZYX" and a["chunk_key"] == "z1_y2_x1"
        assert json.loads(a["chunk_start_zyx"]) == [4, 10, 6] and json.loads(a["chunk_stop_zyx"]) == [8, 13, 9]
        assert json.loads(a["chunk_read_start_zyx"]) == [3, 8, 5] and json.loads(a["chunk_read_stop_zyx"]) == [9, 13, 9]
        assert json.loads(a["halo"]) == [1, 2, 1] and a["intensity_dtype"] == "uint8" and a["intensity_scale"] == 255.0
        assert a["image_path"] == "vol.h5" and a["checkpoint_path"] == "last.ckpt" and a["model_architecture"] == "mednext"
        assert np.array_equal(d[...], want[:, 4:8, 10:13, 6:9])
    idx = json.loads((tmp_path / "pred.h5.index.json").read_text())
    assert set(idx) == {"input_shape", "final_shape", "chunk_shape", "halo", "crop_pad", "checkpoint_path", "world_size", "chunks"}
    assert idx["final_shape"] == [10, 13, 9] and idx["world_size"] == 1 and len(idx["chunks"]) == 18
    assert idx["chunks"][0] == {"key": "z0_y0_x0", "index_zyx": [0, 0, 0], "start_zyx": [0, 0, 0], "stop_zyx": [4, 5, 6],
                               "path": "pred.h5.chunks/chunk_z0_y0_x0.h5"}
    with be.File(tmp_path / "pred.h5", "r") as fh:
        d = fh["main"]
        assert d.shape == (2, 10, 13, 9) and d.chunks == (2, 10, 13, 9)
        a = dict(d.attrs.items())
        assert json.loads(a["final_shape"]) == [10, 13, 9] and "chunk_stitch_source" in a and a["compression"] == "gzip"
    # resume: every chunk is on disk and in the manifest -> nothing is predicted again
    def boom(*_a):
        raise AssertionError("chunk recomputed on resume")
    again = run_chunked_prediction_inference(cfg, None, vol, output_path=tmp_path / "pred.h5", predict_region_fn=boom)
    assert np.array_equal(again, want)
    cfg.inference.save_backend = "zarr"
    with pytest.raises(ValueError, match="single streamed HDF5"):
        run_chunked_prediction_inference(cfg, None, vol, output_path=tmp_path / "x.h5", predict_region_fn=boom)
    cfg.inference.save_backend = "h5"


def test_chunked_precomputed_layer_output(tmp_path):
    """inference.chunking.precomputed: chunk predictions go straight into a raw-encoded neuroglancer precomputed layer (reference
    chunked.py:485-507, :590-612: info fields of CloudVolume.create_new_info, storage-chunk-aligned writes, `.done` markers for
    resume, no stitching).  The layer is read back with an independent decoder of the format specification."""
    import gzip
    from pytorch_connectomics_amd.inference.precomputed import PrecomputedLayer, validate_precomputed_alignment
    vol = np.random.default_rng(5).random((8, 12, 10)).astype(np.float32)
    cfg = _cfg((4, 6, 10), halo=(1, 2, 0))
    cfg.inference.prediction_transform = NS(enabled=True, intensity_scale=255.0, intensity_dtype="uint8")
    ch = cfg.inference.chunking
    ch.precomputed, ch.precomputed_resolution, ch.precomputed_chunk_size = True, [8, 8, 30], [5, 3, 2]
    ch.precomputed_affinity_convention = "none"
    out = run_chunked_prediction_inference(cfg, None, vol, output_path=tmp_path / "pred.h5", predict_region_fn=_fake_predictor(vol))
    assert out is None                                                # the layer is the output: nothing is stitched
    layer = tmp_path / "pred"
    info = json.loads((layer / "info").read_text())
    assert info["type"] == "image" and info["data_type"] == "uint8" and info["num_channels"] == 2
    assert info["scales"] == [{"key": "8_8_30", "size": [10, 12, 8], "resolution": [8, 8, 30], "voxel_offset": [0, 0, 0],
                               "chunk_sizes": [[5, 3, 2]], "encoding": "raw"}]
    want = np.clip(np.stack([vol, vol * 2 + 1]) * np.float32(255.0), 0, 255).astype(np.uint8)
    # independent decode of one storage chunk: x 5..10, y 3..6, z 2..4 -> gzip'd bytes, Fortran order over (x, y, z, channel)
    raw = gzip.decompress((layer / "8_8_30" / "5-10_3-6_2-4.gz").read_bytes())
    cell = np.frombuffer(raw, dtype=np.uint8).reshape((5, 3, 2, 2), order="F")
    assert np.array_equal(cell.transpose(3, 2, 1, 0), want[:, 2:4, 3:6, 5:10])
    assert len(list((layer / "8_8_30").glob("*.gz"))) == 2 * 4 * 4
    got = PrecomputedLayer(layer).read_czyx((0, 0, 0), (8, 12, 10), fill_missing=False)
    assert got.dtype == np.uint8 and np.array_equal(got, want)
    assert np.array_equal(PrecomputedLayer(layer).read_czyx((1, 2, 3), (7, 11, 9)), want[:, 1:7, 2:11, 3:9])
    markers = sorted(p.name for p in (tmp_path / "pred.h5.chunks").glob("chunk_*.done"))
    assert markers == ["chunk_z0_y0_x0.done", "chunk_z0_y1_x0.done", "chunk_z1_y0_x0.done", "chunk_z1_y1_x0.done"]
    m = json.loads((tmp_path / "pred.h5.chunks" / "chunk_z1_y1_x0.done").read_text())
    assert m == {"chunk_key": "z1_y1_x0", "chunk_start_zyx": [4, 6, 0], "chunk_stop_zyx": [8, 12, 10],
                 "written_xyz": [[0, 6, 4], [10, 12, 8]]}

    def boom(*_a):
        raise AssertionError("chunk recomputed on resume")
    assert run_chunked_prediction_inference(cfg, None, vol, output_path=tmp_path / "pred.h5", predict_region_fn=boom) is None
    # ABISS convention: edge shift along each channel's own axis (read from the halo) + channel reversal, on 3-channel affinities
    aff = np.random.default_rng(6).random((3, 8, 12, 10)).astype(np.float32)
    cfg3 = _cfg((4, 6, 10), halo=(1, 1, 1))
    c3 = cfg3.inference.chunking
    c3.precomputed, c3.precomputed_resolution, c3.precomputed_chunk_size = True, [8, 8, 30], [10, 6, 4]
    c3.precomputed_affinity_convention = "abiss"

    def predict_aff(start, stop):
        return aff[(slice(None),) + tuple(slice(a, b) for a, b in zip(start, stop))][None]
    run_chunked_prediction_inference(cfg3, None, vol, output_path=tmp_path / "aff.h5", predict_region_fn=predict_aff)
    shifted = np.zeros_like(aff)
    shifted[0, 1:] = aff[0, :-1]
    shifted[1, :, 1:] = aff[1, :, :-1]
    shifted[2, :, :, 1:] = aff[2, :, :, :-1]
    got3 = PrecomputedLayer(tmp_path / "aff").read_czyx((0, 0, 0), (8, 12, 10), fill_missing=False)
    assert got3.dtype == np.float32 and np.array_equal(got3, shifted[::-1])
    # configuration errors, with the reference's messages
    c3.precomputed_affinity_convention = "seung"
    with pytest.raises(ValueError, match="must be 'none' or 'abiss'"):
        run_chunked_prediction_inference(cfg3, None, vol, output_path=tmp_path / "e.h5", predict_region_fn=boom)
    c3.precomputed_affinity_convention, c3.precomputed_resolution = "none", None
    with pytest.raises(ValueError, match="requires inference.chunking.precomputed_resolution"):
        run_chunked_prediction_inference(cfg3, None, vol, output_path=tmp_path / "e.h5", predict_region_fn=boom)
    c3.precomputed_resolution, c3.precomputed_chunk_size = [8, 8, 30], [4, 6, 4]
    with pytest.raises(ValueError, match="x: inference chunk 10 is not a multiple of storage chunk 4"):
        run_chunked_prediction_inference(cfg3, None, vol, output_path=tmp_path / "e.h5", predict_region_fn=boom)
    validate_precomputed_alignment((4, 6, 10), (5, 3, 2))
    with pytest.raises(ValueError, match="does not cover storage chunk"):
        PrecomputedLayer(layer).write_czyx((0, 0, 0), want[:, :3, :3, :5])


def test_roi_restricted_chunking_and_helpers(tmp_path):
    """inference.chunking.roi (chunked.py:217-272): chunks outside the ROI are dropped, straddling ones cropped, keys keep
    the global grid naming; plus the chunk_grid.py:78-111 resolvers."""
    from pytorch_connectomics_amd.inference.chunk_grid import (resolve_chunk_output_mode, resolve_h5_spatial_chunks,
                                                               validate_chunked_output_format)
    from pytorch_connectomics_amd.inference.chunked import (_filter_chunks_to_roi, _resolve_inference_roi,
                                                            _to_abiss_affinity_convention)
    cfg = _cfg((4, 5, 6))
    assert _resolve_inference_roi(cfg) is None
    cfg.inference.chunking.roi = [6, 7, 9]
    assert _resolve_inference_roi(cfg) == ((0, 0, 0), (6, 7, 9))
    cfg.inference.chunking.roi = [1, 2, 0, 6, 7, 8]
    roi = _resolve_inference_roi(cfg)
    assert roi == ((1, 2, 0), (6, 7, 8))
    for bad in ([1, 2], [3, 3, 3, 3, 9, 9]):
        cfg.inference.chunking.roi = bad
        with pytest.raises(ValueError, match="roi"):
            _resolve_inference_roi(cfg)
    chunks = build_chunk_grid((10, 13, 9), (4, 5, 6))
    kept = _filter_chunks_to_roi(chunks, roi, (0, 0, 0))
    assert [c.key for c in kept] == ["z0_y0_x0", "z0_y0_x1", "z0_y1_x0", "z0_y1_x1", "z1_y0_x0", "z1_y0_x1", "z1_y1_x0", "z1_y1_x1"]
    assert kept[0].start == (1, 2, 0) and kept[0].stop == (4, 5, 6) and kept[-1].start == (4, 5, 6) and kept[-1].stop == (6, 7, 8)
    # with a leading crop the ROI (input coordinates) is shifted into the cropped output space
    k2 = _filter_chunks_to_roi(build_chunk_grid((8, 13, 9), (4, 5, 6)), ((2, 0, 0), (6, 13, 9)), (2, 0, 0))
    assert {c.key for c in k2} == {f"z0_y{y}_x{x}" for y in range(3) for x in range(2)} and all(c.stop[0] == 4 for c in k2)
    vol = np.random.default_rng(4).random((10, 13, 9)).astype(np.float32)
    cfg.inference.chunking.roi = [1, 2, 0, 6, 7, 8]
    out = run_chunked_prediction_inference(cfg, None, vol, output_path=tmp_path / "r.npy", predict_region_fn=_fake_predictor(vol))
    want = np.zeros((2, 10, 13, 9), np.float32)
    want[:, 1:6, 2:7, 0:8] = np.stack([vol, vol * 2 + 1])[:, 1:6, 2:7, 0:8]
    np.testing.assert_array_equal(out, want)
    assert len(list((tmp_path / "r.npy.chunks").glob("chunk_*.npy"))) == 8
    # resolvers
    assert resolve_h5_spatial_chunks((10, 200, 64)) == (10, 64, 64)
    cfg.inference.chunking.output_mode = "raw_prediction"
    assert resolve_chunk_output_mode(cfg) == "raw_prediction"
    cfg.inference.chunking.output_mode = "labels"
    with pytest.raises(ValueError, match="output_mode"):
        resolve_chunk_output_mode(cfg)
    validate_chunked_output_format(cfg)
    # ABISS convention: dst[c, v] = src[c, v-1] along spatial axis c, then channels reversed
    a = np.arange(3 * 2 * 3 * 4, dtype=np.float32).reshape(3, 2, 3, 4)
    b = _to_abiss_affinity_convention(a)
    assert np.array_equal(b[2][1:], a[0][:-1]) and np.all(b[2][0] == 0)          # z-affinity: shifted in z, now channel 2
    assert np.array_equal(b[0][:, :, 1:], a[2][:, :, :-1]) and np.all(b[0][:, :, 0] == 0)
    with pytest.raises(ValueError, match="3-channel"):
        _to_abiss_affinity_convention(a[:2])
