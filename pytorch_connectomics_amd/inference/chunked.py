"""Chunk-grid inference with exact seams -- counterpart of the reference's connectomics/inference/chunked.py
(run_chunked_prediction_inference :725-944, _run_chunked_prediction_per_rank :437-722,
_stitch_chunk_prediction_files :317-434, _write_chunk_index :279-314) and inference/chunk_grid.py:90-111.

Every chunk is predicted as `lazy_predict_region(core +- halo)` from the GLOBAL window grid and cropped to
its core, so the stitched result equals the whole-volume prediction bit for bit (the reference asserts the
same, tests/unit/test_chunked_inference.py:177).  Chunks are dealt to ranks `idx % world == rank`
(chunked.py:471), each rank writes its own disjoint files, one barrier, rank 0 writes the index and
stitches -- no data-path collective.  Files are .npy (C,Z,Y,X) because h5py is not part of this image; the
naming (chunk_{zI_yJ_xK}) and index.json layout follow the reference.
"""
from __future__ import annotations

import json
import logging
import os
from pathlib import Path
from typing import Callable, Optional, Sequence

import numpy as np
import torch

from ..chunked import ChunkRef, ResumeManifest, build_chunk_grid, resolve_halo_region
from .crop import cropped_shape, resolve_global_prediction_crop
from .lazy import get_lazy_image_reference_shape, lazy_predict_region

logger = logging.getLogger(__name__)


def is_chunked_inference_enabled(cfg) -> bool:
    inf = getattr(cfg, "inference", None)
    ch = getattr(inf, "chunking", None)
    strategy = getattr(inf, "strategy", None) or getattr(getattr(inf, "execution", None), "strategy", None)
    return bool(getattr(ch, "enabled", False) or strategy == "chunked")


def resolve_chunk_shape(cfg, volume_shape: Sequence[int]) -> tuple[int, int, int]:
    """chunking.chunk_size, with axes == 'z' keeping full YX (reference inference/chunk_grid.py:90-101)."""
    ch = getattr(getattr(cfg, "inference", None), "chunking", None)
    size = getattr(ch, "chunk_size", None)
    if not size:
        raise ValueError("inference.chunking.chunk_size must be set for chunked inference")
    size = [int(v) for v in size]
    if len(size) != 3 or any(v <= 0 for v in size):
        raise ValueError(f"inference.chunking.chunk_size must be 3 positive ints, got {size}")
    axes = str(getattr(ch, "axes", "all")).lower()
    if axes == "z":
        size = [size[0], int(volume_shape[1]), int(volume_shape[2])]
    elif axes != "all":
        raise ValueError(f"inference.chunking.axes must be 'all' or 'z', got {axes!r}")
    return tuple(min(s, int(v)) for s, v in zip(size, volume_shape))


def _rank_world() -> tuple[int, int]:
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        return torch.distributed.get_rank(), torch.distributed.get_world_size()
    return 0, 1


def _resolve_external_chunk_shard(cfg) -> Optional[tuple[int, int]]:
    ch = getattr(getattr(cfg, "inference", None), "chunking", None)
    sid, n = getattr(ch, "shard_id", None), getattr(ch, "num_shards", None)
    if sid is None and n is None:
        return None
    if sid is None or n is None or int(n) <= 0 or not 0 <= int(sid) < int(n):
        raise ValueError(f"inference.chunking.shard_id/num_shards invalid: shard_id={sid}, num_shards={n}")
    return int(sid), int(n)


def _chunks_dir(output_path: Path) -> Path:
    return output_path.with_suffix(output_path.suffix + ".chunks")


def _chunk_file(chunks_dir: Path, chunk: ChunkRef) -> Path:
    return chunks_dir / f"chunk_{chunk.key}.npy"


def _write_chunk_index(output_path: Path, chunks, chunks_dir: Path, volume_shape, chunk_shape, halo, channels: int):
    index = {"volume_shape": [int(v) for v in volume_shape], "chunk_shape": list(chunk_shape), "halo": list(halo),
             "channels": int(channels),
             "chunks": [{"key": c.key, "index": list(c.index), "start": list(c.start), "stop": list(c.stop),
                         "path": str(_chunk_file(chunks_dir, c).relative_to(output_path.parent))} for c in chunks]}
    p = output_path.with_suffix(output_path.suffix + ".index.json")
    tmp = p.with_suffix(p.suffix + ".tmp")
    tmp.write_text(json.dumps(index, indent=2))
    os.replace(tmp, p)
    return p


def stitch_chunk_prediction_files(output_path, chunks, volume_shape, *, dtype=None) -> np.ndarray:
    """Assemble per-chunk files into one (C,Z,Y,X) array (z-slab streaming is unnecessary here: the result is
    returned in host memory and also saved as <output_path>)."""
    output_path = Path(output_path)
    cdir = _chunks_dir(output_path)
    first = _chunk_file(cdir, chunks[0])
    if not first.exists():
        raise FileNotFoundError(f"Missing first chunk prediction file: {first}")
    head = np.load(first, mmap_mode="r")
    out = np.zeros((head.shape[0],) + tuple(int(v) for v in volume_shape), dtype=dtype or head.dtype)
    for i, c in enumerate(chunks, 1):
        f = _chunk_file(cdir, c)
        if not f.exists():
            raise FileNotFoundError(f"Missing chunk prediction file {i}/{len(chunks)}: {f}")
        arr = np.load(f)
        if arr.shape[0] != out.shape[0]:
            raise ValueError(f"Chunk {c.key} channel mismatch: {arr.shape[0]} vs {out.shape[0]}")
        if tuple(arr.shape[1:]) != c.shape:
            raise ValueError(f"Chunk {c.key} spatial shape mismatch: {tuple(arr.shape[1:])} vs {c.shape}")
        out[(slice(None),) + c.slices] = arr
    np.save(output_path, out)
    return out


class _ChunkWriter:
    """One background thread that turns (tensor | array, path, key) jobs into chunk files.  CUDA tensors are copied into
    pinned host memory with a non-blocking copy; the thread synchronises on the copy's event, not on the device."""

    def __init__(self, manifest, depth: int = 2):
        import queue
        import threading
        self.manifest = manifest
        self.q = queue.Queue(maxsize=depth)          # bounds the pinned memory in flight
        self.err = None
        self.t = threading.Thread(target=self._run, name="pytc-chunk-writer", daemon=True)
        self.t.start()

    def submit(self, core_pred, path: Path, key: str) -> None:
        if self.err is not None:
            raise self.err
        if isinstance(core_pred, torch.Tensor) and core_pred.is_cuda:
            src = core_pred.detach().float()
            host = torch.empty(src.shape, dtype=torch.float32, pin_memory=True)
            host.copy_(src, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self.q.put((host, ev, path, key))
        else:
            arr = core_pred.detach().float().cpu().numpy() if isinstance(core_pred, torch.Tensor) else np.asarray(core_pred)
            self.q.put((arr, None, path, key))

    def _run(self):
        while True:
            job = self.q.get()
            if job is None:
                return
            if self.err is not None:
                continue
            data, ev, path, key = job
            try:
                if ev is not None:
                    ev.synchronize()
                    data = data.numpy()
                tmp = path.with_suffix(".tmp.npy")
                np.save(tmp, data)
                os.replace(tmp, path)
                self.manifest.mark_completed(key)
            except BaseException as e:          # surfaced to the producer at the next submit / close
                self.err = e

    def close(self) -> None:
        self.q.put(None)
        self.t.join()
        if self.err is not None:
            raise self.err


def run_chunked_prediction_inference(cfg, forward_fn, volume, *, output_path, device="cuda",
                                     requested_head: Optional[str] = None,
                                     predict_region_fn: Optional[Callable] = None, stitch: bool = True,
                                     overwrite: bool = False):
    """Predict `volume` chunk by chunk.  Returns the stitched (C,Z,Y,X) numpy array on rank 0 (None elsewhere,
    or when stitch=False / external sharding is active).  `predict_region_fn(start, stop) -> (1,C,*region)`
    replaces the device predictor in host-logic tests."""
    output_path = Path(output_path)
    if output_path.suffix != ".npy":
        output_path = output_path.with_suffix(output_path.suffix + ".npy")
    input_shape = get_lazy_image_reference_shape(volume)
    # the chunk grid lives in the CROPPED output space (user crop_pad + DeepEM affinity border, reference
    # chunked.py:743-755); a chunk's core in input coordinates is shifted by the leading crop
    crop_pad = resolve_global_prediction_crop(cfg)
    crop_before = tuple(int(crop_pad[a][0]) for a in range(3))
    vol_shape = cropped_shape(input_shape[-3:], crop_pad)
    ch_cfg = getattr(getattr(cfg, "inference", None), "chunking", None)
    halo = tuple(int(v) for v in (getattr(ch_cfg, "halo", None) or (0, 0, 0)))
    chunk_shape = resolve_chunk_shape(cfg, vol_shape)
    chunks = build_chunk_grid(vol_shape, chunk_shape)
    cdir = _chunks_dir(output_path)
    cdir.mkdir(parents=True, exist_ok=True)
    rank, world = _rank_world()
    ext = _resolve_external_chunk_shard(cfg)
    if ext is not None:
        mine = [(i, c) for i, c in enumerate(chunks) if i % ext[1] == ext[0]]
    else:
        mine = [(i, c) for i, c in enumerate(chunks) if i % world == rank]
    manifest = ResumeManifest.load_or_create(cdir / f"manifest_rank{rank if ext is None else ext[0]}.json",
                                             {"chunk_shape": list(chunk_shape), "output_shape": list(vol_shape),
                                              "halo": list(halo), "crop_pad": [list(p) for p in crop_pad]},
                                             overwrite=overwrite)
    if predict_region_fn is None:
        def predict_region_fn(start, stop):
            return lazy_predict_region(cfg, forward_fn, volume, region_start=start, region_stop=stop, device=device,
                                       requested_head=requested_head)
    channels = None
    # device -> pinned host copy and the file write of chunk i overlap the prediction of chunk i+1: the copy is issued
    # non-blocking behind the chunk's kernels, a writer thread waits on its event, saves atomically (tmp + rename) and
    # only then marks the chunk completed in the manifest
    writer = _ChunkWriter(manifest)
    try:
        for pos, (idx, c) in enumerate(mine, 1):
            f = _chunk_file(cdir, c)
            if f.exists() and c.key in manifest.completed:     # idempotent resume (reference chunked.py:510-523)
                logger.info("chunk %s already done, skipping", c.key)
                continue
            read_lo, read_hi, core = resolve_halo_region(c, input_shape[-3:], halo=halo, crop_before=crop_before)
            pred = predict_region_fn(read_lo, read_hi)
            core_pred = pred[(0, slice(None)) + core]
            channels = int(core_pred.shape[0])
            writer.submit(core_pred, f, c.key)
    finally:
        writer.close()
    if ext is not None:
        return None       # external shards are stitched by a later call once every shard has run
    if world > 1:
        torch.distributed.barrier()
    if rank != 0:
        return None
    if channels is None:
        channels = int(np.load(_chunk_file(cdir, chunks[0]), mmap_mode="r").shape[0])
    _write_chunk_index(output_path, chunks, cdir, vol_shape, chunk_shape, halo, channels)
    return stitch_chunk_prediction_files(output_path, chunks, vol_shape) if stitch else None


__all__ = ["run_chunked_prediction_inference", "stitch_chunk_prediction_files", "is_chunked_inference_enabled",
           "resolve_chunk_shape"]
