"""Multi-rank helpers of window-sharded / view-sharded single-volume inference -- counterpart of the reference's
connectomics/inference/lazy_distributed.py (:10-169) and of the reduction half of its TTAPredictor (tta.py:262-310,
:1341-1560).

MI355X design.  The reference keeps its accumulators on the host and stages every 128 MB piece CPU -> GPU -> NCCL ->
GPU -> CPU.  Here the value / weight accumulators (and the TTA ensemble) already live in HBM, so a reduction is an
in-place RCCL `reduce` on views of the resident buffer: no staging copies, and the pieces are queued back to back on
the stream.  `distributed_reduce_chunk_mb` is still honoured (it bounds the size of one collective, which is what a
ring over point-to-point xGMI links is priced by), and only the bytes that need a given reduction op travel: a mixed
mean / min / max ensemble reduces each contiguous channel group once with its own op instead of the whole tensor once
per op.

All functions take the tensor's own device for their small metadata exchanges, so the same code runs over RCCL (HBM
tensors) and, in the CPU test-suite, over gloo."""
from __future__ import annotations

from typing import Optional, Sequence

import torch

__all__ = ["reduce_cpu_tensor_to_rank_zero", "distributed_context", "is_distributed_window_sharding_enabled", "distributed_reduction_device",
           "validate_distributed_tensor_shape", "reduce_tensor_to_rank_zero", "validate_distributed_patch_shard",
           "make_accumulator_reduce_hook", "shard_indices", "validate_view_shards", "reduce_view_ensemble"]

_MAX_NDIM = 8


def distributed_context() -> tuple[bool, int, int]:
    """(is_distributed, rank, world_size) -- lazy_distributed.py:10-13."""
    if not torch.distributed.is_available() or not torch.distributed.is_initialized():
        return False, 0, 1
    return True, torch.distributed.get_rank(), torch.distributed.get_world_size()


def is_distributed_window_sharding_enabled(cfg) -> bool:
    """inference.sliding_window.distributed_sharding AND a lazy data path (use_lazy_zarr / use_lazy_h5) AND world > 1
    (lazy_distributed.py:16-31; eager test data must never trip this gate -- manager.py:87-110)."""
    sw = getattr(getattr(cfg, "inference", None), "sliding_window", None)
    if sw is None:
        return False
    is_dist, _rank, world = distributed_context()
    dl = getattr(getattr(cfg, "data", None), "dataloader", None)
    lazy = bool(getattr(dl, "use_lazy_zarr", False) or getattr(dl, "use_lazy_h5", False))
    return bool(lazy and getattr(sw, "distributed_sharding", False) and is_dist and world > 1)


def distributed_reduction_device(infer_device) -> torch.device:
    """Where the reduction runs: the device the accumulators are on (they are never staged through the host here)."""
    return torch.device(infer_device)


def shard_indices(count: int, rank: int, world: int) -> list[int]:
    """The reference's interleaved ownership `[rank::world]` (lazy.py:1104-1110, tta.py:771-792)."""
    return list(range(int(count)))[int(rank)::int(world)]


def validate_distributed_tensor_shape(tensor: torch.Tensor, *, name: str, reduction_device=None) -> None:
    """Every rank must reduce the same shape: all_gather of a (1 + 8) int64 shape vector (lazy_distributed.py:42-75).  The shape
    vector travels on the tensor's own device; `reduction_device` (the reference's keyword) names where a HOST tensor's vector
    goes instead."""
    is_dist, _rank, world = distributed_context()
    if not is_dist:
        return
    if tensor.ndim > _MAX_NDIM:
        raise RuntimeError(f"{name} has rank {tensor.ndim}, exceeding supported rank {_MAX_NDIM}.")
    info = torch.full((_MAX_NDIM + 1,), -1, dtype=torch.int64)
    info[0] = tensor.ndim
    for i, d in enumerate(tensor.shape):
        info[i + 1] = int(d)
    info = info.to(tensor.device if (tensor.is_cuda or reduction_device is None) else torch.device(reduction_device))
    gathered = [torch.empty_like(info) for _ in range(world)]
    torch.distributed.all_gather(gathered, info)
    host = torch.stack(gathered).cpu().tolist()           # one device -> host copy for all ranks' vectors
    shapes = [tuple(row[1:1 + row[0]]) for row in host]
    if any(s != shapes[0] for s in shapes[1:]):
        summary = ", ".join(f"rank {r}: {s}" for r, s in enumerate(shapes))
        raise RuntimeError(f"Distributed lazy sliding-window sharding requires every rank to reduce {name} "
                           f"with the same shape, got {summary}.")


def reduce_tensor_to_rank_zero(tensor: torch.Tensor, *, op, chunk_mb: int, name: str,
                               validate: bool = True) -> Optional[torch.Tensor]:
    """Reduce `tensor` onto rank 0 IN PLACE, one collective per `chunk_mb` piece of its flat view
    (lazy_distributed.py:78-107 without the host staging).  Returns the tensor on rank 0 and None elsewhere (the buffer
    of a non-root rank is unspecified afterwards)."""
    is_dist, rank, _world = distributed_context()
    if not is_dist:
        return tensor
    if validate:
        validate_distributed_tensor_shape(tensor, name=name)
    if not tensor.is_contiguous():
        raise ValueError(f"{name} must be contiguous to be reduced in place")
    flat = tensor.view(-1)
    per = max(1, (max(1, int(chunk_mb or 128)) * 1024 * 1024) // max(1, flat.element_size()))
    for s in range(0, flat.numel(), per):
        torch.distributed.reduce(flat[s:s + per], dst=0, op=op)
    return tensor if rank == 0 else None


def reduce_cpu_tensor_to_rank_zero(tensor: torch.Tensor, *, op, reduction_device=None, chunk_mb: int = 128,
                                   name: str = "tensor") -> Optional[torch.Tensor]:
    """The reference's entry point by its own name and signature (lazy_distributed.py:78-107): reduce an accumulator onto rank 0 in
    `chunk_mb` pieces; rank 0 gets the reduced tensor, every other rank None.  A device tensor is reduced in place where it lives
    (`reduce_tensor_to_rank_zero`; `reduction_device` is then irrelevant); a HOST tensor -- the reference's case -- is staged
    through `reduction_device` piece by piece exactly as the reference does, for callers that keep their accumulators on the CPU."""
    is_dist, rank, _world = distributed_context()
    if not is_dist:
        return tensor
    if tensor.is_cuda:
        return reduce_tensor_to_rank_zero(tensor, op=op, chunk_mb=chunk_mb, name=name)
    validate_distributed_tensor_shape(tensor, name=name)
    if reduction_device is not None:
        dev = torch.device(reduction_device)
    else:       # the reference passes the inference device; without one: this rank's GPU, or the host under a CPU backend (gloo)
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    flat = tensor.contiguous().view(-1)
    per = max(1, (max(1, int(chunk_mb or 128)) * 1024 * 1024) // max(1, flat.element_size()))
    out = torch.empty_like(flat) if rank == 0 else None
    for s in range(0, flat.numel(), per):
        piece = flat[s:s + per].to(device=dev)
        torch.distributed.reduce(piece, dst=0, op=op)
        if rank == 0:
            out[s:s + per].copy_(piece.cpu())
    return out.view_as(tensor) if rank == 0 else None


def validate_distributed_patch_shard(*, local_count: int, total_count: int, device=None, reduction_device=None) -> None:
    """all_gather of the per-rank window counts; ANY empty shard fails on EVERY rank with the same message
    (lazy_distributed.py:110-129), so no rank is left waiting in the reduction."""
    is_dist, _rank, world = distributed_context()
    if not is_dist:
        return
    count = torch.tensor([int(local_count)], dtype=torch.int64, device=device if device is not None else reduction_device)
    gathered = [torch.empty_like(count) for _ in range(world)]
    torch.distributed.all_gather(gathered, count)
    counts = [int(v) for v in torch.cat(gathered).cpu().tolist()]
    if any(v <= 0 for v in counts):
        raise RuntimeError("Distributed lazy sliding-window sharding assigned an empty window shard "
                           f"(total_windows={total_count}, per_rank={counts}). Use fewer GPUs or a "
                           "smaller inference.sliding_window.window_size.")


def make_accumulator_reduce_hook(*, chunk_mb: int, reduction_device=None):
    """hook(value, weight) -> (value, weight) on rank 0, None on every other rank -- never (None, None)
    (lazy_distributed.py:132-169, same keywords).  The accumulators go through `reduce_cpu_tensor_to_rank_zero`, looked up when
    the hook runs: HBM-resident ones are reduced in place where they live (`reduction_device` does not matter then), host
    tensors are staged through `reduction_device` like the reference's."""
    def _hook(value: torch.Tensor, weight: torch.Tensor):
        pair = []
        for tensor, what in ((value, "value accumulator"), (weight, "weight accumulator")):
            pair.append(reduce_cpu_tensor_to_rank_zero(tensor, op=torch.distributed.ReduceOp.SUM, reduction_device=reduction_device,
                                                       chunk_mb=chunk_mb, name=what))
        return None if any(t is None for t in pair) else tuple(pair)
    return _hook


# ---------------------------------------------------------------------------------------------- TTA view sharding
def validate_view_shards(total_views: int) -> tuple[int, int, list[int]]:
    """(rank, world, local view indices).  Every rank knows the full view list, so an empty shard is detected without a
    collective and raised on ALL ranks (the reference raises only on the starved rank, tta.py:779-786, and leaves the
    others blocked in the reduce)."""
    _is_dist, rank, world = distributed_context()
    if total_views < world:
        raise RuntimeError("Distributed TTA sharding produced an empty augmentation shard for "
                           f"ranks >= {total_views} (views={total_views}, world_size={world}). Reduce the GPU count or increase TTA variants.")
    return rank, world, shard_indices(total_views, rank, world)


def _channel_groups(modes: Sequence[str]):
    i = 0
    while i < len(modes):
        j = i + 1
        while j < len(modes) and modes[j] == modes[i]:
            j += 1
        yield i, j, modes[i]
        i = j


def reduce_view_ensemble(acc: torch.Tensor, n_local: int, total_views: int, modes: Sequence[str], *, chunk_mb: int = 128,
                         skip_channels: Sequence[int] = (), stats: Optional[torch.Tensor] = None,
                         counts: Optional[torch.Tensor] = None, partial_modes: Sequence[str] = ()):
    """Combine the per-rank view ensembles on rank 0 (tta.py:1341-1519).

    `acc` (1, C, Z, Y, X) holds this rank's ensemble of its `n_local` views: the running MEAN of mean-channels, the
    min / max of the others.  Mean groups are rescaled to sums, reduced with SUM and divided by `total_views`; min / max
    groups with MIN / MAX.  Each contiguous same-mode channel group is one slab of the buffer and is reduced once.
    `skip_channels` (the partially valid affinity channels) are carried by `stats` / `counts` (P, Z, Y, X) instead:
    statistics with their mode's op, counts with SUM; the caller finalises them on rank 0.
    Returns (acc, stats, counts) on rank 0 and None elsewhere."""
    R = torch.distributed.ReduceOp
    ops_by_mode = {"mean": R.SUM, "min": R.MIN, "max": R.MAX}
    bad = sorted(set(modes) - set(ops_by_mode))
    if bad:
        raise ValueError(f"Unknown TTA ensemble modes: {bad}.")
    _is_dist, rank, _world = distributed_context()
    validate_distributed_tensor_shape(acc, name="TTA ensemble")
    if acc.shape[0] != 1 or not acc.is_contiguous():
        raise ValueError("reduce_view_ensemble expects a contiguous (1, C, ...) ensemble")
    skip = set(int(c) for c in skip_channels)
    for a, b, mode in _channel_groups(list(modes)):
        if all(c in skip for c in range(a, b)):
            continue
        slab = acc[0, a:b]
        if mode == "mean":
            slab.mul_(float(n_local))
        reduce_tensor_to_rank_zero(slab, op=ops_by_mode[mode], chunk_mb=chunk_mb, name="TTA ensemble", validate=False)
        if mode == "mean" and rank == 0:
            slab.div_(float(total_views))
    if stats is not None and stats.numel():
        for pi, mode in enumerate(partial_modes):
            reduce_tensor_to_rank_zero(stats[pi], op=ops_by_mode[mode], chunk_mb=chunk_mb, name="TTA partial statistics", validate=False)
        reduce_tensor_to_rank_zero(counts, op=R.SUM, chunk_mb=chunk_mb, name="TTA partial counts", validate=False)
    return (acc, stats, counts) if rank == 0 else None
