"""World-2 / world-3 gloo tests of the multi-rank single-volume inference protocols (reference
inference/lazy_distributed.py:10-169, lazy.py:1104-1110, tta.py:771-792, :1341-1519): window sharding of
`lazy_predict_volume`, TTA-view sharding of `TTAPredictor.predict`, the empty-shard / shape checks and the chunked
in-place reduce.

The product has no CPU path, so inside these worker processes the kernel module the engines call (`hip_ops`) is replaced
by small torch stand-ins defined HERE (test infrastructure): everything above the kernels -- shard ownership, validation,
collectives, which rank returns what -- is the product code, running over gloo exactly as it runs over RCCL."""
import os
import socket
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import window_oracle as WO


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


# --------------------------------------------------------------------------------------------------- kernel stand-ins
class _CpuOps:
    """torch restatements of the few kernels the lazy / TTA host code calls (signatures of pytorch_connectomics_amd.hip_ops)."""

    @staticmethod
    def require_device(device, what=""):
        return None

    @staticmethod
    def gather_windows(vol, starts, roi, pad_mode="constant", cval=0.0):
        assert pad_mode == "constant"                     # the lazy grid overhangs the volume: constant outer padding here
        C, ext = vol.shape[0], vol.shape[1:]
        out = torch.full((len(starts),) + tuple(roi) + (C,), float(cval))
        for i, s in enumerate(starts):
            lo = [max(0, s[a]) for a in range(3)]
            hi = [min(ext[a], s[a] + roi[a]) for a in range(3)]
            src = vol[:, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]].permute(1, 2, 3, 0)
            out[i][tuple(slice(lo[a] - s[a], hi[a] - s[a]) for a in range(3))] = src
        return out

    @staticmethod
    def blend_accumulate(pred, starts, value, weight, wz, wy, wx, *, view=0, combine=0, floor_w=1e-5, border=None):
        from pytorch_connectomics_amd.inference.window import _combine_axes
        w = _combine_axes([wz, wy, wx], combine, floor_w, "cpu", torch.float32)
        assert view == 0 and not border
        ext, roi = value.shape[1:], pred.shape[1:4]
        for i, s in enumerate(starts):                    # only the part of a window inside the accumulator lands
            lo = [max(0, s[a]) for a in range(3)]
            hi = [min(ext[a], s[a] + roi[a]) for a in range(3)]
            dst = tuple(slice(lo[a], hi[a]) for a in range(3))
            src = tuple(slice(lo[a] - s[a], hi[a] - s[a]) for a in range(3))
            value[(slice(None),) + dst] += (pred[i].permute(3, 0, 1, 2) * w)[(slice(None),) + src]
            if weight is not None:
                weight[dst] += w[src]

    @staticmethod
    def blend_finalize(value, weight, clamp=1e-4, act=0):
        value /= weight.clamp_min(clamp)

    @staticmethod
    def ensemble_update(acc, pred, mode, count):
        from pytorch_connectomics_amd.inference.tta import _MODE_CODE
        if mode == _MODE_CODE["mean"]:
            acc += (pred - acc) / float(count)
        elif mode == _MODE_CODE["min"]:
            torch.minimum(acc, pred, out=acc)
        else:
            torch.maximum(acc, pred, out=acc)


class _WholeImageEngine:
    """Stand-in for the device sliding engine at one window = the whole image: `accumulate` returns the prediction of a
    view mapped back to the canonical frame (what the gather / blend kernels' view index math produces) and a unit weight."""
    cval = 0.0

    def __init__(self, roi):
        self.roi_size = tuple(roi)

    def accumulate(self, vol, network, view=0, weight=None, add_weight=True, chan_map=None):
        from pytorch_connectomics_amd import _native as nat
        assert not view & nat.VIEW_SWAP_YX, "flip views only in this test"
        dims = [d + 2 for d, bit in enumerate((nat.VIEW_FLIP_Z, nat.VIEW_FLIP_Y, nat.VIEW_FLIP_X)) if view & bit]
        x = vol.unsqueeze(0)
        y = network(torch.flip(x, dims) if dims else x)
        y = torch.flip(y, dims) if dims else y
        return y[0].contiguous().clone(), (torch.ones(vol.shape[1:]) if weight is None else weight)


def _net(x):
    """Closed-form, not flip-equivariant, 3 output channels."""
    z = torch.linspace(-1, 1, x.shape[2]).view(1, 1, -1, 1, 1)
    y = torch.linspace(-1, 1, x.shape[3]).view(1, 1, 1, -1, 1)
    w = torch.linspace(-1, 1, x.shape[4]).view(1, 1, 1, 1, -1)
    return torch.cat([x * (1.0 + 0.5 * w) + 0.25 * y, torch.tanh(2 * x - 1) * z + 0.1 * w * y, 3 * x * x - 1.5 * w + z * y], 1)


def _lazy_cfg(roi, *, sharded=True, lazy=True, tta_sharding=False):
    return NS(model=NS(primary_head=None, heads=None, out_channels=3, output_size=None),
              data=NS(dataloader=NS(batch_size=1, use_lazy_h5=lazy, use_lazy_zarr=False)),
              inference=NS(sliding_window=NS(window_size=list(roi), sw_batch_size=2, overlap=0.5, blending="bump",
                                             padding_mode="constant", cval=0.0, snap_to_edge=False, target_context=None,
                                             border_mask=None, distributed_sharding=sharded, distributed_reduce_chunk_mb=1),
                           model=NS(head=None, select_channel=None, output_dtype=None, channel_activations=None, crop_pad=None),
                           test_time_augmentation=NS(enabled=tta_sharding, distributed_sharding=tta_sharding)))


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)


def _patch_ops():
    import pytorch_connectomics_amd.inference.lazy as lazy
    import pytorch_connectomics_amd.inference.tta as tta
    lazy.ops = _CpuOps
    tta.ops = _CpuOps
    return lazy, tta


# --------------------------------------------------------------------------------------------------- window sharding
def _window_worker(rank, world, port):
    _init(rank, world, port)
    lazy, _ = _patch_ops()
    from pytorch_connectomics_amd.inference import lazy_distributed as LD
    vol = np.random.default_rng(3).random((1, 12, 20, 28), dtype=np.float32)
    roi = (8, 8, 8)
    cfg = _lazy_cfg(roi)
    assert LD.is_distributed_window_sharding_enabled(cfg)
    assert not LD.is_distributed_window_sharding_enabled(_lazy_cfg(roi, lazy=False))      # eager data never shards
    assert not LD.is_distributed_window_sharding_enabled(_lazy_cfg(roi, sharded=False))
    out = lazy.lazy_predict_volume(cfg, _net, vol, device="cpu")
    if rank == 0:
        want = WO.lazy_sliding_window(vol, _net, roi=roi, overlap=0.5, mode="bump", sw_batch_size=2, padding_mode="constant")
        assert tuple(out.shape) == (1, 3, 12, 20, 28)
        assert torch.allclose(out, want, rtol=1e-5, atol=1e-6), float((out - want).abs().max())
    else:
        assert out.numel() == 0                                                              # contributors return nothing
    # the sharded result equals the unsharded run of the same code (reduction is a plain sum of disjoint window sets)
    solo = lazy.lazy_predict_volume(_lazy_cfg(roi, sharded=False), _net, vol, device="cpu")
    if rank == 0:
        assert torch.allclose(out, solo, rtol=1e-6, atol=1e-6)

    # one window for two ranks: the empty shard is reported on EVERY rank (no rank is left waiting in a reduce)
    with pytest.raises(RuntimeError, match=r"empty window shard \(total_windows=1, per_rank=\[1, 0\]\)"):
        lazy.lazy_predict_volume(_lazy_cfg((12, 20, 28)), _net, vol, device="cpu")

    # TTA-view sharding cannot be combined with window sharding (reference lazy.py:1039-1043)
    with pytest.raises(RuntimeError, match="does not support"):
        lazy.lazy_predict_volume(_lazy_cfg(roi, tta_sharding=True), _net, vol, device="cpu")

    # shape validation: every rank sees every rank's shape
    t = torch.zeros(2, 3 + rank)
    with pytest.raises(RuntimeError, match=r"rank 0: \(2, 3\), rank 1: \(2, 4\)"):
        LD.validate_distributed_tensor_shape(t, name="value accumulator")
    with pytest.raises(RuntimeError, match="exceeding supported rank"):
        LD.validate_distributed_tensor_shape(torch.zeros((1,) * 9), name="x")

    # chunked in-place reduce: 600k floats at 1 MB per collective = 3 pieces; SUM / MIN / MAX
    base = torch.arange(600_000, dtype=torch.float32)
    for op, want in ((torch.distributed.ReduceOp.SUM, base * 2 + 1), (torch.distributed.ReduceOp.MIN, base),
                     (torch.distributed.ReduceOp.MAX, base + 1)):
        buf = (base + rank).clone()
        calls = []
        real = torch.distributed.reduce
        torch.distributed.reduce = lambda t, dst, op, _r=real: (calls.append(t.numel()), _r(t, dst=dst, op=op))[1]
        try:
            got = LD.reduce_tensor_to_rank_zero(buf, op=op, chunk_mb=1, name="buf")
        finally:
            torch.distributed.reduce = real
        assert calls == [262144, 262144, 75712]
        if rank == 0:
            assert got is buf and torch.equal(got, want)                                   # in place, no staging copy
        else:
            assert got is None
    # the reference's own entry point (host accumulators staged through `reduction_device` in chunk_mb pieces)
    host = torch.arange(300_000, dtype=torch.float32) + rank
    got = LD.reduce_cpu_tensor_to_rank_zero(host, op=torch.distributed.ReduceOp.SUM, reduction_device=torch.device("cpu"), chunk_mb=1,
                                            name="host accumulator")
    assert (got is None) == (rank != 0)
    if rank == 0:
        assert torch.equal(got, 2 * torch.arange(300_000, dtype=torch.float32) + 1) and got.shape == host.shape
    with pytest.raises(ValueError, match="contiguous"):
        LD.reduce_tensor_to_rank_zero(torch.zeros(4, 4).t(), op=torch.distributed.ReduceOp.SUM, chunk_mb=1, name="v", validate=False)
    hook = LD.make_accumulator_reduce_hook(chunk_mb=1)
    res = hook(torch.full((2, 4), 1.0 + rank), torch.full((4,), 2.0))
    assert (res is None) == (rank != 0)
    if rank == 0:
        assert torch.equal(res[0], torch.full((2, 4), 3.0)) and torch.equal(res[1], torch.full((4,), 4.0))
    torch.distributed.destroy_process_group()


def test_window_sharding_world2():
    mp.spawn(_window_worker, args=(2, _free_port()), nprocs=2, join=True)


def _window_worker_world8(rank, world, port):
    """The reference's window sharding ([rank::world] + reduce to rank 0, lazy.py:1104, lazy_distributed.py:78-107) at the rank count
    of the target node, on the Lucchi++ window grid scaled to roi 8^3: 2 x 18 x 13 = 468 windows, 58 or 59 per rank."""
    _init(rank, world, port)
    lazy, _ = _patch_ops()
    vol = np.random.default_rng(8).random((1, 12, 76, 56), dtype=np.float32)
    roi = (8, 8, 8)
    seen = []

    def net(x):
        seen.append(int(x.shape[0]))
        return _net(x)

    out = lazy.lazy_predict_volume(_lazy_cfg(roi), net, vol, device="cpu")
    mine = torch.tensor([sum(seen)])
    counts = [torch.zeros(1, dtype=torch.long) for _ in range(world)]
    torch.distributed.all_gather(counts, mine)
    total = int(sum(int(c) for c in counts))
    # the lazy grid (reference lazy.py:104-214: its own stride / edge rule) has more windows than the eager 468; whatever their number,
    # the shards are the interleaved partition [rank::world] of it: every window predicted exactly once
    assert [int(c) for c in counts] == [len(range(total)[r::world]) for r in range(world)] and total >= 468
    if rank == 0:
        want = WO.lazy_sliding_window(vol, _net, roi=roi, overlap=0.5, mode="bump", sw_batch_size=2, padding_mode="constant")
        assert tuple(out.shape) == (1, 3, 12, 76, 56)
        assert torch.allclose(out, want, rtol=1e-5, atol=1e-5), float((out - want).abs().max())
    else:
        assert out.numel() == 0
    torch.distributed.destroy_process_group()


def test_window_sharding_world8_on_the_lucchi_window_grid():
    mp.spawn(_window_worker_world8, args=(8, _free_port()), nprocs=8, join=True)


# --------------------------------------------------------------------------------------------------- view sharding
def _tta_cfg(mode, *, sharded, flips="all"):
    return NS(model=NS(primary_head=None, heads=None, out_channels=3),
              data=NS(train=NS(do_2d=False), val=NS(do_2d=False), dataloader=NS(batch_size=1), label_transform=None),
              inference=NS(sliding_window=None,
                           model=NS(head=None, select_channel=None, output_dtype=None, channel_activations=None, crop_pad=None),
                           test_time_augmentation=NS(enabled=True, flip_axes=flips, rotation90_axes=None, rotate90_k=None,
                                                     ensemble_mode=mode, patch_first_local=True, distributed_sharding=sharded,
                                                     distributed_reduce_chunk_mb=1, apply_mask=True)))


def _predictor(cfg, roi):
    from pytorch_connectomics_amd.inference.tta import TTAPredictor
    p = TTAPredictor(cfg=cfg, sliding_inferer=_WholeImageEngine(roi), forward_fn=_net)
    p.apply_preprocessing = lambda t: t                  # activations / selection are device kernels, covered by the GPU tests
    p._engine_network = lambda: _net
    return p


def _view_worker(rank, world, port):
    _init(rank, world, port)
    _patch_ops()
    x = torch.rand(1, 1, 6, 10, 12, generator=torch.Generator().manual_seed(5))
    roi = tuple(x.shape[2:])
    for mode in ("mean", "min", [["0", "mean"], ["1", "max"], ["2", "min"]]):
        pred = _predictor(_tta_cfg(mode, sharded=True), roi)
        assert pred.is_distributed_sharding_enabled()
        out = pred.predict(x.clone())
        solo = _predictor(_tta_cfg(mode, sharded=False), roi).predict(x.clone())          # all 8 views on this rank
        if rank == 0:
            assert not pred.should_skip_postprocess_on_rank()
            assert tuple(out.shape) == (1, 3, 6, 10, 12)
            assert torch.allclose(out, solo, rtol=1e-5, atol=1e-6), (mode, float((out - solo).abs().max()))
        else:
            assert out.numel() == 0 and pred.should_skip_postprocess_on_rank()
    # the views of the two ranks are disjoint and interleaved [rank::world]
    from pytorch_connectomics_amd.inference.lazy_distributed import validate_view_shards
    assert validate_view_shards(8)[2] == list(range(8))[rank::2]
    torch.distributed.destroy_process_group()


def test_tta_view_sharding_world2():
    mp.spawn(_view_worker, args=(2, _free_port()), nprocs=2, join=True)


def _starved_worker(rank, world, port):
    _init(rank, world, port)
    _patch_ops()
    x = torch.rand(1, 1, 4, 6, 6)
    pred = _predictor(_tta_cfg("mean", sharded=True, flips=[[0]]), tuple(x.shape[2:]))   # identity + one flip = 2 views
    with pytest.raises(RuntimeError, match=r"empty augmentation shard .*views=2, world_size=3"):
        pred.predict(x)                                                                     # raised on all three ranks
    torch.distributed.destroy_process_group()


def test_tta_view_sharding_empty_shard_world3():
    mp.spawn(_starved_worker, args=(3, _free_port()), nprocs=3, join=True)


def _partial_worker(rank, world, port):
    """Partially valid (shifted affinity) channels travel as statistics + counts: mean -> SUM, min -> MIN, counts -> SUM."""
    _init(rank, world, port)
    from pytorch_connectomics_amd.inference.lazy_distributed import reduce_view_ensemble
    g = torch.Generator().manual_seed(11)
    views = torch.rand(6, 1, 3, 4, 5, 5, generator=g)                   # 6 views of a (1, 3, 4, 5, 5) prediction
    valid = torch.rand(6, 2, 4, 5, 5, generator=g) > 0.3                # validity of the two partial channels (1 and 2)
    valid[0] = True
    modes = ["max", "mean", "min"]
    mine = list(range(6))[rank::world]
    acc = views[mine[0]].clone()
    for v in mine[1:]:
        acc[0, 0] = torch.maximum(acc[0, 0], views[v][0, 0])
    stats = torch.stack([torch.zeros(4, 5, 5), torch.full((4, 5, 5), float("inf"))])
    counts = torch.zeros(2, 4, 5, 5)
    for v in mine:
        stats[0] += torch.where(valid[v, 0], views[v][0, 1], torch.zeros(()))
        stats[1] = torch.where(valid[v, 1], torch.minimum(stats[1], views[v][0, 2]), stats[1])
        counts += valid[v].float()
    red = reduce_view_ensemble(acc, len(mine), 6, modes, chunk_mb=1, skip_channels=[1, 2], stats=stats, counts=counts,
                               partial_modes=["mean", "min"])
    if rank != 0:
        assert red is None
    else:
        acc, stats, counts = red
        assert torch.equal(acc[0, 0], views[:, 0, 0].max(0).values)
        assert torch.equal(counts, valid.float().sum(0))
        want_mean = (views[:, 0, 1] * valid[:, 0]).sum(0)
        assert torch.allclose(stats[0], want_mean, atol=1e-6)
        want_min = torch.where(valid[:, 1], views[:, 0, 2], torch.full((), float("inf"))).min(0).values
        assert torch.equal(stats[1], want_min)
    torch.distributed.destroy_process_group()


def test_partial_channel_statistics_reduce_world2():
    mp.spawn(_partial_worker, args=(2, _free_port()), nprocs=2, join=True)
