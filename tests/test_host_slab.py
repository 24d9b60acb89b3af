"""Slab ownership + halo exchange: plan properties (window-balanced staircase plan and the whole-row plan) and
the exchange protocol over gloo (world 2, 3 and 8) with the CPU oracle as the accumulator, against the single-process
sliding-window oracle."""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import window_oracle as WO
from pytorch_connectomics_amd.inference.slab import (_lex_boxes, exchange_schedule, gather_slabs, plan_slabs, slab_predict)


def _net(x):            # closed form, position dependent inside the window, 2 output channels
    z = torch.linspace(-1, 1, x.shape[2]).view(1, 1, -1, 1, 1)
    w = torch.linspace(-1, 1, x.shape[4]).view(1, 1, 1, 1, -1)
    return torch.cat([x * (1 + 0.5 * w) + 0.1 * z, torch.tanh(2 * x - 1) - 0.2 * w * z], 1)


def _vol(box):
    return (box[0][1] - box[0][0]) * (box[1][1] - box[1][0]) * (box[2][1] - box[2][0])


def _mark(arr, box, inc=1):
    arr[box[0][0]:box[0][1], box[1][0]:box[1][1], box[2][0]:box[2][1]] += inc


@pytest.mark.parametrize("lo,hi,dims", [(0, 24, (2, 3, 4)), (5, 6, (2, 3, 4)), (3, 21, (2, 3, 4)), (7, 17, (4, 2, 3)), (0, 0, (2, 2, 2)),
                                         (11, 12, (1, 1, 13)), (1, 467, (18, 13, 2)), (58, 117, (18, 13, 2))])
def test_lexicographic_range_as_boxes(lo, hi, dims):
    boxes = _lex_boxes(lo, hi, dims)
    assert len(boxes) <= 2 * len(dims) - 1
    seen = np.zeros(dims, np.int32)
    for b in boxes:
        seen[tuple(slice(a, c) for a, c in b)] += 1
    want = np.zeros(int(np.prod(dims)), np.int32)
    want[lo:hi] = 1
    np.testing.assert_array_equal(seen.reshape(-1), want)


@pytest.mark.parametrize("balance", ["windows", "rows"])
@pytest.mark.parametrize("img,roi,world", [((165, 1024, 768), (112, 112, 112), 8), ((448, 448, 448), (112, 112, 112), 8),
                                            ((20, 30, 34), (8, 12, 12), 3), ((9, 40, 12), (8, 12, 12), 2),
                                            ((8, 12, 12), (8, 12, 12), 4), ((12, 76, 56), (8, 8, 8), 8), ((12, 76, 56), (8, 8, 8), 5),
                                            ((30, 30, 30), (8, 8, 8), 7)])
def test_slab_plan_partitions_windows_and_voxels(img, roi, world, balance):
    grown = tuple(max(i, r) for i, r in zip(img, roi))
    starts = WO.window_starts(grown, roi, WO.scan_interval(grown, roi, 0.5))
    plan = plan_slabs(grown, roi, starts, world, balance=balance)
    ax = plan.axis
    assert sorted(sum((plan.windows_of(r) for r in range(world)), [])) == sorted(tuple(s) for s in starts)
    per_rank = [len(plan.windows_of(r)) for r in range(world)]
    if balance == "windows":
        assert max(per_rank) - min(per_rank) <= 1            # the reference's [rank::world] balance (lazy.py:1104)
    # the owned boxes of all ranks partition the volume (checked voxel by voxel on a coarse scale for the large grids)
    scale = 8 if max(grown) > 200 else 1
    if scale == 1:
        seen = np.zeros(grown, np.int16)
        for r in range(world):
            for b in plan.own_boxes[r]:
                _mark(seen, b)
        assert seen.min() == 1 and seen.max() == 1
    assert sum(_vol(b) for r in range(world) for b in plan.own_boxes[r]) == int(np.prod(grown))
    for r in range(world):
        wins = plan.windows_of(r)
        if wins:
            assert plan.extent[r] == (min(w[ax] for w in wins), min(grown[ax], max(w[ax] for w in wins) + roi[ax]))
            for w in wins:          # every window lies inside one of the rank's cover boxes, and the cell of the window is owned by the rank
                assert any(all(cb[a][0] <= w[a] and min(grown[a], w[a] + roi[a]) <= cb[a][1] for a in range(3)) for cb in plan.cover_boxes[r])
        for b in plan.own_boxes[r]:     # what a rank owns lies inside the planes its accumulators cover
            assert plan.extent[r][0] <= b[ax][0] and b[ax][1] <= plan.extent[r][1]
        sends, recvs = exchange_schedule(plan, r)
        for q, box in sends:                      # what r sends to q is exactly what q expects from r
            assert (r, box) in exchange_schedule(plan, q)[1]
            assert any(all(ob[a][0] <= box[a][0] and box[a][1] <= ob[a][1] for a in range(3)) for ob in plan.own_boxes[q])
        for q, box in recvs:
            assert (r, box) in exchange_schedule(plan, q)[0]
    if scale == 1:
        # every voxel a rank's windows touch outside its own cells is covered by exactly one send; sends never overlap
        for r in range(world):
            touched = np.zeros(grown, bool)
            for w in plan.windows_of(r):
                touched[w[0]:w[0] + roi[0], w[1]:w[1] + roi[1], w[2]:w[2] + roi[2]] = True
            owned = np.zeros(grown, np.int16)
            for b in plan.own_boxes[r]:
                _mark(owned, b)
            sent = np.zeros(grown, np.int16)
            for _q, box in exchange_schedule(plan, r)[0]:
                _mark(sent, box)
            assert sent.max(initial=0) <= 1
            assert not (touched & (owned == 0) & (sent == 0)).any()
            assert not ((sent == 1) & (owned == 1)).any()


def test_lucchi_grid_at_world_8_is_balanced_and_talks_to_neighbours_only():
    """468 windows (2 x 18 x 13) at world 8 -> 58 / 59 per rank (whole rows gave 52 / 78: <= 6.0x)."""
    grown, roi = (165, 1024, 768), (112, 112, 112)
    starts = WO.window_starts(grown, roi, WO.scan_interval(grown, roi, 0.5))
    plan = plan_slabs(grown, roi, starts, 8)
    per_rank = [len(plan.windows_of(r)) for r in range(8)]
    assert plan.axis == 1 and sum(per_rank) == 468 and sorted(set(per_rank)) == [58, 59]
    assert max(per_rank) / (468 / 8) <= 1.05
    rows = plan_slabs(grown, roi, starts, 8, balance="rows")
    assert sorted({len(rows.windows_of(r)) for r in range(8)}) == [52, 78]
    for r in range(8):
        peers = {q for q, _b in exchange_schedule(plan, r)[0]}
        assert peers <= {r - 2, r - 1, r + 1, r + 2} and peers          # contiguous runs: only ranks whose cells touch
        assert len(plan.own_boxes[r]) <= 5


def _oracle_accumulate(vol, roi, plan):
    wmap = torch.from_numpy(WO.importance_map(roi, "bump"))
    ax = plan.axis

    def accumulate(windows, ext):
        L, H = ext
        dims = list(plan.image_size)
        dims[ax] = H - L
        value = torch.zeros([2] + dims)
        weight = torch.zeros(dims)
        for s in windows:
            pred = _net(WO.extract_window(vol, s, roi, "constant", 0.0))[0]
            sl = [slice(s[a], s[a] + roi[a]) for a in range(3)]
            sl[ax] = slice(s[ax] - L, s[ax] - L + roi[ax])
            value[(slice(None), *sl)] += pred * wmap
            weight[tuple(sl)] += wmap
        return value, weight
    return accumulate


def _worker(rank, world, port, img, roi, tmp, balance):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    vol = torch.rand((1, 1) + img, generator=torch.Generator().manual_seed(5))
    starts = WO.window_starts(img, roi, WO.scan_interval(img, roi, 0.5))
    plan = plan_slabs(img, roi, starts, world, balance=balance)
    pieces = slab_predict(plan, rank, _oracle_accumulate(vol, roi, plan),
                          lambda v, w: v / torch.clamp_min(w, 1e-4))          # normalize_weighted_accumulator, window.py:275-294
    assert [b for b, _t in pieces] == list(plan.own_boxes[rank])
    np.savez(os.path.join(tmp, f"pieces{rank}.npz"), **{f"p{i}": t.numpy() for i, (_b, t) in enumerate(pieces)})
    # device-side gather (one tensor all_gather of padded flat pieces; channel count agreed by all_reduce)
    full = gather_slabs(pieces, plan, img, torch.device("cpu"))
    np.save(os.path.join(tmp, f"gathered{rank}.npy"), full.numpy())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


# (12, 76, 56) at roi 8^3 / overlap 0.5 has the window grid of the Lucchi++ volume at roi 112^3: 2 x 18 x 13 = 468 windows -- the split
# the 8-GPU strong-scaling leg of bench.py runs: 58 / 59 windows per rank (whole rows: 52 / 78)
@pytest.mark.parametrize("img,world,roi,balance", [((20, 30, 34), 2, (8, 12, 12), "windows"), ((14, 22, 50), 3, (8, 12, 12), "windows"),
                                                    ((12, 76, 56), 8, (8, 8, 8), "windows"), ((14, 22, 50), 3, (8, 12, 12), "rows")])
def test_slab_exchange_over_gloo_matches_single_process(img, world, roi, balance, tmp_path):
    starts = WO.window_starts(img, roi, WO.scan_interval(img, roi, 0.5))
    plan = plan_slabs(img, roi, starts, world, balance=balance)
    if world == 8:
        assert len(starts) == 468 and len({s[1] for s in starts}) == 18 and len({s[2] for s in starts}) == 13
        per_rank = [len(plan.windows_of(r)) for r in range(world)]
        assert plan.axis == 1 and sum(per_rank) == 468 and sorted(set(per_rank)) == [58, 59]
    mp.spawn(_worker, args=(world, 29500 + (os.getpid() * 7 + world) % 2000, img, roi, str(tmp_path), balance), nprocs=world, join=True)
    got = np.full((2,) + img, np.nan, np.float32)
    for r in range(world):
        data = np.load(tmp_path / f"pieces{r}.npz")
        for i, b in enumerate(plan.own_boxes[r]):
            got[:, b[0][0]:b[0][1], b[1][0]:b[1][1], b[2][0]:b[2][1]] = data[f"p{i}"]
    vol = torch.rand((1, 1) + img, generator=torch.Generator().manual_seed(5))
    ref = WO.eager_sliding_window(vol, _net, roi=roi, overlap=0.5, mode="bump", sw_batch_size=2)[0].numpy()
    assert got.shape == ref.shape and not np.isnan(got).any()
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-6)
    for r in range(world):            # every rank holds the full volume after gather_slabs
        np.testing.assert_array_equal(np.load(tmp_path / f"gathered{r}.npy"), got)
