"""Slab ownership with neighbour halo exchange: one volume tiled across the GPUs of a node.

The reference's multi-GPU inference either replicates full-volume accumulators on every rank and reduces them to
rank 0 (`lazy_distributed.py:78-107`) or goes through chunk files.  On a node whose GPUs are linked point-to-point
(xGMI) the cheaper plan is ownership: the windows are dealt out in contiguous runs of the axis-major window order (counts
within one of each other, the balance of the reference's `[rank::world]` deal, `inference/lazy.py:1104`), rank r runs its
windows into accumulators that cover only the planes those windows touch, and the only communication is the boxes of
weighted partial sums (value + weight) that fall into cells another rank owns -- at overlap 0.5 at most roi/2 voxels
deep per face, sent once per volume with paired send/recv (RCCL p2p; no all-reduce, no rank-0 gather) to the ranks whose
cells touch (round 6: rounds 3-5 cut whole window rows, 52 / 78 windows per rank on the Lucchi++ grid at world 8, which capped
strong scaling at 6.0x; `balance="rows"` keeps that plan).  Every rank then normalises the boxes it owns; the result stays
sharded unless a gather is requested.

`plan_slabs` is pure integer logic (tested on CPU); `slab_predict` takes the accumulation as a callable, so the exchange
protocol is covered by world-size-2 / 3 / 8 gloo tests with the CPU oracle as the accumulator, while on the GPU box the
accumulator is `EagerSlidingWindowEngine.accumulate` (HIP kernels).  Sums arrive in a different order than in the
single-process engine (own windows first, then the neighbours' bands), so results agree to fp32 rounding, not bit-exactly.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


Box = Tuple[Tuple[int, int], Tuple[int, int], Tuple[int, int]]      # ((z0, z1), (y0, y1), (x0, x1)), global voxel indices


@dataclass(frozen=True)
class SlabPlan:
    image_size: Tuple[int, int, int]            # grown (>= roi) volume size the grid was planned on
    roi: Tuple[int, int, int]
    starts: Tuple[Tuple[int, int, int], ...]     # every window origin, reference order (z outermost)
    row_rank: Tuple[int, ...]                    # owner rank of each distinct row of windows along `axis` (-1: a row shared by ranks)
    z_rows: Tuple[int, ...]                      # distinct window origins along `axis`, ascending
    own: Tuple[Tuple[int, int], ...]             # balance="rows": [B_r, B_{r+1}) planes along `axis` owned by rank r (may be empty);
                                                 # balance="windows": the plane range the rank's owned boxes span
    extent: Tuple[Tuple[int, int], ...]          # [L_r, H_r) planes along `axis` rank r's windows touch (empty -> (0, 0))
    axis: int = 0                                # primary axis of the cut (0 = z, 1 = y, 2 = x): accumulators are slabs along it
    balance: str = "rows"
    win_rank: Tuple[int, ...] = ()               # owner rank of every window of `starts`
    own_boxes: Tuple[Tuple[Box, ...], ...] = ()  # disjoint voxel boxes owned by rank r; over all ranks they partition the volume
    cover_boxes: Tuple[Tuple[Box, ...], ...] = ()   # voxel boxes that together contain every window of rank r (may overlap each other)

    def windows_of(self, rank: int) -> List[Tuple[int, int, int]]:
        return [s for s, r in zip(self.starts, self.win_rank) if r == rank]


def _lex_boxes(lo: int, hi: int, dims: Sequence[int]) -> List[List[Tuple[int, int]]]:
    """The lexicographic index range [lo, hi) of a grid of `dims` cells (first dimension slowest) as disjoint index boxes: at most
    2 * len(dims) - 1 of them (partial head runs, one block of full rows, partial tail runs)."""
    if hi <= lo:
        return []
    if len(dims) == 1:
        return [[(lo, hi)]]
    stride = 1
    for d in dims[1:]:
        stride *= int(d)
    full = [(0, int(d)) for d in dims[1:]]
    i0, r0 = divmod(lo, stride)
    i1, r1 = divmod(hi, stride)
    if i0 == i1:
        return [[(i0, i0 + 1)] + b for b in _lex_boxes(r0, r1, dims[1:])]
    out = []
    if r0 > 0:
        out += [[(i0, i0 + 1)] + b for b in _lex_boxes(r0, stride, dims[1:])]
        i0 += 1
    if i1 > i0:
        out.append([(i0, i1)] + full)
    if r1 > 0:
        out += [[(i1, i1 + 1)] + b for b in _lex_boxes(0, r1, dims[1:])]
    return out


def _cell_edges(rows: Sequence[int], r: int, size: int) -> List[int]:
    """Ownership boundaries along one axis: half-way between the centres of consecutive window rows (any partition of [0, size) is
    valid; this one minimises the bands), 0 and `size` at the ends."""
    e = [0] + [min(size, max(0, (a + b + r) // 2)) for a, b in zip(rows, rows[1:])] + [size]
    for i in range(1, len(e)):
        e[i] = max(e[i], e[i - 1])
    return e


def plan_slabs(image_size: Sequence[int], roi: Sequence[int], starts: Sequence[Sequence[int]], world: int,
               axis: Optional[int] = None, balance: str = "windows") -> SlabPlan:
    """Who runs which windows and who owns (normalises) which voxels.

    balance="windows" (default): the windows, ordered lexicographically with the axis that has the most window rows slowest (Lucchi++
    165x1024x768: 18 rows in y, 13 in x, 2 in z), are dealt out in contiguous runs whose lengths differ by at most one -- the balance
    of the reference's `[rank::world]` deal (`inference/lazy.py:1104`; 468 windows at world 8: 58 / 59 each) -- but contiguous, so a
    rank's windows stay a staircase of neighbouring rows and its p2p bands go to the few ranks whose cells touch it.  Ownership
    follows the windows: the volume is cut into cells by the midpoints between neighbouring window centres on every axis, and the
    cell of a window belongs to the window's rank (at most five boxes per rank).
    balance="rows" (rounds 3-5): whole window rows along one axis per rank (one box per rank; 52 / 78 windows on that grid)."""
    if balance not in ("windows", "rows"):
        raise ValueError(f"plan_slabs: balance must be 'windows' or 'rows', got {balance!r}")
    st = tuple(tuple(int(v) for v in s) for s in starts)
    size = tuple(int(v) for v in image_size)
    rr = tuple(int(v) for v in roi)
    rows = [sorted({s[a] for s in st}) for a in range(3)]
    if axis is None:
        axis = max(range(3), key=lambda a: (len(rows[a]), -a))
    axis = int(axis)
    grid_is_product = len(st) == len(rows[0]) * len(rows[1]) * len(rows[2]) and len(set(st)) == len(st)
    if balance == "windows" and not grid_is_product:
        balance = "rows"                                   # a filtered window list: fall back to whole rows
    z_rows = tuple(rows[axis])
    n = len(z_rows)
    if balance == "rows":
        Z, rz = size[axis], rr[axis]
        row_rank = tuple(min(world - 1, (i * world) // n) for i in range(n)) if n >= world else tuple(range(n))
        bounds = [0] * (world + 1)
        bounds[world] = Z
        last_owner_edge = 0
        for r in range(1, world):
            prev_rows = [z for z, q in zip(z_rows, row_rank) if q < r]
            next_rows = [z for z, q in zip(z_rows, row_rank) if q >= r]
            if not prev_rows:
                b = 0
            elif not next_rows:
                b = Z
            else:
                b = (prev_rows[-1] + next_rows[0] + rz) // 2       # midpoint of the two rows' centres
            bounds[r] = min(Z, max(last_owner_edge, b))
            last_owner_edge = bounds[r]
        own = tuple((bounds[r], bounds[r + 1]) for r in range(world))
        rank_of_row = dict(zip(z_rows, row_rank))
        win_rank = tuple(rank_of_row[s[axis]] for s in st)
        own_boxes, cover_boxes, extent = [], [], []
        for r in range(world):
            mine = [z for z, q in zip(z_rows, row_rank) if q == r]
            extent.append((max(0, mine[0]), min(Z, mine[-1] + rz)) if mine else (0, 0))
            ob = [(0, size[a]) for a in range(3)]
            ob[axis] = own[r]
            own_boxes.append((tuple(ob),) if own[r][1] > own[r][0] else ())
            cb = [(0, size[a]) for a in range(3)]
            cb[axis] = extent[r]
            cover_boxes.append((tuple(cb),) if mine else ())
        return SlabPlan(size, rr, st, row_rank, z_rows, own, tuple(extent), axis, "rows", win_rank, tuple(own_boxes),
                        tuple(cover_boxes))

    # ---- balance == "windows": contiguous runs of the axis-major window order, counts within one of each other
    order = [axis] + sorted((a for a in range(3) if a != axis), key=lambda a: (-len(rows[a]), a))      # slowest -> fastest
    dims = [len(rows[a]) for a in order]
    index = [{v: i for i, v in enumerate(rows[a])} for a in range(3)]
    total = len(st)
    cuts = [(r * total) // world for r in range(world + 1)]
    edges = [_cell_edges(rows[a], rr[a], size[a]) for a in range(3)]

    def flat(s):
        f = 0
        for a, d in zip(order, dims):
            f = f * d + index[a][s[a]]
        return f

    def rank_of_flat(f):
        lo, hi = 0, world                                   # largest r with cuts[r] <= f
        while hi - lo > 1:
            mid = (lo + hi) // 2
            if cuts[mid] <= f:
                lo = mid
            else:
                hi = mid
        return lo

    win_rank = tuple(rank_of_flat(flat(s)) for s in st)
    own_boxes, cover_boxes, extent, own = [], [], [], []
    for r in range(world):
        ob, cb = [], []
        for ib in _lex_boxes(cuts[r], cuts[r + 1], dims):
            vox_own = [None] * 3
            vox_cov = [None] * 3
            for a, (i0, i1) in zip(order, ib):
                vox_own[a] = (edges[a][i0], edges[a][i1])
                vox_cov[a] = (rows[a][i0], min(size[a], rows[a][i1 - 1] + rr[a]))
            if all(hi > lo for lo, hi in vox_own):
                ob.append(tuple(vox_own))
            cb.append(tuple(vox_cov))
        own_boxes.append(tuple(ob))
        cover_boxes.append(tuple(cb))
        extent.append((min(b[axis][0] for b in cb), max(b[axis][1] for b in cb)) if cb else (0, 0))
        own.append((min(b[axis][0] for b in ob), max(b[axis][1] for b in ob)) if ob else (0, 0))
    per_row = {}
    for s, q in zip(st, win_rank):
        per_row.setdefault(s[axis], set()).add(q)
    row_rank = tuple(next(iter(per_row[z])) if len(per_row[z]) == 1 else -1 for z in z_rows)
    return SlabPlan(size, rr, st, row_rank, z_rows, tuple(own), tuple(extent), axis, "windows", win_rank, tuple(own_boxes),
                    tuple(cover_boxes))


def _box_and(a: Box, b: Box) -> Optional[Box]:
    out = tuple((max(p[0], q[0]), min(p[1], q[1])) for p, q in zip(a, b))
    return out if all(hi > lo for lo, hi in out) else None


def _bands(plan: SlabPlan, src: int, dst: int) -> List[Box]:
    """Voxel boxes of `src`'s partial sums that `dst` owns: per owned box of `dst` the bounding box of what `src`'s windows can have
    touched inside it (boxes of different owned boxes are disjoint, so nothing is added twice; inside a bounding box the voxels
    `src` did not touch hold zeros)."""
    out = []
    for ob in plan.own_boxes[dst]:
        hit = [h for h in (_box_and(cb, ob) for cb in plan.cover_boxes[src]) if h]
        if hit:
            out.append(tuple((min(h[a][0] for h in hit), max(h[a][1] for h in hit)) for a in range(3)))
    return out


def exchange_schedule(plan: SlabPlan, rank: int):
    """(sends, recvs): sends = [(peer, box)] parts of MY partial sums that PEER owns; recvs = parts of PEER's sums that I own.
    Boxes are ((z0, z1), (y0, y1), (x0, x1)) in global voxel indices; both lists ascend in (peer, box)."""
    world = len(plan.own)
    sends, recvs = [], []
    for q in range(world):
        if q == rank:
            continue
        sends += [(q, b) for b in _bands(plan, rank, q)]
        recvs += [(q, b) for b in _bands(plan, q, rank)]
    return sends, recvs


def slab_predict(plan: SlabPlan, rank: int, accumulate: Callable[[List[Tuple[int, int, int]], Tuple[int, int]], Tuple[torch.Tensor, torch.Tensor]],
                 finalize: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], *, group=None) -> List[Tuple[Box, torch.Tensor]]:
    """Run this rank's windows, exchange the bands, normalise the owned boxes.

    accumulate(windows, (L, H)) -> (value (C, *dims), weight (*dims)) with dims = image size except H-L along
    plan.axis: un-normalised sums of `windows` (global origins) over planes [L, H) of that axis.
    finalize(value, weight) -> normalised value (may work in place).
    Returns [(box, normalised (C, *box size))] for the boxes this rank owns (one box per rank with balance="rows"; empty when the
    rank owns nothing)."""
    world = len(plan.own)
    L, H = plan.extent[rank]
    ax = plan.axis

    def cut(t, box, lead):                          # `box` of a tensor that starts at plane L of the slab axis (lead = 1: (C, ...))
        for a, (lo, hi) in enumerate(box):
            off = L if a == ax else 0
            t = t.narrow(a + lead, lo - off, hi - lo)
        return t

    mine = plan.windows_of(rank)
    value = weight = None
    if mine:
        value, weight = accumulate(mine, (L, H))
    sends, recvs = exchange_schedule(plan, rank)
    if world > 1 and (sends or recvs):
        if value is None:
            raise RuntimeError("slab plan inconsistent: a rank without windows cannot have bands to send or cells to own")
        ops, bufs = [], []
        for peer, box in sends:
            ops += [dist.P2POp(dist.isend, cut(value, box, 1).contiguous(), peer, group),
                    dist.P2POp(dist.isend, cut(weight, box, 0).contiguous(), peer, group)]
        C = value.shape[0]
        for peer, box in recvs:
            shp = [hi - lo for lo, hi in box]
            v = torch.empty([C] + shp, dtype=value.dtype, device=value.device)
            w = torch.empty(shp, dtype=weight.dtype, device=weight.device)
            bufs.append((box, v, w))
            ops += [dist.P2POp(dist.irecv, v, peer, group), dist.P2POp(dist.irecv, w, peer, group)]
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        for box, v, w in bufs:                          # fixed order: ascending (peer rank, box)
            cut(value, box, 1).add_(v)
            cut(weight, box, 0).add_(w)
    if value is None:
        return []
    return [(box, finalize(cut(value, box, 1).contiguous(), cut(weight, box, 0).contiguous())) for box in plan.own_boxes[rank]]


def join_pieces(pieces: List[Tuple[Box, torch.Tensor]], plan: SlabPlan, rank: int) -> Optional[torch.Tensor]:
    """balance="rows": the rank's one box as a tensor (None when it owns nothing) -- the return value of rounds 3-5."""
    if plan.balance != "rows":
        raise ValueError("join_pieces: only a balance='rows' plan gives every rank one box")
    return pieces[0][1] if pieces else None


def slab_extent(full_size: Sequence[int], engine, world: int, rank: int, balance: str = "windows") -> Tuple[int, Tuple[int, int]]:
    """(axis, (L, H)): the planes of a `full_size` volume that rank's windows read -- what a rank has to hold in HBM when the
    volume is handed to slab_predict_volume as per-rank extents (`full_size=`); H is clipped to the stored size."""
    orig = tuple(int(v) for v in full_size)
    image_size, starts = engine.plan(orig)
    plan = plan_slabs(image_size, engine.roi_size, starts, world, balance=balance)
    L, H = plan.extent[rank]
    return plan.axis, (L, min(H, orig[plan.axis]))


@torch.no_grad()
def slab_predict_volume(vol: torch.Tensor, engine, network, *, group=None, gather: bool = False,
                        full_size: Optional[Sequence[int]] = None, balance: str = "windows"):
    """Device path: `vol` (C, Z, Y, X) fp32 on this rank's GPU, `engine` an EagerSlidingWindowEngine.  Either every rank
    passes the whole volume, or -- with `full_size` = the (Z, Y, X) size of the whole volume -- only the planes its own
    windows touch (`slab_extent`): nothing outside a rank's extent is ever read.  Returns this rank's owned pieces
    [(box, (C_out, *box size))] cropped to the original size, or with gather=True the full volume on every rank (one all_gather)."""
    from .. import _native as nat
    from .. import hip_ops as ops
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    orig = tuple(int(v) for v in (full_size if full_size is not None else vol.shape[1:]))
    image_size, starts = engine.plan(orig)
    plan = plan_slabs(image_size, engine.roi_size, starts, world, balance=balance)

    ax = plan.axis
    held_from = 0
    if full_size is not None:
        held_from, held_to = plan.extent[rank][0], min(plan.extent[rank][1], orig[ax])
        want = list(orig)
        want[ax] = max(0, held_to - held_from)
        if [int(v) for v in vol.shape[1:]] != want:
            raise ValueError(f"slab_predict_volume(full_size={orig}): rank {rank} must pass planes [{held_from}, {held_to}) along "
                             f"axis {ax}, i.e. spatial shape {tuple(want)}; got {tuple(vol.shape[1:])}")

    def accumulate(windows, ext):
        L, H = ext
        sub = vol.narrow(ax + 1, L - held_from, min(H, orig[ax]) - L).contiguous()
        shifted = [tuple(c - L if a == ax else c for a, c in enumerate(w)) for w in windows]
        value, weight = engine.accumulate(sub, network, starts=shifted)
        # the engine grows its accumulators to at least one window; keep exactly the planes of the extent
        have = value.shape[ax + 1]
        if have > H - L:
            value, weight = value.narrow(ax + 1, 0, H - L).contiguous(), weight.narrow(ax, 0, H - L).contiguous()
        elif have < H - L:
            shp = list(weight.shape)
            shp[ax] = H - L - have
            value = torch.cat([value, value.new_zeros([value.shape[0]] + shp)], dim=ax + 1)
            weight = torch.cat([weight, weight.new_zeros(shp)], dim=ax)
        return value, weight

    def finalize(value, weight):
        ops.blend_finalize(value, weight, clamp=1e-4, act=nat.ACT_NONE)
        return value

    pieces = []
    for box, t in slab_predict(plan, rank, accumulate, finalize, group=group):
        keep = tuple((lo, min(hi, orig[a])) for a, (lo, hi) in enumerate(box))       # the grown volume's pad leaves here
        if all(hi > lo for lo, hi in keep):
            pieces.append((keep, t[:, :keep[0][1] - keep[0][0], :keep[1][1] - keep[1][0], :keep[2][1] - keep[2][0]].contiguous()))
    if not gather:
        return pieces
    return gather_slabs(pieces, plan, orig, vol.device, group=group)


def _owned_in(plan: SlabPlan, rank: int, orig: Sequence[int]) -> List[Box]:
    out = []
    for box in plan.own_boxes[rank]:
        keep = tuple((lo, min(hi, int(orig[a]))) for a, (lo, hi) in enumerate(box))
        if all(hi > lo for lo, hi in keep):
            out.append(keep)
    return out


def gather_slabs(pieces, plan: SlabPlan, orig: Sequence[int], device, *, c_out: Optional[int] = None, group=None) -> torch.Tensor:
    """All ranks -> the full (C_out, *orig) volume, on the device: every rank contributes its owned pieces, flattened one after the
    other and padded to the largest per-rank voxel count (the plan is known everywhere, so sizes need no exchange) through ONE tensor
    all-gather (RCCL all_gather over xGMI on the GPU box; no pickling, no host staging); every rank then places every rank's pieces."""
    world = len(plan.own)
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if isinstance(pieces, torch.Tensor) or pieces is None:        # rounds 3-5 signature: the one slab of a balance="rows" plan
        boxes = _owned_in(plan, rank, orig)
        pieces = [(boxes[0], pieces)] if (pieces is not None and boxes) else []
    if c_out is None:
        # a rank that owns nothing still has to join the collective with the right channel count
        c = torch.tensor([int(pieces[0][1].shape[0]) if pieces else 0], device=device, dtype=torch.int64)
        if world > 1:
            dist.all_reduce(c, op=dist.ReduceOp.MAX, group=group)
        c_out = int(c.item())
    owned = [_owned_in(plan, r, orig) for r in range(world)]
    counts = [sum((b[0][1] - b[0][0]) * (b[1][1] - b[1][0]) * (b[2][1] - b[2][0]) for b in bs) for bs in owned]
    mine = torch.zeros(c_out * max(max(counts), 1), dtype=torch.float32, device=device)
    at = 0
    for (box, t), want in zip(pieces, owned[rank]):
        if tuple(box) != tuple(want):
            raise ValueError(f"gather_slabs: rank {rank} passed box {box}, the plan says {want}")
        mine[at:at + t.numel()].copy_(t.reshape(-1))
        at += t.numel()
    if world == 1:
        parts = [mine]
    else:
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
    full = torch.empty([c_out] + [int(v) for v in orig], dtype=torch.float32, device=device)
    for r in range(world):
        at = 0
        for box in owned[r]:
            shp = [c_out] + [hi - lo for lo, hi in box]
            nel = shp[0] * shp[1] * shp[2] * shp[3]
            full[:, box[0][0]:box[0][1], box[1][0]:box[1][1], box[2][0]:box[2][1]] = parts[r][at:at + nel].view(shp)
            at += nel
    return full


__all__ = ["SlabPlan", "plan_slabs", "exchange_schedule", "slab_predict", "slab_predict_volume", "slab_extent", "gather_slabs",
           "join_pieces"]
