"""z-slab ownership with neighbour halo exchange: one volume tiled across the GPUs of a node.

The reference's multi-GPU inference either replicates full-volume accumulators on every rank and reduces them to
rank 0 (`lazy_distributed.py:78-107`) or goes through chunk files.  On a node whose GPUs are linked point-to-point
(xGMI) the cheaper plan is ownership: the window grid is cut into contiguous groups of z-rows, rank r runs its
windows into accumulators that cover only the extent of those windows, and the only communication is the band of
weighted partial sums (value + weight) that falls into a neighbour's slab -- at overlap 0.5 at most roi_z/2 planes per
face, sent once per volume with paired send/recv (RCCL p2p; no all-reduce, no rank-0 gather).  Every rank then
normalises its own slab; the result stays sharded unless a gather is requested.

`plan_slabs` is pure integer logic (tested on CPU); `slab_predict` takes the accumulation as a callable, so the exchange
protocol is covered by a world-size-2 gloo test with the CPU oracle as the accumulator, while on the GPU box the
accumulator is `EagerSlidingWindowEngine.accumulate` (HIP kernels).  Sums arrive in a different order than in the
single-process engine (own windows first, then the neighbours' bands), so results agree to fp32 rounding, not bit-exactly.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class SlabPlan:
    image_size: Tuple[int, int, int]            # grown (>= roi) volume size the grid was planned on
    roi: Tuple[int, int, int]
    starts: Tuple[Tuple[int, int, int], ...]     # every window origin, reference order (z outermost)
    row_rank: Tuple[int, ...]                    # owner rank of each distinct z-row of windows
    z_rows: Tuple[int, ...]                      # distinct window z origins, ascending
    own: Tuple[Tuple[int, int], ...]             # [B_r, B_{r+1}) output planes owned by rank r (may be empty)
    extent: Tuple[Tuple[int, int], ...]          # [L_r, H_r) planes rank r's windows touch (empty -> (0, 0))
    axis: int = 0                                # spatial axis the slabs are cut along (0 = z, 1 = y, 2 = x)

    def windows_of(self, rank: int) -> List[Tuple[int, int, int]]:
        rows = {z for z, r in zip(self.z_rows, self.row_rank) if r == rank}
        return [s for s in self.starts if s[self.axis] in rows]


def plan_slabs(image_size: Sequence[int], roi: Sequence[int], starts: Sequence[Sequence[int]], world: int,
               axis: Optional[int] = None) -> SlabPlan:
    """Contiguous, balanced split of the window rows along `axis` (default: the axis with the most rows -- Lucchi++
    165x1024x768 has 2 rows in z but 18 in y); ownership boundaries half-way between the centres of the neighbouring
    rows of two ranks (any partition of [0, size) works; this one minimises the bands)."""
    st = tuple(tuple(int(v) for v in s) for s in starts)
    if axis is None:
        axis = max(range(3), key=lambda a: (len({s[a] for s in st}), -a))
    Z = int(image_size[axis])
    rz = int(roi[axis])
    z_rows = tuple(sorted({s[axis] for s in st}))
    n = len(z_rows)
    row_rank = tuple(min(world - 1, (i * world) // n) for i in range(n)) if n >= world else tuple(range(n))
    bounds = [0] * (world + 1)
    bounds[world] = Z
    extent = []
    for r in range(world):
        rows = [z for z, rr in zip(z_rows, row_rank) if rr == r]
        extent.append((max(0, rows[0]), min(Z, rows[-1] + rz)) if rows else (0, 0))
    last_owner_edge = 0
    for r in range(1, world):
        prev_rows = [z for z, rr in zip(z_rows, row_rank) if rr < r]
        next_rows = [z for z, rr in zip(z_rows, row_rank) if rr >= r]
        if not prev_rows:
            b = 0
        elif not next_rows:
            b = Z
        else:
            b = (prev_rows[-1] + next_rows[0] + rz) // 2       # midpoint of the two rows' centres
        bounds[r] = min(Z, max(last_owner_edge, b))
        last_owner_edge = bounds[r]
    own = tuple((bounds[r], bounds[r + 1]) for r in range(world))
    return SlabPlan(tuple(int(v) for v in image_size), tuple(int(v) for v in roi), st, row_rank, z_rows, own, tuple(extent),
                    int(axis))


def _overlap(a: Tuple[int, int], b: Tuple[int, int]) -> Optional[Tuple[int, int]]:
    lo, hi = max(a[0], b[0]), min(a[1], b[1])
    return (lo, hi) if lo < hi else None


def exchange_schedule(plan: SlabPlan, rank: int):
    """(sends, recvs): sends = [(peer, z_lo, z_hi)] bands of MY extent inside PEER's slab; recvs = bands of PEER's
    extent inside MY slab.  Global plane indices."""
    world = len(plan.own)
    sends, recvs = [], []
    for q in range(world):
        if q == rank:
            continue
        ov = _overlap(plan.extent[rank], plan.own[q])
        if ov:
            sends.append((q, ov[0], ov[1]))
        ov = _overlap(plan.extent[q], plan.own[rank])
        if ov:
            recvs.append((q, ov[0], ov[1]))
    return sends, recvs


def slab_predict(plan: SlabPlan, rank: int, accumulate: Callable[[List[Tuple[int, int, int]], Tuple[int, int]], Tuple[torch.Tensor, torch.Tensor]],
                 finalize: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], *, group=None) -> Optional[torch.Tensor]:
    """Run this rank's windows, exchange the bands, normalise the owned slab.

    accumulate(windows, (L, H)) -> (value (C, *dims), weight (*dims)) with dims = image size except H-L along
    plan.axis: un-normalised sums of `windows` (global origins) over planes [L, H) of that axis.
    finalize(value, weight) -> normalised value (may work in place).
    Returns the owned slab (B1-B0 planes along plan.axis) or None when this rank owns nothing."""
    world = len(plan.own)
    L, H = plan.extent[rank]
    B0, B1 = plan.own[rank]
    ax = plan.axis

    def cut(t, lo, hi, lead):                       # planes [lo, hi) along the slab axis (lead = 1 for (C, ...) tensors)
        return t.narrow(ax + lead, lo, hi - lo)

    mine = plan.windows_of(rank)
    value = weight = None
    if mine:
        value, weight = accumulate(mine, (L, H))
    sends, recvs = exchange_schedule(plan, rank)
    if world > 1 and (sends or recvs):
        if value is None:
            raise RuntimeError("slab plan inconsistent: a rank without windows cannot have bands to send")
        ops, bufs = [], []
        for peer, z0, z1 in sends:
            v = cut(value, z0 - L, z1 - L, 1).contiguous()
            w = cut(weight, z0 - L, z1 - L, 0).contiguous()
            ops += [dist.P2POp(dist.isend, v, peer, group), dist.P2POp(dist.isend, w, peer, group)]
        C = None if value is None else value.shape[0]
        for peer, z0, z1 in recvs:
            ref = value if value is not None else None
            if ref is None:
                raise RuntimeError("slab plan inconsistent: a rank that owns planes must have windows")
            shp = list(weight.shape)
            shp[ax] = z1 - z0
            v = torch.empty([C] + shp, dtype=ref.dtype, device=ref.device)
            w = torch.empty(shp, dtype=weight.dtype, device=weight.device)
            bufs.append((z0, z1, v, w))
            ops += [dist.P2POp(dist.irecv, v, peer, group), dist.P2POp(dist.irecv, w, peer, group)]
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        for z0, z1, v, w in bufs:                       # fixed order: ascending peer rank
            cut(value, z0 - L, z1 - L, 1).add_(v)
            cut(weight, z0 - L, z1 - L, 0).add_(w)
    if B1 <= B0 or value is None:
        return None
    return finalize(cut(value, B0 - L, B1 - L, 1).contiguous(), cut(weight, B0 - L, B1 - L, 0).contiguous())


def slab_extent(full_size: Sequence[int], engine, world: int, rank: int) -> Tuple[int, Tuple[int, int]]:
    """(axis, (L, H)): the planes of a `full_size` volume that rank's windows read -- what a rank has to hold in HBM when the
    volume is handed to slab_predict_volume as per-rank extents (`full_size=`); H is clipped to the stored size."""
    orig = tuple(int(v) for v in full_size)
    image_size, starts = engine.plan(orig)
    plan = plan_slabs(image_size, engine.roi_size, starts, world)
    L, H = plan.extent[rank]
    return plan.axis, (L, min(H, orig[plan.axis]))


@torch.no_grad()
def slab_predict_volume(vol: torch.Tensor, engine, network, *, group=None, gather: bool = False,
                        full_size: Optional[Sequence[int]] = None) -> Optional[torch.Tensor]:
    """Device path: `vol` (C, Z, Y, X) fp32 on this rank's GPU, `engine` an EagerSlidingWindowEngine.  Either every rank
    passes the whole volume, or -- with `full_size` = the (Z, Y, X) size of the whole volume -- only the planes its own
    windows touch (`slab_extent`): nothing outside a rank's extent is ever read.  Returns this rank's slab (C_out, B1-B0, Y, X)
    cropped to the original size, or with gather=True the full volume on every rank (all_gather of the slabs)."""
    from .. import _native as nat
    from .. import hip_ops as ops
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    orig = tuple(int(v) for v in (full_size if full_size is not None else vol.shape[1:]))
    image_size, starts = engine.plan(orig)
    plan = plan_slabs(image_size, engine.roi_size, starts, world)

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

    slab = slab_predict(plan, rank, accumulate, finalize, group=group)
    B0, B1 = plan.own[rank]
    if slab is not None:
        keep = [min(orig[a], slab.shape[a + 1]) for a in range(3)]
        keep[ax] = max(0, min(B1, orig[ax]) - B0)
        slab = slab[:, :keep[0], :keep[1], :keep[2]].contiguous()
    if not gather:
        return slab
    return gather_slabs(slab, plan, orig, vol.device, group=group)


def gather_slabs(slab: Optional[torch.Tensor], plan: SlabPlan, orig: Sequence[int], device, *, c_out: Optional[int] = None,
                 group=None) -> torch.Tensor:
    """All ranks -> the full (C_out, *orig) volume, on the device: every rank contributes its slab padded to the widest
    ownership range (the plan is known everywhere, so sizes need no exchange) through ONE tensor all-gather (RCCL
    all_gather over xGMI on the GPU box; no pickling, no host staging), then the pads are dropped."""
    world = len(plan.own)
    ax = plan.axis
    widths = [max(0, min(b1, int(orig[ax])) - b0) for b0, b1 in plan.own]
    wmax = max(widths)
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if c_out is None:
        # a rank that owns nothing still has to join the collective with the right channel count
        c = torch.tensor([0 if slab is None else int(slab.shape[0])], device=device, dtype=torch.int64)
        if world > 1:
            dist.all_reduce(c, op=dist.ReduceOp.MAX, group=group)
        c_out = int(c.item())
    shp = [c_out] + [int(v) for v in orig]
    shp[ax + 1] = wmax
    mine = torch.zeros(shp, dtype=torch.float32, device=device)
    if slab is not None and widths[rank] > 0:
        mine.narrow(ax + 1, 0, widths[rank]).copy_(slab)
    if world == 1:
        return mine.narrow(ax + 1, 0, widths[0]).contiguous()
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    return torch.cat([p.narrow(ax + 1, 0, w) for p, w in zip(parts, widths) if w > 0], dim=ax + 1)


__all__ = ["SlabPlan", "plan_slabs", "exchange_schedule", "slab_predict", "slab_predict_volume", "slab_extent", "gather_slabs"]
