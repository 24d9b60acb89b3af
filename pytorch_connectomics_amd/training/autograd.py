"""Autograd Functions whose forward AND backward are hand-written gfx950 kernels -- the training half of the
hot path (reference: ConnectomicsModule.training_step, training/lightning/model.py:863-910, where Lightning
calls autograd through nnunet_mednext's PyTorch ops).

One Function per MedNeXt block kind (block / down / up) plus one for plain 1x1 convs (stem, heads).  Activations
are NDHWC tensors in the compute dtype (fp32 or bf16 storage); parameter gradients are fp32 in PyTorch layout.
Round-1 status: correctness-first, un-fused schedule (the fused inference kernels are not used here):
forward = dwconv(+stats) -> finalize -> 1x1 expand (pre-activation saved) -> 1x1 project with GELU in its operand
prologue (+residual);
backward = the mirrored sequence with two-stage deterministic reductions for every parameter gradient.
Index shuffles of the down/up residual paths (strided slicing / zeroing of the padded faces) are torch views and
copies; every arithmetic op is a HIP kernel.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.nn as nn

from .. import _native as nat
from .. import hip_ops as ops


# training forward of a block's channel mixer as ONE fused launch that also stores the hidden pre-activation
# (pytc_pw_mlp_train_fwd); False: two GEMM launches (expand, then project with the GELU in its operand prologue)
FUSED_TRAIN_MIXER = True
# ... but only where the voxel rows alone fill the chip: the fused mixer parallelises over rows (64 per workgroup), so the 7^3 / 14^3
# levels of a 112^3 patch run it on 86 / 686 workgroups that each stream BOTH weight images; below this many rows (N * voxels) the
# two single GEMMs, which also share out their output channels (pw_fast_kernels.hip: blockIdx.z), are faster
FUSED_TRAIN_MIXER_MIN_ROWS = int(os.environ.get("PYTC_FUSED_MIXER_MIN_ROWS", "16384"))
# ... and only below this hidden width: a 64-row workgroup of the fused mixer streams BOTH weight images (256 -> 512 -> 128: 384 KB per 64
# rows; the up block at 28^3 ran 233 us fused against ~140 as two GEMMs: 21.95 -> 21.82 ms per 4 x 112^3 step; 256: 21.95)
FUSED_TRAIN_MIXER_MAX_HID = int(os.environ.get("PYTC_FUSED_MIXER_MAX_HID", "512"))
# single 1x1x1 GEMMs (expand / project / data gradients) with at most this many voxel rows in the batch and C_in % 64 == C_out % 128 == 0 run
# on the LDS-tiled GEMM (pytc_pw_conv_fwd, w_paired = 2) instead of the paired-row kernel; 0 = off
TRAIN_GEMM_MAX_ROWS = int(os.environ.get("PYTC_TRAIN_GEMM_MAX_ROWS", "100000"))
# the two data-gradient GEMMs of the mixer as one launch (pytc_pw_mlp_bwd): bit-identical results and 25 % less traffic,
# but measured slower than the two launches it replaces (503 vs ~440 us at 4x112^3, level 0: the exact GELU' between the
# GEMMs sits on the MFMA critical path instead of in a store epilogue) -> off
FUSED_TRAIN_MIXER_BWD = False
# the projecting conv's weight gradient fused with the data gradient behind the activation: one pass over (hp, dy) instead of two at the
# level-0 shapes (csrc/train_kernels.hip pw_wgrad_mfma_kernel<.., DG>; same arithmetic, 27.4 -> 26.4 ms per 4 x 112^3 step)
FUSED_WGRAD_DGRAD = os.environ.get("PYTC_FUSED_WGRAD_DGRAD", "1") != "0"
# data gradient of a residual block, dx = conv_reversed(dt) + dy, with the "+ dy" inside the depthwise kernel (bf16, z-march
# shapes) instead of a separate read-modify-write pass over dx
FUSED_RESIDUAL_DGRAD = True
# full-resolution blocks (32 -> c_hid -> 32): the forward does not store the hidden pre-activation (same arithmetic, same y bits) and the
# backward rebuilds it from the depthwise output inside ONE pass that also yields dhp, dW3 / db3 and -- with NORM_STATS_FROM_WGRAD -- the
# expand conv's weight gradient and the GroupNorm backward sums (csrc/train_kernels.hip mixer_bwd_rc_kernel; round 6)
MIXER_BWD_RC = os.environ.get("PYTC_MIXER_BWD_RC", "1") != "0"
# ... and the forward of those blocks with the packed-fp16 GELU + f16 projection of the inference mixers (pw_mlp_kernel<.., 3, .., STOREH> without
# the store; the hidden pre-activation is still rounded to bf16 first): the backward's gelu_fast(hp) then differs from the forward's
# activation by the polynomial's fit error (<= 1.1e-3, mean 1.5e-4: under the bf16 rounding of the activation)
RC_FWD_F16_GELU = os.environ.get("PYTC_RC_FWD_F16_GELU", "0") == "1"
# GroupNorm backward of a block without its two passes over (dtn, t): the statistics (sum dtn, sum dtn * xhat per sample and channel)
# are contractions of the expand conv's PER-SAMPLE weight-gradient sums with its weights (pytc_pw_wgrad_groupnorm), and the
# data-gradient GEMM applies dt = A*dtn + B*t + C to its own unrounded result in its epilogue (PYTC_RES_NORM_BWD); dtn is never
# stored.  A first version kept the apply PASS over the stored bf16(dtn): statistics of the unrounded values then leave a component
# delta * xhat / rows in dt that the two-pass form removes exactly, xhat correlates strongly with the depthwise conv's input, and the
# level-0 depthwise weight gradients moved from 4.6e-2 to 6.0e-2 relative L2 against the fp32 oracle at BASELINE width
# (tests/test_gpu_baseline_sizes.py, gate 5e-2) -- hence the epilogue form.  False: pytc_norm_bwd.
NORM_STATS_FROM_WGRAD = os.environ.get("PYTC_NORM_STATS_FROM_WGRAD", "1") != "0"
UP_NORM_STATS_FROM_WGRAD = os.environ.get("PYTC_UP_NORM_STATS_FROM_WGRAD", "1") != "0"


# every per-step weight re-layout (MFMA images, tap-major stencils) of a model is rebuilt by ONE launch: the StepPacks set of
# the trunk whose training forward runs (ops.StepPacks; False: one pack launch per use, ~120 per MedNeXt-S step)
BATCHED_WEIGHT_PACKS = True


# weight-gradient kernels of a block backward on a second HIP stream: they are off the critical path (only the optimizer reads
# them), the data-gradient chain is what the next block waits for.  Bit-identical gradients (tools/history/exp_r03_train_lane.py), but
# MEASURED: 34.3 -> 33.9 ms per step before the weight-gradient kernels were fixed and 31.1 -> 31.1 ms after (round 3; the same
# lane around the dense-conv backward of RSUNet / the MONAI-style U-Net: 10.5 -> 10.8 / 11.4 -> 11.4 ms): the step is bound by the
# sum of its HBM-bound level-0 / level-1 kernels, not by the latency-bound deep ones.  Off by default; kept as the switch.
SIDE_STREAM_WGRAD = False
_SIDE_STREAMS = {}


# U-Net skip connections: the encoder features of a level feed its down block AND the up block's skip input, so autograd adds two
# gradients for them (a three-pass elementwise add per level: 0.24 ms of a 22 ms MedNeXt-S step).  With a _SkipBox shared by the two blocks of
# a level the up block's backward leaves its skip gradient in the box (and returns none), and the down block's backward -- which always runs
# later in the same pass: its output feeds the up block -- adds it inside its data-gradient launch (pytc_dwconv3d_bwd_data_add).
FUSE_SKIP_GRAD = os.environ.get("PYTC_FUSE_SKIP_GRAD", "1") != "0"


class _SkipBox:
    """Mailbox between the up block (writer) and the down block (reader) of one level, alive for one forward / backward."""
    __slots__ = ("grad", "armed")

    def __init__(self):
        self.grad = None
        self.armed = False         # the down block has promised to collect: its forward ran with this box and its input needs a gradient


class _WgradLane:
    """`run(fn, *deps)`: fn's launches go to the side stream once everything enqueued on the caller's stream so far is done
    (deps = tensors of the caller's stream fn reads: kept from being recycled under it); `join()`: the caller's stream waits for
    the side stream.  Off (everything on the caller's stream) while per-kernel timing is on."""

    def __init__(self, device):
        self.on = bool(SIDE_STREAM_WGRAD and device.type == "cuda" and not ops.PROFILER.enabled
                       and not torch.cuda.is_current_stream_capturing())
        if self.on:
            self.main = torch.cuda.current_stream(device)
            key = (device.index, self.main.cuda_stream)
            side = _SIDE_STREAMS.get(key)
            if side is None:
                side = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
            self.side = side

    def run(self, fn, *deps):
        if not self.on:
            return fn()
        self.side.wait_stream(self.main)
        for t in deps:
            if t is not None:
                t.record_stream(self.side)
        with torch.cuda.stream(self.side):
            return fn()

    def join(self, *produced):
        """The caller's stream waits for the side stream; `produced`: tensors allocated on the side stream that the caller's
        stream uses from here on."""
        if self.on:
            self.main.wait_stream(self.side)
            for t in produced:
                if t is not None:
                    t.record_stream(self.main)


def _packs_of(owner):
    if not BATCHED_WEIGHT_PACKS:
        return None
    ps = getattr(owner, "_step_packs", None)
    if ps is None:
        ps = ops.StepPacks()
        object.__setattr__(owner, "_step_packs", ps)      # not a parameter / buffer / sub-module
    return ps


def _taps(w: torch.Tensor, packs=None, flipped: bool = False):
    if w.dim() == 4:
        # dim='2d' block: the k x k stencil is the centre z-plane of a k^3 one (models/architectures/mednext.py
        # embed_stencil_2d); a temporary, so it does not join the per-step pack set
        from ..models.architectures.mednext import embed_stencil_2d
        return ops.packed_taps(embed_stencil_2d(w.detach().float()), flipped=flipped, packs=None), w.shape[-1]
    return ops.packed_taps(w, flipped=flipped, packs=packs), w.shape[-1]


def _mat(w: torch.Tensor) -> torch.Tensor:
    return w.detach().float().reshape(w.shape[0], w.shape[1]).contiguous()


def _f(p: Optional[torch.Tensor]):
    return None if p is None else p.detach().float().contiguous()


def _rows(t: torch.Tensor) -> int:
    return t.shape[1] * t.shape[2] * t.shape[3]


def _pw(x, w_mat, bias, *, c_out, out_dtype=None, transposed=False, packs=None, **kw):
    """y = pw_conv with this step's packed weights (they change at every optimizer step; `packs`: the model's StepPacks)."""
    dt = x.dtype if x.dtype == torch.bfloat16 or out_dtype != torch.bfloat16 else torch.bfloat16
    wdt = torch.bfloat16 if (x.dtype == torch.bfloat16 or out_dtype == torch.bfloat16) else torch.float32
    odt = out_dtype or x.dtype
    paired = wdt == torch.bfloat16 and ops.pw_conv_paired_supported(
        c_in=x.shape[-1], c_out=c_out, in_dtype=x.dtype, out_dtype=odt, act=kw.get("act", nat.ACT_NONE),
        gather=kw.get("gather", 0))
    N = x.shape[0]
    rows = kw.pop("rows", None) or x.numel() // (N * x.shape[-1])
    # deep levels (few voxel rows, wide channels): the LDS-tiled GEMM on plain row-major weights instead of the paired-row kernel, which
    # streams every weight fragment from L2 per 16-64 rows (TRAIN_GEMM_MAX_ROWS; same prologues / epilogues through pytc_pw_conv_fwd)
    if (paired and TRAIN_GEMM_MAX_ROWS and N * rows <= TRAIN_GEMM_MAX_ROWS and x.dtype == torch.bfloat16 and odt == torch.bfloat16
            and ops.pw_conv_rowmajor_supported(c_in=x.shape[-1], c_out=c_out, in_dtype=x.dtype, out_dtype=odt,
                                               act=kw.get("act", nat.ACT_NONE), gather=kw.get("gather", 0))):
        wp = ops.packed_rowmajor(w_mat, transposed=transposed, packs=packs)
        return ops.pw_conv(x, wp, bias, N=N, rows_per_sample=rows, c_in=x.shape[-1], c_out=c_out, out_dtype=odt, w_paired=2, **kw)
    wp = ops.packed_paired(w_mat, transposed=transposed, packs=packs) if paired else ops.pw_pack_weight(w_mat, wdt, transposed=transposed)
    y = ops.pw_conv(x, wp, bias, N=N, rows_per_sample=rows, c_in=x.shape[-1], c_out=c_out,
                    out_dtype=odt, w_paired=paired, **kw)
    return y


class PointwiseFn(torch.autograd.Function):
    """y = W x + b on channels-last rows (stem, output heads, task-head projections)."""

    @staticmethod
    def forward(ctx, x, weight, bias, transposed: bool, out_dtype, packs=None):
        w = _mat(weight)                              # (a, b) as stored
        c_out = w.shape[1] if transposed else w.shape[0]
        y = _pw(x, w, _f(bias), c_out=c_out, out_dtype=out_dtype, transposed=transposed, packs=packs)
        ctx.save_for_backward(x, weight)
        ctx.packs = packs
        ctx.meta = (transposed, bias is not None, c_out)
        return y.view(*x.shape[:-1], c_out)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        transposed, has_bias, c_out = ctx.meta
        w = _mat(weight)
        c_in = x.shape[-1]
        N = x.shape[0]
        rows = x.numel() // (N * c_in)
        dyc = dy.contiguous()
        dyx, xin = dyc, x
        if dyc.dtype != x.dtype and x.dtype in (torch.float32, torch.bfloat16):
            # one dtype for both GEMM operands: convert the SMALLER tensor.  The stem meets its fp32 one-channel input with a bf16
            # 32-channel gradient: rounding the input (what autocast does to a conv input) moves 34 MB, converting the gradient
            # moved 1.1 GB and doubled the bytes the weight-gradient kernel read
            if x.numel() < dyc.numel() and not ctx.needs_input_grad[0]:
                xin = x.to(dyc.dtype)
            else:
                dyx = dyc.to(x.dtype)
        dx = None
        if ctx.needs_input_grad[0]:
            # dX = dY . W : operator c_out -> c_in with matrix W^T
            dx = _pw(dyx, w, None, c_out=c_in, transposed=not transposed, packs=ctx.packs).view_as(x)
        dW, db = ops.pw_wgrad(xin.contiguous(), dyx, N=N, rows_per_sample=rows, c_in=c_in, c_out=c_out,
                              want_bias=has_bias)
        dW = dW.t().contiguous() if transposed else dW
        return (dx, torch.empty_like(weight).copy_(dW.reshape(weight.shape)), (db.to(weight.dtype) if has_bias else None),
                None, None, None)


def _dw_backward(kind, dt_, t, x, dy, w1, wres, taps, K, packs, do_res, has_res, dr, lane=None, dtc=None, norm_is_per_channel=False,
                 skip_add=None):
    """Depthwise-conv half of a block backward, shared by BlockFn and NormVariantBlockFn: from dt (gradient of the depthwise
    output) to dx, dW1 (channel-major (C, K^3): the reduction launch writes the parameter's layout), db1 and the gradients of the resampling residual conv.  Slot reductions join `dr`."""
    N, D, H, W, C = x.shape
    rows = _rows(t)
    c_out = dy.shape[-1]
    dwres_m = None
    dwres = dbres = None
    lane = lane or _WgradLane(torch.device("cpu"))
    if kind == "block":
        dW1, db1 = lane.run(lambda: ops.dw_wgrad(dt_.view_as(t), x, K=K, stride=1, defer=dr, channel_major=True), dt_, x)
        flipped, _ = _taps(w1, packs, flipped=True)             # correlation with the reversed stencil
        gt = dt_.view_as(t)
        if do_res and FUSED_RESIDUAL_DGRAD and ops.dwconv3d_res_supported(gt, K, 1):
            dx = ops.dwconv3d_res(gt, flipped, dy, K=K)        # dx = conv_reversed(dt) + dy in the conv kernel's epilogue
        else:
            dx, _ = ops.dwconv3d(gt, flipped, None, K=K, stride=1, stats=False, wide_range=True)
            if do_res:
                ops.add_(dx, dy)
    elif kind == "down":
        dW1, db1 = lane.run(lambda: ops.dw_wgrad(dt_.view_as(t), x, K=K, stride=2, defer=dr, channel_major=True), dt_, x)
        if skip_add is not None and tuple(skip_add.shape) == tuple(x.shape) and skip_add.dtype == dt_.dtype:
            dx = ops.dwconv3d_bwd_data(dt_.view_as(t), taps, (D, H, W), K=K, stride=2, add=skip_add.contiguous())
        else:
            dx = ops.dwconv3d_bwd_data(dt_.view_as(t), taps, (D, H, W), K=K, stride=2)
            if skip_add is not None:
                dx = dx + skip_add.to(dx.dtype)
        if has_res:
            xg = x[:, ::2, ::2, ::2, :].contiguous()
            dwres, dbres = lane.run(lambda: ops.pw_wgrad(xg, dy, N=N, rows_per_sample=rows, c_in=C, c_out=c_out, defer=dr), xg, dy)
            dxg = _pw(dy, _mat(wres), None, c_out=C, transposed=True, rows=rows, packs=packs).view_as(xg)
            dx[:, ::2, ::2, ::2, :] += dxg         # strided in-place add: only the 1/8 of dx the 1x1x1 stride-2 conv read
    else:
        if dtc is None:                                             # (given: the norm backward already wrote the compact grid)
            dtc = dt_.view_as(t)[:, 1:, 1:, 1:, :].contiguous()     # compact (2D-1)^3 grid of the transposed conv
        dW1, _ = lane.run(lambda: ops.dw_wgrad(x, dtc, K=K, stride=2, want_bias=False, defer=dr, channel_major=True), x, dtc)
        if norm_is_per_channel:
            # GroupNorm(C, C) subtracts the per-(sample, channel) mean over exactly the voxels the bias reaches: its gradient
            # sum_r dt[r][c] = rstd*gamma * (S1 - count*S1/count - S2/count * sum_r xhat) is zero identically; what a pass over dt
            # returned for it was rounding noise (as autograd's value is in the reference)
            db1 = torch.zeros((C,), dtype=torch.float32, device=x.device)
        else:
            db1 = ops.channel_stats(dtc).sum(1)[:, 0].sum(0)
        dx, _ = ops.dwconv3d(dtc, taps, None, K=K, stride=2, stats=False, wide_range=True)
        if has_res:
            drl = dy[:, 1::2, 1::2, 1::2, :].contiguous()       # positions fed by the transposed 1x1 conv
            dwres_m, _ = lane.run(lambda: ops.pw_wgrad(x, drl, N=N, rows_per_sample=D * H * W, c_in=C, c_out=c_out, want_bias=False,
                                                       defer=dr, in_major=True), x, drl)        # ConvTranspose layout (C_in, C_out)
            ops.add_(dx, _pw(drl, _mat(wres), None, c_out=C, transposed=False, packs=packs).view_as(dx))
    return dx, dW1, db1, dwres, dbres, dwres_m


def _rc_level(kind, dt, C, c_hid, c_out, fused: bool) -> int:
    """0: the block stores its hidden pre-activation; 1 / 2: it is rebuilt in the backward (2: with the GroupNorm sums in the same pass)."""
    if not (MIXER_BWD_RC and fused and kind == "block" and dt == torch.bfloat16 and not FUSED_TRAIN_MIXER_BWD):
        return 0
    return ops.mixer_bwd_rc_supported(C, c_hid, c_out, dt)


class BlockFn(torch.autograd.Function):
    """MedNeXt block / down block / up block.  `kind` in {"block", "down", "up"}.  `recompute` = the reference's
    `outside_block` activation checkpointing (mednext_models.py:386-393: torch.utils.checkpoint around every block): only the
    block input and the (N, 2, C) norm vectors are kept; the depthwise output and the hidden pre-activation are rebuilt by
    the same kernels at the start of the backward (bit-identical values, one extra block forward)."""

    @staticmethod
    def forward(ctx, x, skip, w1, b1, gamma, beta, w2, b2, w3, b3, wres, bres, kind: str, do_res: bool, eps: float,
                recompute: bool = False, packs=None, skip_box=None):
        ctx.packs = packs
        ctx.skip_box = skip_box if FUSE_SKIP_GRAD else None
        if ctx.skip_box is not None and kind == "down" and ctx.needs_input_grad[0]:
            skip_box.armed = True        # this block's backward will collect the skip gradient of its level
        y, t, ab, mr, hp, taps, K, count = BlockFn._core(x, skip, w1, b1, gamma, beta, w2, b2, w3, b3, wres, bres, kind,
                                                         do_res, eps, None, packs)
        keep = x.new_zeros(0)
        rc = hp is None                   # the hidden pre-activation was not stored: the backward rebuilds it (needs b2)
        ctx.save_for_backward(x, keep if recompute else t, ab, mr, keep if (recompute or rc) else hp, w1, gamma, w2, w3,
                              wres if wres is not None else keep, skip if (recompute and skip is not None) else keep,
                              b1 if (recompute and b1 is not None) else keep, b2 if (recompute or rc) else keep, b3 if recompute else keep,
                              bres if (recompute and bres is not None) else keep)
        ctx.meta = (kind, do_res, K, count, wres is not None, skip is not None, b1 is not None, bres is not None, recompute, eps,
                    b2 is not None, b3 is not None)
        ctx.taps = taps                  # derived from w1 (no gradient flows through it): reused by the backward
        return y

    @staticmethod
    def _core(x, skip, w1, b1, gamma, beta, w2, b2, w3, b3, wres, bres, kind, do_res, eps, ab_mr, packs=None):
        """The block's forward kernels.  ab_mr = (ab, mr) of an earlier identical call: the statistics pass is skipped
        (recomputation inside the backward)."""
        N, D, H, W, C = x.shape
        dt = x.dtype
        taps, K = _taps(w1, packs)
        if kind == "up":
            t, st = ops.dwconv3d(x, taps, _f(b1), K=K, transposed=True, stats=ab_mr is None)
            count = float((2 * D - 1) * (2 * H - 1) * (2 * W - 1))
        else:
            t, st = ops.dwconv3d(x, taps, _f(b1), K=K, stride=2 if kind == "down" else 1, stats=ab_mr is None)
            count = float(_rows(t))
        ab, mr = ops.groupnorm_finalize_mr(st, count, _f(gamma), _f(beta), eps) if ab_mr is None else ab_mr
        rows = _rows(t)
        c_hid, c_out = w2.shape[0], w3.shape[0]
        fused = (dt == torch.bfloat16 and b2 is not None and b3 is not None and ops.pw_mlp_supported(C, c_hid, c_out)
                 and FUSED_TRAIN_MIXER and N * rows >= FUSED_TRAIN_MIXER_MIN_ROWS and c_hid < FUSED_TRAIN_MIXER_MAX_HID)
        rc = _rc_level(kind, dt, C, c_hid, c_out, fused)
        if fused:
            # one launch: norm affine -> expand -> (store pre-activation hp) -> GELU -> project -> residual epilogue
            hp = None if rc else torch.empty((N, rows, c_hid), dtype=dt, device=x.device)
            w2p = ops.packed_paired(_mat(w2), packs=packs)
            # bf16 image: the hidden-storing training forward measured FASTER with the bf16 sigmoid-form GELU than with the
            # packed-fp16 one (357 vs 395 us at 32->64->32: the stored pre-activation is rounded to bf16 and read back first,
            # and the extra conversions cost the kernel its 4th wave per SIMD), unlike the inference mixers
            w3p = ops.packed_paired(_mat(w3), f16=bool(rc and RC_FWD_F16_GELU), packs=packs)
            mk = dict(N=N, rows_per_sample=rows, c_in=C, c_hid=c_hid, c_out=c_out, hidden_pre=hp, train_nostore=bool(rc))
        else:
            hp = _pw(t, _mat(w2), _f(b2), c_out=c_hid, ab=ab, rows=rows, packs=packs)              # pre-activation (saved)
        h, G = hp, dict(pre_act=nat.ACT_GELU)       # GELU runs in the operand prologue of the projecting GEMM
        res_low = None
        if kind == "block":
            if fused:
                y = ops.pw_mlp(t, ab, w2p, _f(b2), w3p, _f(b3), res=x if do_res else None,
                               res_mode=nat.RES_ADD if do_res else nat.RES_NONE, **mk)
            else:
                y = _pw(h, _mat(w3), _f(b3), c_out=c_out, rows=rows, res=x if do_res else None,
                        res_mode=nat.RES_ADD if do_res else nat.RES_NONE, packs=packs, **G)
        elif kind == "down":
            r = None
            if wres is not None:
                r = _pw(x, _mat(wres), _f(bres), c_out=c_out, rows=rows, gather=2, grid=(D, H, W), packs=packs)
            if fused:
                y = ops.pw_mlp(t, ab, w2p, _f(b2), w3p, _f(b3), res=r, res_mode=nat.RES_ADD if r is not None else nat.RES_NONE, **mk)
            else:
                y = _pw(h, _mat(w3), _f(b3), c_out=c_out, rows=rows, res=r,
                        res_mode=nat.RES_ADD if r is not None else nat.RES_NONE, packs=packs, **G)
        else:
            if wres is not None:
                res_low = _pw(x, _mat(wres), _f(bres), c_out=c_out, transposed=True, packs=packs)
            sk = skip if skip is not None else torch.zeros((N,) + tuple(t.shape[1:4]) + (c_out,), dtype=dt, device=x.device)
            if fused:
                y = ops.pw_mlp(t, ab, w2p, _f(b2), w3p, _f(b3), res=sk, res_mode=nat.RES_UPSAMPLE, grid=tuple(t.shape[1:4]),
                               res_low=res_low, res_bias=_f(bres) if wres is not None else None, **mk)
            else:
                y = _pw(h, _mat(w3), _f(b3), c_out=c_out, rows=rows, res=sk, res_mode=nat.RES_UPSAMPLE,
                        grid=tuple(t.shape[1:4]), res_low=res_low, res_bias=_f(bres) if wres is not None else None, packs=packs, **G)
        return y.view(N, *t.shape[1:4], c_out), t, ab, mr, hp, taps, K, count

    @staticmethod
    def backward(ctx, dy):
        x, t, ab, mr, hp, w1, gamma, w2, w3, wres, skip_s, b1_s, b2_s, b3_s, bres_s = ctx.saved_tensors
        kind, do_res, K, count, has_res, has_skip, has_b1, has_bres, recompute, eps, has_b2, has_b3 = ctx.meta
        packs = ctx.packs
        if recompute:
            # outside-block checkpointing: rebuild t and the hidden pre-activation with the forward's own kernels
            with torch.no_grad():
                _y, t, _ab, _mr, hp, _taps_, _K, _c = BlockFn._core(
                    x, skip_s if (has_skip and skip_s.numel()) else None, w1, b1_s if has_b1 else None, gamma, None, w2, b2_s, w3,
                    b3_s, wres if has_res else None, bres_s if has_bres else None, kind, do_res, eps, (ab, mr), packs)
            del _y
        N, D, H, W, C = x.shape
        dy = dy.contiguous()
        rows = _rows(t)
        c_hid, c_out = w2.shape[0], w3.shape[0]
        taps = ctx.taps
        dskip = dy if (kind == "up" and has_skip) else None
        box = ctx.skip_box
        if (dskip is not None and box is not None and box.armed and (dy.shape[-1] * dy.element_size()) % 16 == 0):
            box.grad, dskip = dy, None   # collected by the level's down block inside its data-gradient launch
        skip_add = None
        if kind == "down" and box is not None and box.grad is not None:
            skip_add, box.grad = box.grad, None
        dcore = dy
        if kind == "up":
            dcore = ops.copy_zero_front(dy)      # the padded front faces are not outputs of the mixer
        # every slot reduction of this block's gradients (dW3/db3, dW2/db2, the norm sums, dW1/db1, the residual conv) joins
        # ONE launch at the end (ops.DeferredReduce): ~5 tiny launches per block become 1, bit-identical results
        dr = ops.DeferredReduce()
        lane = _WgradLane(x.device)
        # ---- project: y = W3 h + b3
        fused_bwd = (dy.dtype == torch.bfloat16 and FUSED_TRAIN_MIXER_BWD and ops.pw_mlp_supported(c_out, c_hid, C))
        # round 5: the projecting conv's weight gradient and the data gradient behind the activation in ONE pass over (hp, dy)
        # (pytc_pw_wgrad_dgrad_partial: level-0 shapes)
        rc = hp is None or hp.numel() == 0
        wg_dg = (not rc and FUSED_WGRAD_DGRAD and not fused_bwd and hp.dim() == 3 and ops.pw_wgrad_dgrad_supported(c_hid, c_out, dy.dtype)
                 and hp.dtype == torch.bfloat16)
        dhp = None
        rc_gn = None
        if rc:
            # full-resolution block: hp rebuilt from t inside the pass that forms dhp, dW3 / db3 (and, level 2, the GroupNorm form's sums)
            lvl = _rc_level(kind, dy.dtype, C, c_hid, c_out, True)
            assert lvl >= 1, "the forward dropped the hidden pre-activation of a block the backward cannot rebuild"
            gn = (lvl >= 2 and NORM_STATS_FROM_WGRAD and not fused_bwd
                  and ops.pw_conv_paired_supported(c_in=c_hid, c_out=C, in_dtype=dy.dtype, out_dtype=dy.dtype))
            out = ops.mixer_bwd_rc(t.view(N, rows, C), ab, dcore.view(N, rows, c_out), ops.packed_paired(_mat(w2), packs=packs), _f(b2_s),
                                   ops.packed_paired(_mat(w3), transposed=True, packs=packs), N=N, rows_per_sample=rows, c=C, c_hid=c_hid,
                                   c_out=c_out, mean_rstd=mr if gn else None, w2=_mat(w2) if gn else None,
                                   gamma=_f(gamma) if gn else None, count=count if gn else 0.0, defer=dr)
            dW3, db3, dhp = out[:3]
            rc_gn = out[3:] if gn else None
        elif wg_dg:
            dW3, db3, dhp = ops.pw_wgrad_dgrad(hp, dcore.view(N, rows, c_out), ops.packed_paired(_mat(w3), transposed=True, packs=packs),
                                               N=N, rows_per_sample=rows, c_in=c_hid, c_out=c_out, defer=dr)
        else:
            dW3, db3 = lane.run(lambda: ops.pw_wgrad(hp, dcore, N=N, rows_per_sample=rows, c_in=c_hid, c_out=c_out, x_act=nat.ACT_GELU,
                                                     defer=dr), hp, dcore)
        # expand conv hp = W2 (a t + b) + b2.  NORM_STATS_FROM_WGRAD: its weight-gradient pass (against xhat, per-sample slots) also
        # yields the GroupNorm backward sums, and the data-gradient GEMM applies the norm backward to its own unrounded result in its
        # epilogue (RES_NORM_BWD): no statistics pass, no apply pass, dtn never stored.  Up blocks (cropped output) take the two-pass form.
        # (round 6: up blocks too -- dhp is zero on the padded front faces (dcore is), so the per-sample sums cover the transposed conv's
        # (2D-1)^3 outputs exactly, and the epilogue writes the compact grid that conv's backward reads, dropping the face rows)
        stats_from_wgrad = (NORM_STATS_FROM_WGRAD and not fused_bwd and (kind != "up" or UP_NORM_STATS_FROM_WGRAD)
                            and ops.pw_wgrad_groupnorm_supported(C, c_hid, dy.dtype)
                            and ops.pw_conv_paired_supported(c_in=c_hid, c_out=C, in_dtype=dy.dtype, out_dtype=dy.dtype))
        dtc = None
        if rc_gn is not None:
            dW2, db2, s, coef = rc_gn
            dt_ = _pw(dhp, _mat(w2), None, c_out=C, transposed=True, rows=rows, res=t.view(N, rows, C), res_mode=nat.RES_NORM_BWD,
                      res_bias=coef, packs=packs)
            del dhp
        elif fused_bwd:
            # both data-gradient GEMMs in one launch: dtn = W2^T ((W3^T dy) * gelu'(hp)); dhp comes back for wgrad2
            dtn, dhp = ops.pw_mlp_bwd(dcore.view(N, rows, c_out), hp, ops.packed_paired(_mat(w3), transposed=True, packs=packs),
                                      ops.packed_paired(_mat(w2), transposed=True, packs=packs), N=N, rows_per_sample=rows,
                                      c_in=C, c_hid=c_hid, c_out=c_out)
        elif dhp is None:
            # dhp = (W3^T dy) * gelu'(hp): the GELU derivative is the epilogue of the data-gradient GEMM
            dhp = _pw(dcore, _mat(w3), None, c_out=c_hid, transposed=True, rows=rows, res=hp, res_mode=nat.RES_GELU_BWD, packs=packs)
        if rc_gn is not None:
            pass
        elif stats_from_wgrad:
            dW2, db2, s, coef = ops.pw_wgrad_groupnorm(t, mr, ab, dhp, _mat(w2), _f(gamma), N=N, rows_per_sample=rows, c=C, c_hid=c_hid,
                                                       count=count, defer=dr)
            if kind == "up":
                gd, gh, gw = (int(v) for v in t.shape[1:4])
                dtc = torch.empty((N, gd - 1, gh - 1, gw - 1, C), dtype=t.dtype, device=t.device)
                _pw(dhp, _mat(w2), None, c_out=C, transposed=True, rows=rows, res=t.view(N, rows, C), res_mode=nat.RES_NORM_BWD,
                    res_bias=coef, packs=packs, grid=(gd, gh, gw), y=dtc)
                dt_ = None
            else:
                dt_ = _pw(dhp, _mat(w2), None, c_out=C, transposed=True, rows=rows, res=t.view(N, rows, C), res_mode=nat.RES_NORM_BWD,
                          res_bias=coef, packs=packs)
            del dhp
        else:
            dW2, db2 = lane.run(lambda: ops.pw_wgrad(t, dhp, N=N, rows_per_sample=rows, c_in=C, c_out=c_hid, ab=ab, defer=dr), t, dhp, ab)
            if not fused_bwd:
                dtn = _pw(dhp, _mat(w2), None, c_out=C, transposed=True, rows=rows, packs=packs)
            del dhp
            # ---- GroupNorm(C, C)
            if kind == "up" and ops.norm_bwd_apply_supported(dtn):
                # statistics pass, then an apply pass that writes the compact (2D-1)^3 grid the transposed conv's backward reads (the
                # same kernels and sums as pytc_norm_bwd; no crop copy of the 64-channel tensor afterwards)
                s = ops.norm_bwd_stats(dtn, t, mr)
                dtc = ops.norm_bwd_apply(dtn, t, mr, _f(gamma), s, count=count, crop_grid=tuple(t.shape[1:4]))
                dt_ = None
            else:
                dt_, s = ops.norm_bwd(dtn, t, mr, _f(gamma), count=count)
            del dtn
        ssum = torch.empty((2, C), dtype=torch.float32, device=x.device)
        dr.add(s, ssum, 2 * C, N)        # (N, 2, C) -> (2, C): the samples are the "slots"
        dgamma, dbeta = ssum[1], ssum[0]
        # ---- depthwise conv
        dx, dW1, db1, dwres, dbres, dwres_m = _dw_backward(kind, dt_, t, x, dy, w1, wres, taps, K, packs, do_res, has_res, dr, lane,
                                                           dtc=dtc, norm_is_per_channel=True, skip_add=skip_add)
        lane.join()
        dr.flush()                       # all weight / bias / norm gradients of the block are final from here on
        if kind == "up" and has_res:
            dwres = dwres_m
            dbres = db3.clone()                                     # bias reaches every interior voxel exactly once
        # gradients in the parameter's own (contiguous) strides, which DDP's bucket views expect: a view when the kernel
        # output is already laid out that way (1x1x1 weights, norm vectors), a copy only for the transposed ones
        def g(v, like):
            if v is None:
                return None
            if v.is_contiguous() and like.is_contiguous():
                return v.view(like.shape)
            return torch.empty_like(like).copy_(v.reshape(like.shape))
        if w1.dim() == 4:                # dim='2d': only the centre z-plane of the embedded stencil is a parameter
            dW1 = dW1.view(C, 1, K, K, K)[:, :, K // 2].contiguous()
        return (dx, dskip, g(dW1, w1), (db1.to(w1.dtype) if has_b1 else None), g(dgamma, gamma),
                g(dbeta, gamma), g(dW2, w2), (db2.to(w2.dtype) if has_b2 else None), g(dW3, w3),
                (db3.to(w3.dtype) if has_b3 else None),
                (g(dwres, wres) if has_res else None), (dbres.to(w3.dtype) if (has_res and has_bres) else None),
                None, None, None, None, None, None)


def _grad_like(v, like):
    """A gradient in the parameter's own (contiguous) strides: a view when the kernel output is already laid out that way."""
    if v is None:
        return None
    if v.is_contiguous() and like.is_contiguous():
        return v.view(like.shape)
    return torch.empty_like(like).copy_(v.reshape(like.shape))


def _grn_coeffs(h, grid, kind: str, gamma_vec: torch.Tensor):
    """GRN statistics of the activated hidden tensor h (N, rows, C): gx = ||h||_2 over the block's spatial support per
    (n, c), nx = gx / (mean_c gx + 1e-6), A = gamma * nx + 1 (upstream MedNeXtBlock.forward, grn branch).  For the up block
    the padded front faces are not part of the support.  -> gx, nx, A (each (N, C) fp32)."""
    N, c_hid = h.shape[0], h.shape[-1]
    hv = h.view(N, *grid, c_hid)
    core = hv[:, 1:, 1:, 1:].contiguous() if kind == "up" else hv
    gx = ops.channel_stats(core)[:, :, 1].sum(1).clamp_min(0).sqrt()
    nx = gx / (gx.mean(1, keepdim=True) + 1e-6)
    return gx, nx, (gamma_vec.view(1, c_hid) * nx + 1.0).contiguous()


class NormVariantBlockFn(torch.autograd.Function):
    """MedNeXt block / down block / up block with norm_type='layer' (per-voxel LayerNorm over C) and / or grn=True (global
    response normalisation of the expanded tensor) -- the two constructor variants of mednext_models.py:449-463 that
    BlockFn's fused GroupNorm schedule does not cover.  Un-fused schedule:

      forward   dwconv(+stats for GroupNorm) -> [layernorm_rows] -> 1x1 expand (pre-activation hp saved)
                -> grn: h = gelu(hp), h2 = h * (gamma * nx + 1) + beta (two elementwise passes + one statistics pass)
                -> 1x1 project (GELU in its operand prologue when there is no GRN) + the block's residual epilogue
      backward  the mirrored sequence; LayerNorm: pytc_layernorm_rows_bwd; GRN: (sum dh2, sum dh2 * h) per (n, c) from
                pytc_norm_bwd_stats, the tiny (N, C) coefficient algebra on the host side, then pytc_grn_bwd_apply, which
                also carries the GELU derivative.

    Saved: x, t (depthwise output), hp, the norm vectors and gx (N, C)."""

    @staticmethod
    def forward(ctx, x, skip, w1, b1, gamma, beta, w2, b2, w3, b3, wres, bres, grn_g, grn_b, kind: str, do_res: bool,
                eps: float, is_ln: bool, packs=None):
        N, D, H, W, C = x.shape
        dt = x.dtype
        has_grn = grn_g is not None
        taps, K = _taps(w1, packs)
        if kind == "up":
            t, st = ops.dwconv3d(x, taps, _f(b1), K=K, transposed=True, stats=not is_ln)
            count = float((2 * D - 1) * (2 * H - 1) * (2 * W - 1))
        else:
            t, st = ops.dwconv3d(x, taps, _f(b1), K=K, stride=2 if kind == "down" else 1, stats=not is_ln)
            count = float(_rows(t))
        rows = _rows(t)
        grid = tuple(t.shape[1:4])
        c_hid, c_out = w2.shape[0], w3.shape[0]
        keep = x.new_zeros(0)
        if is_ln:
            tn, ab, mr = ops.layernorm_rows(t, _f(gamma), _f(beta), eps), None, None
        else:
            ab, mr = ops.groupnorm_finalize_mr(st, count, _f(gamma), _f(beta), eps)
            tn = t
        hp = _pw(tn, _mat(w2), _f(b2), c_out=c_hid, ab=ab, rows=rows, packs=packs)                  # pre-activation (saved)
        del tn
        gx = None
        if has_grn:
            h = ops.gelu(hp)
            gx, _nx, A = _grn_coeffs(h, grid, kind, _f(grn_g).reshape(-1))
            ab2 = torch.stack([A, _f(grn_b).reshape(1, c_hid).expand(N, c_hid)], 1).contiguous()
            hin = ops.affine_act(h.view(N, *grid, c_hid), ab2, nat.ACT_NONE, 0.0).view(N, rows, c_hid)
            del h
            G = {}
        else:
            hin, G = hp, dict(pre_act=nat.ACT_GELU)
        if kind == "block":
            y = _pw(hin, _mat(w3), _f(b3), c_out=c_out, rows=rows, res=x if do_res else None,
                    res_mode=nat.RES_ADD if do_res else nat.RES_NONE, packs=packs, **G)
        elif kind == "down":
            r = None
            if wres is not None:
                r = _pw(x, _mat(wres), _f(bres), c_out=c_out, rows=rows, gather=2, grid=(D, H, W), packs=packs)
            y = _pw(hin, _mat(w3), _f(b3), c_out=c_out, rows=rows, res=r,
                    res_mode=nat.RES_ADD if r is not None else nat.RES_NONE, packs=packs, **G)
        else:
            res_low = None
            if wres is not None:
                res_low = _pw(x, _mat(wres), _f(bres), c_out=c_out, transposed=True, packs=packs)
            sk = skip if skip is not None else torch.zeros((N,) + grid + (c_out,), dtype=dt, device=x.device)
            y = _pw(hin, _mat(w3), _f(b3), c_out=c_out, rows=rows, res=sk, res_mode=nat.RES_UPSAMPLE, grid=grid,
                    res_low=res_low, res_bias=_f(bres) if wres is not None else None, packs=packs, **G)
        ctx.save_for_backward(x, t, hp, ab if ab is not None else keep, mr if mr is not None else keep,
                              gx if gx is not None else keep, w1, gamma, beta if beta is not None else keep, w2, w3,
                              wres if wres is not None else keep, grn_g if has_grn else keep, grn_b if has_grn else keep)
        ctx.meta = (kind, do_res, K, count, wres is not None, skip is not None, b1 is not None, bres is not None, eps,
                    b2 is not None, b3 is not None, is_ln, has_grn)
        ctx.taps, ctx.packs = taps, packs
        return y.view(N, *grid, c_out)

    @staticmethod
    def backward(ctx, dy):
        x, t, hp, ab, mr, gx, w1, gamma, beta, w2, w3, wres, grn_g, grn_b = ctx.saved_tensors
        kind, do_res, K, count, has_res, has_skip, has_b1, has_bres, eps, has_b2, has_b3, is_ln, has_grn = ctx.meta
        packs, taps = ctx.packs, ctx.taps
        N, D, H, W, C = x.shape
        dy = dy.contiguous()
        rows = _rows(t)
        grid = tuple(t.shape[1:4])
        c_hid, c_out = w2.shape[0], w3.shape[0]
        dskip = dy if (kind == "up" and has_skip) else None
        dcore = dy
        if kind == "up":
            dcore = ops.copy_zero_front(dy)      # the padded front faces are not outputs of the mixer
        dr = ops.DeferredReduce()
        dgrn_g = dgrn_b = None
        if has_grn:
            # h2 = h * A + beta_grn with h = gelu(hp): rebuilt by the forward's own kernels
            gvec = _f(grn_g).reshape(-1)
            h = ops.gelu(hp)
            gx_, nx, A = _grn_coeffs(h, grid, kind, gvec)
            ab2 = torch.stack([A, _f(grn_b).reshape(1, c_hid).expand(N, c_hid)], 1).contiguous()
            h2 = ops.affine_act(h.view(N, *grid, c_hid), ab2, nat.ACT_NONE, 0.0).view(N, rows, c_hid)
            dW3, db3 = ops.pw_wgrad(h2, dcore, N=N, rows_per_sample=rows, c_in=c_hid, c_out=c_out, defer=dr)
            del h2
            dh2 = _pw(dcore, _mat(w3), None, c_out=c_hid, transposed=True, rows=rows, packs=packs)
            # per (n, c): Q = sum dh2, P = sum dh2 * h (xhat = h with mean 0, rstd 1); dh2 is zero on an up block's padded faces
            unit = torch.stack([torch.zeros_like(gx_), torch.ones_like(gx_)], 1).contiguous()
            s = ops.norm_bwd_stats(dh2, h, unit)
            del h
            Q, P = s[:, 0], s[:, 1]
            dgrn_b = Q.sum(0)
            dgrn_g = (nx * P).sum(0)
            dnx = gvec.view(1, c_hid) * P
            den = gx_.mean(1, keepdim=True) + 1e-6
            dgx = dnx / den - (dnx * gx_).sum(1, keepdim=True) / (c_hid * den * den)
            Bc = torch.where(gx_ > 0, dgx / gx_.clamp_min(1e-30), torch.zeros_like(dgx)).contiguous()   # d||h|| / dh = h / ||h||
            dhp = ops.grn_bwd_apply(dh2, hp, A, Bc)
            del dh2
            if kind == "up":         # h on the padded faces is not part of the GRN support: no gradient reaches it
                dv = dhp.view(N, *grid, c_hid)
                dv[:, 0] = 0
                dv[:, :, 0] = 0
                dv[:, :, :, 0] = 0
        else:
            dW3, db3 = ops.pw_wgrad(hp, dcore, N=N, rows_per_sample=rows, c_in=c_hid, c_out=c_out, x_act=nat.ACT_GELU, defer=dr)
            dhp = _pw(dcore, _mat(w3), None, c_out=c_hid, transposed=True, rows=rows, res=hp, res_mode=nat.RES_GELU_BWD, packs=packs)
        # ---- expand: hp = W2 norm(t) + b2
        if is_ln:
            tn = ops.layernorm_rows(t, _f(gamma), _f(beta) if beta.numel() else None, eps)
            dW2, db2 = ops.pw_wgrad(tn, dhp, N=N, rows_per_sample=rows, c_in=C, c_out=c_hid, defer=dr)
            del tn
        else:
            dW2, db2 = ops.pw_wgrad(t, dhp, N=N, rows_per_sample=rows, c_in=C, c_out=c_hid, ab=ab, defer=dr)
        dtn = _pw(dhp, _mat(w2), None, c_out=C, transposed=True, rows=rows, packs=packs)
        del dhp
        # ---- norm
        ssum = torch.empty((2, C), dtype=torch.float32, device=x.device)
        if is_ln:
            dt_, part = ops.layernorm_rows_bwd(dtn, t, _f(gamma), eps)
            dr.add(part, ssum, 2 * C, part.shape[0], keep=part)
        else:
            dt_, s = ops.norm_bwd(dtn, t, mr, _f(gamma), count=count)
            dr.add(s, ssum, 2 * C, N, keep=s)
        dgamma, dbeta = ssum[1], ssum[0]
        del dtn
        dx, dW1, db1, dwres, dbres, dwres_m = _dw_backward(kind, dt_, t, x, dy, w1, wres, taps, K, packs, do_res, has_res, dr)
        dr.flush()
        if kind == "up" and has_res:
            dwres = dwres_m
            dbres = db3.clone()                                     # bias reaches every interior voxel exactly once
        if w1.dim() == 4:                # dim='2d': only the centre z-plane of the embedded stencil is a parameter
            dW1 = dW1.view(C, 1, K, K, K)[:, :, K // 2].contiguous()
        g = _grad_like
        return (dx, dskip, g(dW1, w1), (db1.to(w1.dtype) if has_b1 else None), g(dgamma, gamma),
                (g(dbeta, gamma) if beta.numel() else None), g(dW2, w2), (db2.to(w2.dtype) if has_b2 else None), g(dW3, w3),
                (db3.to(w3.dtype) if has_b3 else None),
                (g(dwres, wres) if has_res else None), (dbres.to(w3.dtype) if (has_res and has_bres) else None),
                (g(dgrn_g, grn_g) if has_grn else None), (g(dgrn_b, grn_b) if has_grn else None),
                None, None, None, None, None)


def _block(m, x, skip=None, recompute: bool = False, packs=None, skip_box=None):
    is_ln = not isinstance(m.norm, nn.GroupNorm)
    if is_ln and type(m.norm).__name__ != "_ChannelLayerNorm":
        raise NotImplementedError(f"no training kernels for MedNeXt norm module {type(m.norm).__name__}")
    res = getattr(m, "res_conv", None) if getattr(m, "resample_do_res", False) else None
    flat_up = m.dim == "2d" and m.kind == "up"
    if m.dim == "2d":
        if x.shape[1] != 1:
            raise ValueError(f"dim='2d' MedNeXt blocks take depth-1 volumes (N, 1, H, W, C), got {tuple(x.shape)}")
        if flat_up and skip is not None:
            # the 3-D up block doubles the depth too: plane 0 of its depth-2 grid is the padded face, plane 1 the 2-D answer
            skip = torch.cat([torch.zeros_like(skip), skip], dim=1)
    if m.grn or is_ln:
        # norm_type='layer' / grn=True: the un-fused variant schedule (no block-level recomputation: `outside_block`
        # checkpointing keeps its activations here)
        y = NormVariantBlockFn.apply(x, skip, m.conv1.weight, m.conv1.bias, m.norm.weight, m.norm.bias, m.conv2.weight,
                                     m.conv2.bias, m.conv3.weight, m.conv3.bias, None if res is None else res.weight,
                                     None if res is None else res.bias, m.grn_gamma if m.grn else None,
                                     m.grn_beta if m.grn else None, m.kind, bool(m.do_res), float(m.norm.eps), is_ln, packs)
    else:
        y = BlockFn.apply(x, skip, m.conv1.weight, m.conv1.bias, m.norm.weight, m.norm.bias, m.conv2.weight,
                          m.conv2.bias, m.conv3.weight, m.conv3.bias, None if res is None else res.weight,
                          None if res is None else res.bias, m.kind, bool(m.do_res), float(m.norm.eps), bool(recompute), packs,
                          None if (flat_up or m.dim == "2d") else skip_box)
    return y[:, 1:2].contiguous() if flat_up else y


def saved_activation_bytes(trunk, in_shape, compute_dtype: torch.dtype) -> int:
    """Bytes BlockFn keeps per training forward without checkpointing (block input + depthwise output + hidden
    pre-activation of every block), from the trunk's widths and the (N, D, H, W, C) input shape."""
    N, D, H, W = (int(v) for v in in_shape[:4])
    esz = 2 if compute_dtype == torch.bfloat16 else 4
    total = 0
    for name, mod in trunk.named_modules():
        if hasattr(mod, "conv2") and hasattr(mod, "kind"):
            lvl_vox = None
            c_in, c_hid = mod.conv2.weight.shape[1], mod.conv2.weight.shape[0]
            # the level of a block follows from its width: width = n_channels * 2^level
            lvl = max(0, (c_in // trunk.stem.weight.shape[0]).bit_length() - 1)
            if mod.kind == "up":
                lvl -= 1
            elif mod.kind == "down":
                lvl += 1
            lvl_vox = N * max(1, D >> lvl) * (H >> lvl) * (W >> lvl)     # (dim='2d': D stays 1)
            total += lvl_vox * (2 * c_in + c_hid) * esz
    return int(total)


def use_block_recompute(trunk, x_cl: torch.Tensor, compute_dtype: torch.dtype) -> bool:
    """`outside_block` checkpointing policy: 'always' / 'never', or 'auto' (default) = recompute only when the activations
    this forward would keep exceed half of the free HBM.  MedNeXt-S at 4 x 112^3 keeps ~25 GB of 288 GB: with 'auto' the flag
    that the reference's Lucchi++ config sets (mito_lucchi++.yaml:22) costs nothing here, and MedNeXt-L at large patches
    still trains."""
    if not getattr(trunk, "outside_block_checkpointing", False):
        return False
    policy = str(getattr(trunk, "checkpoint_policy", "auto")).lower()
    if policy in ("always", "never"):
        return policy == "always"
    if policy != "auto":
        raise ValueError(f"MedNeXt.checkpoint_policy must be 'auto', 'always' or 'never', got {policy!r}")
    free, _total = torch.cuda.mem_get_info(x_cl.device)
    return saved_activation_bytes(trunk, x_cl.shape, compute_dtype) > 0.5 * free


def mednext_train_features(trunk, x_cl: torch.Tensor, compute_dtype: torch.dtype):
    """Differentiable trunk up to the full-resolution features: -> (features, [bottleneck, dec_3, dec_2, dec_1])."""
    rc = use_block_recompute(trunk, x_cl, compute_dtype)
    packs = _packs_of(trunk)
    if packs is not None:
        packs.refresh()          # ONE launch rebuilds every weight image / stencil whose parameter changed since the last step
    x = PointwiseFn.apply(x_cl, trunk.stem.weight, trunk.stem.bias, False, compute_dtype, packs)
    skips = []
    boxes = [_SkipBox() if FUSE_SKIP_GRAD else None for _ in range(4)]      # one mailbox per level: up block -> down block (skip gradient)
    for lvl in range(4):
        for blk in getattr(trunk, f"enc_block_{lvl}"):
            x = _block(blk, x, recompute=rc, packs=packs)
        skips.append(x)
        x = _block(getattr(trunk, f"down_{lvl}"), x, recompute=rc, packs=packs, skip_box=boxes[lvl])
    for blk in trunk.bottleneck:
        x = _block(blk, x, recompute=rc, packs=packs)
    feats = [x]
    for lvl in (3, 2, 1, 0):
        x = _block(getattr(trunk, f"up_{lvl}"), x, skip=skips[lvl], recompute=rc, packs=packs, skip_box=boxes[lvl])
        for blk in getattr(trunk, f"dec_block_{lvl}"):
            x = _block(blk, x, recompute=rc, packs=packs)
        if lvl:
            feats.append(x)
    return x, feats


def mednext_train_forward(trunk, x_cl: torch.Tensor, compute_dtype: torch.dtype):
    """Differentiable forward of the MedNeXt trunk on channels-last input (N,D,H,W,C_in) fp32.
    Returns fp32 channels-last logits, or the list [out, ds_1..ds_4] with deep supervision."""
    x, feats = mednext_train_features(trunk, x_cl, compute_dtype)
    packs = _packs_of(trunk)
    head = lambda ft, i: PointwiseFn.apply(ft, getattr(trunk, f"out_{i}").conv_out.weight,
                                           getattr(trunk, f"out_{i}").conv_out.bias, True, torch.float32, packs)
    out = head(x, 0)
    if not trunk.do_ds:
        return out
    ds = [head(ft, h) for ft, h in zip(feats, (4, 3, 2, 1))]
    return [out, ds[3], ds[2], ds[1], ds[0]]


def mednext_multihead_train_forward(wrapper, x_cl: torch.Tensor, compute_dtype: torch.dtype):
    """Differentiable forward of MedNeXtMultiHeadWrapper (mednext_models.py:197-273): shared trunk features, then per
    named head an optional 1x1 in-projection, its MedNeXt blocks and the 1x1 out-projection.  -> {head: fp32 NDHWC logits}."""
    feat, _ = mednext_train_features(wrapper.model, x_cl, compute_dtype)
    outs = {}
    for name, head in wrapper.heads.items():
        x = feat
        if not isinstance(head.input_projection, nn.Identity):
            x = PointwiseFn.apply(x, head.input_projection.weight, head.input_projection.bias, False, compute_dtype,
                                  _packs_of(wrapper.model))
        if not isinstance(head.blocks, nn.Identity):
            for blk in head.blocks:
                x = _block(blk, x, packs=_packs_of(wrapper.model))
        outs[name] = PointwiseFn.apply(x, head.projection.weight, head.projection.bias, False, torch.float32,
                                       _packs_of(wrapper.model))
    return outs


__all__ = ["PointwiseFn", "BlockFn", "mednext_train_forward", "mednext_train_features", "mednext_multihead_train_forward"]
