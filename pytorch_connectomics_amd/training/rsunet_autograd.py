"""RSUNet training on HIP kernels: autograd Functions for the building blocks of
models/architectures/rsunet.py (reference rsunet.py:73-259, differentiated by torch autograd there).

The training forward is the un-fused schedule (the inference path keeps its fused kernels):
    NormAct      statistics -> finalize (affine + mean/rstd) -> affine_act            (activated tensor is materialised)
    Conv3d       implicit-GEMM forward kernel (+ residual epilogue)
    MaxPool3d / BilinearUp3d / 1x1 projection
and every backward piece is a HIP kernel: conv data gradient = the forward kernel with flipped / transposed weights,
`conv3d_wgrad`, `act_bwd`, `norm_bwd_stats` + `norm_bwd_apply_general` (GroupNorm / InstanceNorm / BatchNorm share one
formula, only the statistics group differs), `maxpool3d_bwd`, `dwconv3d_generic`.  The (N, 2, C)-sized combinations of
statistics into group means are torch ops on tiny tensors.  Correctness-first kernels (VALU weight gradient).
"""
from __future__ import annotations

import os

import torch

from .. import _native as nat
from .. import hip_ops as ops

_ACT = {"relu": nat.ACT_RELU, "leakyrelu": nat.ACT_LEAKY, "prelu": nat.ACT_LEAKY, "elu": nat.ACT_ELU}


def _f(p):
    return None if p is None else p.detach().float().reshape(-1).contiguous()


def _rows(x):
    return x.numel() // (x.shape[0] * x.shape[-1])


# The kernels take the PReLU slope by value.  Reading a device scalar costs a stream synchronisation, so the values are
# cached per parameter version and a model can refresh all of its (stale) slopes with ONE transfer per forward.
_PRELU_VALUES: dict = {}      # id(weight) -> (weakref to the weight, _version, data_ptr, slope)


def _prelu_hit(w: torch.Tensor):
    """The cached slope of THIS tensor object at THIS version and storage, else None.  id() alone is not an identity: a
    collected parameter's id is handed to the next allocation, so the entry keeps a weak reference and is only valid while
    that reference still points at `w`."""
    hit = _PRELU_VALUES.get(id(w))
    if hit is None or hit[0]() is not w or hit[1] != w._version or hit[2] != w.data_ptr():
        return None
    return hit[3]


def _prelu_put(w: torch.Tensor, value: float) -> None:
    import weakref
    key = id(w)
    _PRELU_VALUES[key] = (weakref.ref(w, lambda _r, _k=key: _PRELU_VALUES.pop(_k, None)), w._version, w.data_ptr(), float(value))


def prelu_value(w: torch.Tensor) -> float:
    hit = _prelu_hit(w)
    if hit is None:
        hit = float(w.detach().reshape(-1)[0])
        _prelu_put(w, hit)
    return hit


def prefetch_prelu(weights) -> None:
    stale = [w for w in weights if _prelu_hit(w) is None]
    if not stale:
        return
    vals = torch.stack([w.detach().reshape(-1)[0].float() for w in stale]).tolist()      # one device -> host copy
    for w, v in zip(stale, vals):
        _prelu_put(w, v)


# conv-weight images (forward, data gradient, strided / transposed forms) through ops.CONV_PACKS: after the first step every image
# of the model is rebuilt by ONE launch at the start of the training forward (refresh_conv_packs) instead of one launch per conv and
# direction; False: one pack launch per use
BATCHED_CONV_PACKS = True


def _packed(weight, layout: str, dtype, pad_to=None):
    """pad_to = (C_out, C_in) of the conv the image is FOR when that conv runs on channel-padded activations (ops.pad_channels)."""
    if BATCHED_CONV_PACKS:
        return ops.CONV_PACKS.get(weight, layout, dtype, pad_to)
    own = ops._conv_layout_rule(weight.shape, layout)[:2]
    return ops._conv_pack_single(weight.detach().float().contiguous(), layout, dtype,
                                 None if (pad_to is None or tuple(pad_to) == tuple(own)) else tuple(pad_to))


def _padded_vec(v, n: int):
    """fp32 vector `v` (or None) zero-extended to n entries: per-channel parameters next to channel-padded activations."""
    if v is None or v.numel() == n:
        return v
    return torch.nn.functional.pad(v, (0, n - v.numel()))


def refresh_conv_packs() -> None:
    if BATCHED_CONV_PACKS:
        ops.CONV_PACKS.refresh()


# activation derivative + norm backward without the intermediate dt tensor (pytc_act_norm_bwd_stats / _apply): the same dt values as
# the three passes act_bwd -> norm_bwd_stats -> norm_bwd_apply_general (rounded where they were stored), 5 instead of 8 tensor-sized
# memory passes
FUSED_ACT_NORM_BWD = True


class NormActFn(torch.autograd.Function):
    """a = act(norm(x)) on channels-last x.  kind in {none, group, instance, batch}."""

    @staticmethod
    def forward(ctx, x, gamma, beta, prelu_w, kind: str, groups: int, eps: float, act_kind: str, act_prm: float, bn):
        N, C = x.shape[0], x.shape[-1]
        rows = _rows(x)
        act = _ACT[act_kind]
        prm = prelu_value(prelu_w) if act_kind == "prelu" else float(act_prm)
        ab = mr = None
        mode = kind
        # x may carry alignment padding (ops.pad_channels): the norm's own channel count is its parameters'; the all-zero tail gets
        # the affine (0, 0) and stays zero through the activation
        c_real = int(gamma.numel()) if gamma is not None else (int(bn.num_features) if bn is not None else C)
        if kind in ("group", "instance"):
            st = ops.channel_stats(x)
            g = groups if kind == "group" else c_real
            ab, mr = ops.norm_finalize_groups_mr(st, rows, _f(gamma), _f(beta), eps, g, c_real // g if c_real != C else 0)
        elif kind == "batch":
            if bn.training and (not bn.track_running_stats or bn.momentum is not None):
                # statistics pass, then ONE launch: batch affine for every sample, running-buffer blend, counter increment
                st = ops.channel_stats(x)
                track = bn.track_running_stats
                ab, mr = ops.bn_train_finalize(st, float(N * rows), _f(gamma), _f(beta), eps, float(bn.momentum or 0.0),
                                               bn.running_mean if track else None, bn.running_var if track else None,
                                               bn.num_batches_tracked if track else None, N)
                if track:   # written through raw pointers: bump the version counters the eval-mode affine cache keys on
                    torch.autograd.graph.increment_version([bn.running_mean, bn.running_var, bn.num_batches_tracked])
            elif bn.training:          # momentum=None: cumulative moving average needs the counter's value on the host
                st = ops.channel_stats(x)
                st1 = st.reshape(1, st.shape[0] * st.shape[1], 2, C)
                ab1, mr1 = ops.norm_finalize_groups_mr(st1, N * rows, _f(gamma), _f(beta), eps, c_real, 1 if c_real != C else 0)
                ab, mr = ab1.expand(N, 2, C).contiguous(), mr1.expand(N, 2, C).contiguous()
                if bn.track_running_stats:
                    with torch.no_grad():
                        bn.num_batches_tracked += 1
                        mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                        ops.bn_update_running(mr1 if c_real == C else mr1[:, :, :c_real].contiguous(), bn.running_mean,
                                              bn.running_var, float(N * rows), eps, mom)
                        # written through raw pointers: bump the version counters the eval-mode affine cache keys on
                        torch.autograd.graph.increment_version([bn.running_mean, bn.running_var])
            else:
                a = gamma.detach().float() / torch.sqrt(bn.running_var.float() + eps)
                b = beta.detach().float() - bn.running_mean.float() * a
                ab = torch.stack([_padded_vec(a, C), _padded_vec(b, C)], 0).unsqueeze(0).expand(N, 2, C).contiguous()
                mode = "batch_eval"
        out = ops.affine_act(x, ab, act, prm)
        ctx.save_for_backward(x, ab if ab is not None else x.new_zeros(0), mr if mr is not None else x.new_zeros(0),
                              gamma if gamma is not None else x.new_zeros(0))
        ctx.meta = (mode, groups, act, prm, act_kind == "prelu", gamma is not None, beta is not None)
        return out

    @staticmethod
    def backward(ctx, da):
        x, ab, mr, gamma = ctx.saved_tensors
        mode, groups, act, prm, is_prelu, has_g, has_b = ctx.meta
        N, C = x.shape[0], x.shape[-1]
        rows = _rows(x)
        ab_ = ab if ab.numel() else None
        da = da.contiguous()
        # channel-padded activations: the group algebra sees the norm's own channels (cpg), kernels that index gamma per channel
        # of x get it zero-extended, the parameter gradients are the leading c_real entries
        c_real = int(gamma.numel()) if has_g else C
        cpg = (c_real // groups) if (mode == "group" and c_real != C) else 0
        grp = 0 if mode == "batch" else (groups if mode == "group" else c_real)
        # the fused pair below takes the norm's own gamma next to padded activations (norm_bwd_means with groups reads the real channels only,
        # act_norm_bwd_apply_cg zero-extends in the kernel): no padded copy per backward (a fill + a copy launch, 36 times per RSUNet step)
        unpadded_ok = FUSED_ACT_NORM_BWD and grp > 0 and mode in ("group", "instance") and da.dtype == x.dtype and ops.act_norm_bwd_supported(x)
        g32 = (_f(gamma) if unpadded_ok else _padded_vec(_f(gamma), C)) if has_g else None
        if mode == "instance" and c_real != C:
            cpg = 1
        cast = lambda v, like: None if v is None else v[:like.numel()].to(like.dtype).reshape(like.shape)      # noqa: E731
        if FUSED_ACT_NORM_BWD and mode in ("group", "instance", "batch") and da.dtype == x.dtype and ops.act_norm_bwd_supported(x):
            # dt = da * act'(t) is recomputed from (da, x) inside the statistics pass and the apply pass: never stored
            s, p = ops.act_norm_bwd_stats(da, x, ab_, mr, act, prm, want_prelu=is_prelu)
            dprelu = p.sum().reshape(1) if is_prelu else None
            M, dgamma, dbeta = ops.norm_bwd_means(s, g32, grp, rows, want_gamma=has_g, want_beta=has_b, cpg=cpg)
            dx = ops.act_norm_bwd_apply(da, x, ab_, mr, g32, M, act, prm)
            return (dx, cast(dgamma, gamma) if has_g else None, cast(dbeta, gamma) if has_b else None, dprelu, None, None, None,
                    None, None, None)
        dt, dp = ops.act_bwd(da, x, ab_, act, prm, want_prelu=is_prelu)
        dprelu = None
        if is_prelu:
            dprelu = ops.channel_stats(dp)[:, :, 0].sum().reshape(1)
        dgamma = dbeta = None
        if mode == "none":
            dx = dt
        elif mode == "batch_eval":
            # frozen statistics: the norm is a fixed per-channel affine t = a*x + b
            scale = torch.stack([ab[:, 0], torch.zeros_like(ab[:, 0])], 1).contiguous()
            dx = ops.affine_act(dt, scale, nat.ACT_NONE, 0.0)
            if has_g or has_b:
                st = ops.channel_stats(dt)[:, :, 0].sum((0, 1))                 # sum dt
                if has_b:
                    dbeta = st
                if has_g:
                    # sum dt * xhat with xhat = (x - running_mean) * running_rstd = (a*x + b - beta) / gamma
                    raise NotImplementedError("BatchNorm3d in eval mode with a trainable scale is not supported; "
                                              "call model.train() or freeze the norm parameters")
        else:
            s = ops.norm_bwd_stats(dt, x, mr)                          # (N, 2, C): sum d, sum d*xhat
            M, dgamma, dbeta = ops.norm_bwd_means(s, g32, grp, rows, want_gamma=has_g, want_beta=has_b, cpg=cpg)
            dx = ops.norm_bwd_apply_general(dt, x, mr, g32, M)
        return (dx, cast(dgamma, gamma) if has_g else None, cast(dbeta, gamma) if has_b else None, dprelu, None, None, None,
                None, None, None)


class Conv3dFn(torch.autograd.Function):
    """y = conv3d(a, W) (+ bias) (+ res): stride 1, 'same' padding, channels-last."""

    @staticmethod
    def forward(ctx, a, weight, bias, res, pad_out=False):
        """pad_out: the output is an internal feature map and is carried with ops.pad_channels(C_out) channels (all-zero tail: zero
        weight rows, zero bias); the INPUT padding follows `a` (zero weight columns).  Both directions pack their images for the
        padded channel counts; the gradients returned are the weight's own shape."""
        ks = tuple(int(k) for k in weight.shape[2:])
        co, ci = int(weight.shape[0]), int(weight.shape[1])
        ci_p = int(a.shape[-1])
        co_p = ops.pad_channels(co, a.dtype) if pad_out else co
        if ci_p < ci or (res is not None and int(res.shape[-1]) != co_p):
            raise ValueError(f"Conv3dFn: input / residual channels ({ci_p}, {None if res is None else res.shape[-1]}) do not fit a "
                             f"{ci} -> {co} conv (padded output {co_p})")
        wp = _packed(weight, "fwd", a.dtype, (co_p, ci_p))
        y = ops.conv3d(a, wp, c_out=co_p, kernel=ks, bias=_padded_vec(_f(bias), co_p), res=res)
        ctx.save_for_backward(a, weight)
        ctx.meta = (ks, bias is not None, res is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        a, weight = ctx.saved_tensors
        ks, has_bias, has_res = ctx.meta
        dy = dy.contiguous()
        if dy.dtype != a.dtype:
            dy = dy.to(a.dtype)
        co_r, ci_r = int(weight.shape[0]), int(weight.shape[1])
        ci, co = a.shape[-1], dy.shape[-1]                 # channel counts the conv ran with (>= the weight's)
        da = None
        if ctx.needs_input_grad[0]:
            # data gradient = the forward kernel on dY with the transposed, tap-mirrored weights (packed in one launch)
            da = ops.conv3d(dy, _packed(weight, "dgrad", dy.dtype, (ci, co)), c_out=ci, kernel=ks)
        db = None
        if ks == (1, 1, 1) and a.dtype == torch.bfloat16 and ci % 16 == 0 and co % 16 == 0:
            # 1x1x1 projections: the pointwise MFMA weight-gradient kernel (also returns the bias gradient)
            dW2, db = ops.pw_wgrad(a, dy, N=a.shape[0], rows_per_sample=_rows(a), c_in=ci, c_out=co, want_bias=has_bias)
            dW = dW2.view(co, ci, 1, 1, 1)
            db = db[:co_r].to(weight.dtype) if has_bias else None
        else:
            dW = ops.conv3d_wgrad(a, dy, ks)
            if has_bias:
                db = ops.channel_stats(dy)[:, :, 0].sum((0, 1))[:co_r].to(weight.dtype)
        if co != co_r or ci != ci_r:
            dW = dW[:co_r, :ci_r]
        return da, dW.to(weight.dtype).contiguous(), db, (dy if has_res else None), None


class MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, factor):
        ctx.save_for_backward(x)
        ctx.factor = tuple(int(f) for f in factor)
        return ops.maxpool3d(x, ctx.factor)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.maxpool3d_bwd(x, dy.contiguous(), ctx.factor), None


class UpsampleFn(torch.autograd.Function):
    """Fixed-weight depthwise transposed conv (BilinearUp3d); only the data gradient exists."""

    @staticmethod
    def forward(ctx, x, taps, kernel, factor, pad):
        ctx.save_for_backward(taps)
        ctx.meta = (tuple(kernel), tuple(factor), tuple(pad), tuple(x.shape[1:4]))
        return ops.dwconvT3d_generic(x, taps, kernel, factor, pad)

    @staticmethod
    def backward(ctx, dy):
        (taps,) = ctx.saved_tensors
        kernel, factor, pad, in_dims = ctx.meta
        return ops.dwconv3d_generic(dy.contiguous(), taps, kernel, factor, pad, in_dims), None, None, None, None


class ResampleConv3dFn(torch.autograd.Function):
    """Dense conv with a stride, or dense ConvTranspose3d (k, stride s, padding p, output_padding s - 1), channels-last -- the
    resampling convolutions of the MONAI-style U-Net (models/architectures/monai_models.py).  stride 1 takes the stride-1
    kernels of Conv3dFn.  Backward: data gradient of a strided conv = the transposed gather with the same weights, that of a
    transposed conv = the strided conv; weight gradient = `conv3d_wgrad_strided` with the roles of the two grids swapped
    for the transposed case (csrc/conv3d_strided_kernels.hip)."""

    @staticmethod
    def forward(ctx, a, weight, bias, stride: int, pad: int, transposed: bool):
        ks = tuple(int(k) for k in weight.shape[2:])
        s3, p3 = (int(stride),) * 3, (int(pad),) * 3
        plain = (not transposed) and stride == 1 and all(pad == k // 2 for k in ks)
        w32 = weight.detach().float().contiguous()
        if plain:
            y = ops.conv3d(a, _packed(weight, "fwd", a.dtype), c_out=weight.shape[0], kernel=ks, bias=_f(bias))
        elif (transposed and ks == (3, 3, 3) and stride == 2 and pad == 1 and a.dtype in (torch.bfloat16, torch.float32)
              and ops.convT3d_thin_supported(weight.shape[0], weight.shape[1])
              and (weight.shape[1] == 1 or not ops.convT3d_phase_supported(weight.shape[1], weight.shape[0], a.dtype))):
            # few output channels (the network's last up-sampling layer): one thread per output voxel instead of an MFMA tile
            # padded to 16 channels (csrc/conv3d_strided_kernels.hip convT3d_thin_kernel); the backward is unchanged
            y = ops.convT3d_thin(a.contiguous(), w32, _f(bias))
        elif (transposed and ks == (3, 3, 3) and stride == 2 and pad == 1
              and ops.convT3d_phase_supported(weight.shape[1], weight.shape[0], a.dtype)):
            # eight stride-1 convs (one per output parity) on the LDS-tiled kernel instead of the gather form (round 6)
            y = ops.convT3d_phase(a.contiguous(), _packed(weight, "convT_phase", a.dtype), c_out=weight.shape[1], bias=_f(bias))
        elif transposed:
            out_dims = tuple((int(d) - 1) * stride - 2 * pad + k + (stride - 1) for d, k in zip(a.shape[1:4], ks))
            y = ops.conv3d_strided(a, _packed(weight, "convT", a.dtype), c_out=weight.shape[1],
                                   kernel=ks, stride=s3, pad=p3, out_dims=out_dims, transposed=True, bias=_f(bias))
        else:
            out_dims = tuple((int(d) + 2 * pad - k) // stride + 1 for d, k in zip(a.shape[1:4], ks))
            y = ops.conv3d_strided(a, _packed(weight, "conv", a.dtype), c_out=weight.shape[0],
                                   kernel=ks, stride=s3, pad=p3, out_dims=out_dims, transposed=False, bias=_f(bias))
        ctx.save_for_backward(a, weight)
        ctx.meta = (ks, s3, p3, bool(transposed), plain, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        a, weight = ctx.saved_tensors
        ks, s3, p3, transposed, plain, has_bias = ctx.meta
        dy = dy.contiguous()
        if dy.dtype != a.dtype:
            dy = dy.to(a.dtype)
        in_dims = tuple(int(v) for v in a.shape[1:4])
        da = None
        if plain:
            if ctx.needs_input_grad[0]:
                da = ops.conv3d(dy, _packed(weight, "dgrad", dy.dtype), c_out=weight.shape[1], kernel=ks)
            dW = ops.conv3d_wgrad(a, dy, ks)
        elif transposed:
            if ctx.needs_input_grad[0]:
                da = ops.conv3d_strided(dy, _packed(weight, "convT_dgrad", dy.dtype),
                                        c_out=weight.shape[0], kernel=ks, stride=s3, pad=p3, out_dims=in_dims, transposed=False)
            dW = ops.conv3d_wgrad_strided(dy, a, ks, s3, p3)          # (C_in_T, C_out_T, k): ConvTranspose3d layout
        else:
            if ctx.needs_input_grad[0]:
                even = all(i == 2 * o for i, o in zip(in_dims, dy.shape[1:4]))
                if (ks == (3, 3, 3) and s3 == (2, 2, 2) and p3 == (1, 1, 1) and even
                        and ops.convT3d_phase_supported(weight.shape[1], weight.shape[0], dy.dtype)):
                    da = ops.convT3d_phase(dy, _packed(weight, "conv_dgrad_phase", dy.dtype), c_out=weight.shape[1], tag="conv3d_s_dgrad")
                else:
                    da = ops.conv3d_strided(dy, _packed(weight, "conv_dgrad", dy.dtype),
                                            c_out=weight.shape[1], kernel=ks, stride=s3, pad=p3, out_dims=in_dims, transposed=True)
            dW = ops.conv3d_wgrad_strided(a, dy, ks, s3, p3)          # (C_out, C_in, k)
        db = ops.channel_stats(dy)[:, :, 0].sum((0, 1)).to(weight.dtype) if has_bias else None
        return da, dW.to(weight.dtype), db, None, None, None


class AddFn(torch.autograd.Function):
    """y = a + b (the residual sum of a MONAI ResidualUnit) with the HIP add kernel; both gradients are dy."""

    @staticmethod
    def forward(ctx, a, b):
        y = a.clone()
        ops.add_(y, b.contiguous() if b.dtype == a.dtype else b.to(a.dtype).contiguous())
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


# ---- module-level composition ----------------------------------------------------------------------------------------
def _norm_act(na, x):
    m = na.norm
    kind = na.kind
    gamma = getattr(m, "weight", None) if kind != "none" else None
    beta = getattr(m, "bias", None) if kind != "none" else None
    groups = m.num_groups if kind == "group" else 1
    eps = float(getattr(m, "eps", 1e-5))
    if na.act_kind == "leakyrelu":
        prm = float(na.act.negative_slope)
    elif na.act_kind == "elu":
        prm = float(na.act.alpha)
    else:
        prm = 0.0
    prelu_w = na.act.weight if na.act_kind == "prelu" else None
    return NormActFn.apply(x, gamma, beta, prelu_w, kind, groups, eps, na.act_kind, prm, m if kind == "batch" else None)


def _nac(na, conv, x, res=None, pad_out=True):
    return Conv3dFn.apply(_norm_act(na, x), conv.weight, conv.bias, res, pad_out)


def _conv_block(blk, x):
    """Feature maps between the convs of RSUNet travel with ops.pad_channels(width) channels (16-byte rows for every kernel: the
    reference's stock widths 18 / 36 are not multiples of 8); only the heads produce their own channel count."""
    x = _nac(blk.pre[0], blk.pre[1], x)
    r = blk.res
    a1 = _norm_act(r.norm_act1, x)
    # reference quirk: with norm='none' the in-place activation also rewrites the residual source (rsunet.py:103-113)
    res = a1 if (r.norm_act1.kind == "none" and r.norm_act1.act_kind != "prelu") else x
    h = Conv3dFn.apply(a1, r.conv1.weight, r.conv1.bias, None, True)
    x = _nac(r.norm_act2, r.conv2, h, res=res)
    return _nac(blk.post[0], blk.post[1], x)


def rsunet_train_forward(model, x_cl: torch.Tensor, compute_dtype: torch.dtype):
    """Differentiable RSUNet forward on channels-last input; returns {"output", "ds_i"...} of fp32 channels-last maps."""
    refresh_conv_packs()
    x = x_cl if x_cl.dtype == compute_dtype else x_cl.to(compute_dtype)
    x = _conv_block(model.input_conv, x.contiguous())
    skips = []
    for down in model.down_blocks:
        skips.append(x)
        x = _conv_block(down.conv, MaxPoolFn.apply(x, down.pool.kernel_size))
    ds_feats = []
    for i, up in enumerate(model.up_blocks):
        if model.supports_deep_supervision and (model.depth - i - 1) < len(model.ds_heads):
            ds_feats.append(x)
        u = up.up
        taps = u.weight.detach().float().reshape(u.groups, -1).t().contiguous()
        if taps.shape[1] != x.shape[-1]:                      # fixed bilinear stencil per channel: any value serves the zero tail
            taps = torch.nn.functional.pad(taps, (0, x.shape[-1] - taps.shape[1]))
        x = UpsampleFn.apply(x, taps, tuple(u.kernel_size), tuple(u.factor), tuple(u.padding))
        x = Conv3dFn.apply(x, up.proj.weight, up.proj.bias, skips.pop(), True)
        x = _conv_block(up.conv, x)
    out = {"output": _nac(model.final_norm, model.output_head, x, pad_out=False).float()}
    if model.supports_deep_supervision:
        for i, (ft, head) in enumerate(zip(ds_feats, model.ds_heads)):
            out[f"ds_{i + 1}"] = Conv3dFn.apply(ft, head.weight, head.bias, None, False).float()
    return out


__all__ = ["NormActFn", "Conv3dFn", "ResampleConv3dFn", "AddFn", "MaxPoolFn", "UpsampleFn", "rsunet_train_forward"]
