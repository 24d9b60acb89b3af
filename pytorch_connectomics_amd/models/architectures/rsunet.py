"""Residual Symmetric U-Net on the MI355X -- counterpart of the reference's
connectomics/models/architectures/rsunet.py (BilinearUp3d :33-70, NormAct :73-118, ResBlock :121-154,
ConvBlock :157-198, DownBlock :201-222, UpBlock :225-259, RSUNet :262-461, builders :469-541).

The module tree and parameter / buffer names equal the reference's (state dicts interchange); the
torch.nn children are parameter holders.  `forward` runs hand-written gfx950 kernels on NDHWC tensors:

  NormAct -> Conv3d   one implicit-GEMM MFMA launch, norm-apply + activation fused as its prologue
                      (statistics: channel_stats + norm_finalize_groups; BatchNorm uses running stats)
  ResBlock            the residual add is the second conv's epilogue
  DownBlock           maxpool3d kernel, then ConvBlock
  UpBlock             fixed-weight depthwise transposed conv ("bilinear"), 1x1 projection with the skip
                      add as epilogue, then ConvBlock
  heads               1x1 conv with bias (final NormAct fused as prologue)

The fused kernels above are the inference path; with autograd enabled `forward` runs the un-fused training schedule of
training/rsunet_autograd.py (every forward and backward op a HIP kernel; BatchNorm uses batch statistics and updates
its running buffers like nn.BatchNorm3d).  No CPU path.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple, Union

import torch
import torch.nn as nn

from ... import _native as nat
from ... import hip_ops as ops
from .base import ConnectomicsModel
from .mednext import _WeightCache, resolve_compute_dtype, to_channels_first, to_channels_last
from .registry import register_architecture


def _triple(k) -> Tuple[int, int, int]:
    return (k, k, k) if isinstance(k, int) else tuple(int(v) for v in k)


class BilinearUp3d(nn.Module):
    """Depthwise transposed conv with fixed separable 'bilinear' weights in (y, x) (Caffe style)."""

    def __init__(self, in_channels: int, out_channels: int, factor: Tuple[int, int, int] = (1, 2, 2)):
        super().__init__()
        if in_channels != out_channels:
            raise ValueError("BilinearUp3d requires in_channels == out_channels")
        self.groups = in_channels
        self.factor = tuple(int(f) for f in factor)
        self.kernel_size = [2 * f - f % 2 for f in self.factor]
        self.padding = [int(math.ceil((f - 1) / 2.0)) for f in self.factor]
        self.init_weights()

    def init_weights(self):
        kz, ky, kx = self.kernel_size
        if ky != kx:
            raise ValueError("Bilinear weight assumes square kernel in HW")
        f = float(math.ceil(kx / 2.0))
        c = float(kx - 1) / (2.0 * f)
        ramp = torch.tensor([1 - abs(i / f - c) for i in range(kx)])
        plane = ramp[:, None] * ramp[None, :]           # depends on (h, w) only; replicated along z
        self.register_buffer("weight", plane.expand(self.groups, 1, kz, ky, kx).clone())


class NormAct(nn.Module):
    def __init__(self, channels: int, norm: str = "batch", activation: str = "relu", num_groups: int = 8,
                 **act_kwargs):
        super().__init__()
        self.kind = norm
        if norm == "batch":
            self.norm = nn.BatchNorm3d(channels)
        elif norm == "group":
            g = min(num_groups, channels)
            while channels % g:
                g -= 1
            self.norm = nn.GroupNorm(g, channels)
        elif norm == "instance":
            self.norm = nn.InstanceNorm3d(channels)
        elif norm == "none":
            self.norm = nn.Identity()
        else:
            raise ValueError(f"Unknown normalization: {norm}")
        self.act_kind = activation
        if activation == "relu":
            self.act = nn.ReLU(inplace=True)
        elif activation == "leakyrelu":
            self.act = nn.LeakyReLU(act_kwargs.get("negative_slope", 0.01), inplace=True)
        elif activation == "prelu":
            self.act = nn.PReLU(init=act_kwargs.get("init", 0.25))
        elif activation == "elu":
            self.act = nn.ELU(act_kwargs.get("alpha", 1.0), inplace=True)
        else:
            raise ValueError(f"Unknown activation: {activation}")


class ResBlock(nn.Module):
    def __init__(self, channels: int, kernel_size=3, norm="batch", activation="relu", num_groups=8, **act_kwargs):
        super().__init__()
        ks = _triple(kernel_size)
        pad = tuple(k // 2 for k in ks)
        self.norm_act1 = NormAct(channels, norm, activation, num_groups, **act_kwargs)
        self.conv1 = nn.Conv3d(channels, channels, ks, padding=pad, bias=False)
        self.norm_act2 = NormAct(channels, norm, activation, num_groups, **act_kwargs)
        self.conv2 = nn.Conv3d(channels, channels, ks, padding=pad, bias=False)


class ConvBlock(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, norm="batch", activation="relu", num_groups=8,
                 **act_kwargs):
        super().__init__()
        ks = _triple(kernel_size)
        pad = tuple(k // 2 for k in ks)
        self.pre = nn.Sequential(NormAct(in_channels, norm, activation, num_groups, **act_kwargs),
                                 nn.Conv3d(in_channels, out_channels, ks, padding=pad, bias=False))
        self.res = ResBlock(out_channels, ks, norm, activation, num_groups, **act_kwargs)
        self.post = nn.Sequential(NormAct(out_channels, norm, activation, num_groups, **act_kwargs),
                                  nn.Conv3d(out_channels, out_channels, ks, padding=pad, bias=False))


class DownBlock(nn.Module):
    def __init__(self, in_channels, out_channels, down_factor=(1, 2, 2), kernel_size=3, norm="batch",
                 activation="relu", num_groups=8, **act_kwargs):
        super().__init__()
        self.pool = nn.MaxPool3d(tuple(down_factor))
        self.conv = ConvBlock(in_channels, out_channels, kernel_size, norm, activation, num_groups, **act_kwargs)


class UpBlock(nn.Module):
    def __init__(self, in_channels, out_channels, up_factor=(1, 2, 2), kernel_size=3, norm="batch",
                 activation="relu", num_groups=8, **act_kwargs):
        super().__init__()
        self.up = BilinearUp3d(in_channels, in_channels, factor=tuple(up_factor))
        self.proj = nn.Conv3d(in_channels, out_channels, kernel_size=1, bias=False)
        self.conv = ConvBlock(out_channels, out_channels, kernel_size, norm, activation, num_groups, **act_kwargs)


class _RSUNetHip:
    """Kernel-level forward of the RSUNet building blocks on channels-last tensors."""

    def __init__(self):
        self.cache = _WeightCache()

    def _conv_w(self, conv: nn.Conv3d, dt, pad_to=None):
        """pad_to = (C_out, C_in) the conv runs with when its feature maps carry alignment padding (ops.pad_channels)."""
        w = conv.weight
        if pad_to is not None and tuple(pad_to) == (conv.out_channels, conv.in_channels):
            pad_to = None
        if pad_to is None:
            return self.cache.get(("c3", id(conv), dt), [w],
                                  lambda: ops.conv3d_pack_weight(w.detach().float().contiguous(), dt))
        return self.cache.get(("c3", id(conv), dt, tuple(pad_to)), [w],
                              lambda: ops.conv3d_pack_weight_padded(w.detach().float().contiguous(), "fwd", dt, tuple(pad_to)))

    def _vec(self, owner, name, p, n: Optional[int] = None):
        """fp32 copy of a per-channel parameter, zero-extended to n entries next to channel-padded activations."""
        if p is None:
            return None
        if n is None or n == p.numel():
            return self.cache.get(("v", id(owner), name), [p], lambda: p.detach().float().reshape(-1).contiguous())
        return self.cache.get(("v", id(owner), name, n), [p],
                              lambda: torch.nn.functional.pad(p.detach().float().reshape(-1), (0, n - p.numel())).contiguous())

    def _act(self, na: NormAct):
        k = na.act_kind
        if k == "relu":
            return nat.ACT_RELU, 0.0
        if k == "leakyrelu":
            return nat.ACT_LEAKY, float(na.act.negative_slope)
        if k == "prelu":
            w = na.act.weight
            return nat.ACT_LEAKY, self.cache.get(("prelu", id(na)), [w], lambda: float(w.detach().reshape(-1)[0].item()))
        return nat.ACT_ELU, float(na.act.alpha)

    def norm_affine(self, na: NormAct, x: torch.Tensor) -> Optional[torch.Tensor]:
        """(N, 2, C) per-sample affine of the norm layer for input x, or None for norm='none'."""
        N, C = x.shape[0], x.shape[-1]
        rows = x.numel() // (N * C)
        m = na.norm
        if na.kind == "none":
            return None
        if na.kind == "batch":
            if m.training:
                raise NotImplementedError("BatchNorm3d in training mode is not supported by the HIP engine "
                                          "(call model.eval())")
            def make():
                a = m.weight.detach().float() / torch.sqrt(m.running_var.float() + m.eps)
                b = m.bias.detach().float() - m.running_mean.float() * a
                pad = (0, C - a.numel())          # channel-padded features: affine (0, 0) on the zero tail
                return torch.stack([torch.nn.functional.pad(a, pad), torch.nn.functional.pad(b, pad)],
                                   0).unsqueeze(0).expand(N, 2, C).contiguous()
            return self.cache.get(("bn", id(m), N, C), [m.weight, m.bias, m.running_mean, m.running_var], make)
        st = ops.channel_stats(x)
        if na.kind == "group":
            c_real = int(m.num_channels)
            return ops.norm_finalize_groups(st, rows, self._vec(m, "w", m.weight), self._vec(m, "b", m.bias), m.eps,
                                            m.num_groups, c_real // m.num_groups if c_real != C else 0)
        return ops.norm_finalize_groups(st, rows, None, None, m.eps, C)     # instance norm, no affine (a zero tail stays zero)

    def norm_act_conv(self, na: NormAct, conv: nn.Conv3d, x: torch.Tensor, res=None, pad_out: bool = True) -> torch.Tensor:
        """pad_out: the result is an internal feature map and travels with ops.pad_channels(C_out) channels (16-byte rows for every
        kernel; the reference's stock widths 18 / 36 become 24 / 40, the extra channels exactly zero); heads pass False."""
        act, prm = self._act(na)
        co = ops.pad_channels(conv.out_channels, x.dtype) if pad_out else conv.out_channels
        return ops.conv3d(x, self._conv_w(conv, x.dtype, (co, int(x.shape[-1]))), c_out=co, kernel=conv.kernel_size,
                          bias=self._vec(conv, "bias", conv.bias, co), ab=self.norm_affine(na, x), act_in=act,
                          act_param=prm, res=res)

    def conv_block(self, blk: ConvBlock, x: torch.Tensor) -> torch.Tensor:
        x = self.norm_act_conv(blk.pre[0], blk.pre[1], x)
        r = blk.res
        res = x
        if r.norm_act1.kind == "none" and r.norm_act1.act_kind != "prelu":
            # reference quirk: with norm='none' the in-place activation overwrites the residual source
            # (rsunet.py:103-113,150-154) -- reproduced for checkpoint-compatible numerics
            act, prm = self._act(r.norm_act1)
            res = ops.affine_act(x, None, act, prm)
        h = self.norm_act_conv(r.norm_act1, r.conv1, x)
        x = self.norm_act_conv(r.norm_act2, r.conv2, h, res=res)
        return self.norm_act_conv(blk.post[0], blk.post[1], x)

    def up_block(self, blk: UpBlock, x: torch.Tensor, skip: torch.Tensor) -> torch.Tensor:
        up = blk.up
        cp = int(x.shape[-1])
        taps = self.cache.get(("up", id(up), cp), [up.weight],
                              lambda: torch.nn.functional.pad(up.weight.detach().float().reshape(up.groups, -1).t(),
                                                              (0, cp - up.groups)).contiguous())
        x = ops.dwconvT3d_generic(x, taps, up.kernel_size, up.factor, up.padding)
        co = ops.pad_channels(blk.proj.out_channels, x.dtype)
        x = ops.conv3d(x, self._conv_w(blk.proj, x.dtype, (co, cp)), c_out=co, kernel=(1, 1, 1),
                       bias=self._vec(blk.proj, "bias", blk.proj.bias, co), res=skip)
        return self.conv_block(blk.conv, x)


class RSUNet(ConnectomicsModel):
    """Residual Symmetric U-Net (pre-activation residual blocks, additive skips, bilinear transposed-conv
    upsampling, anisotropic (1,2,2) pooling by default, optional 2-D kernels in shallow levels, optional deep
    supervision: {"output", "ds_1" (deepest) ...})."""

    def __init__(self, in_channels: int, out_channels: int, width: Optional[List[int]] = None,
                 kernel_sizes: Union[int, List] = 3, down_factors: Optional[List[Tuple[int, int, int]]] = None,
                 norm: str = "batch", activation: str = "relu", num_groups: int = 8, deep_supervision: bool = False,
                 depth_2d: int = 0, kernel_2d: Tuple[int, int, int] = (1, 3, 3), **act_kwargs):
        super().__init__()
        if width is None:
            width = [16, 32, 64, 128, 256]
        if len(width) <= 1:
            raise ValueError("Need at least 2 levels")
        self.depth = len(width) - 1
        self.width = list(width)
        self.supports_deep_supervision = deep_supervision
        self.output_scales = 5 if deep_supervision else 1
        if down_factors is None:
            down_factors = [(1, 2, 2)] * self.depth
        if len(down_factors) != self.depth:
            raise ValueError(f"down_factors length ({len(down_factors)}) must match depth ({self.depth})")
        if isinstance(kernel_sizes, int):
            kernel_sizes = [kernel_sizes] * len(width)
        else:
            kernel_sizes = list(kernel_sizes) + [kernel_sizes[-1]] * (len(width) - len(kernel_sizes))
        for i in range(min(depth_2d, len(kernel_sizes))):
            kernel_sizes[i] = tuple(kernel_2d)
        kw = dict(norm=norm, activation=activation, num_groups=num_groups, **act_kwargs)
        self.input_conv = ConvBlock(in_channels, width[0], kernel_sizes[0], **kw)
        self.down_blocks = nn.ModuleList(DownBlock(width[d], width[d + 1], down_factors[d], kernel_sizes[d + 1], **kw)
                                         for d in range(self.depth))
        self.up_blocks = nn.ModuleList(UpBlock(width[d + 1], width[d], down_factors[d], kernel_sizes[d], **kw)
                                       for d in reversed(range(self.depth)))
        self.final_norm = NormAct(width[0], norm, activation, num_groups, **act_kwargs)
        self.output_head = nn.Conv3d(width[0], out_channels, kernel_size=1)
        if deep_supervision:
            self.ds_heads = nn.ModuleList(nn.Conv3d(width[self.depth - d], out_channels, kernel_size=1)
                                          for d in range(min(4, self.depth)))
        self.init_weights()
        self.compute_dtype: Optional[torch.dtype] = None
        self._hip = _RSUNetHip()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, (nn.BatchNorm3d, nn.GroupNorm, nn.InstanceNorm3d)):
                if m.weight is not None:
                    nn.init.constant_(m.weight, 1)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    # ---- HIP forward -------------------------------------------------------------------------------
    def _forward_cl(self, x_cl: torch.Tensor) -> Dict[str, torch.Tensor]:
        hip = self._hip
        dt = resolve_compute_dtype(self.compute_dtype)
        x = x_cl if x_cl.dtype == dt else x_cl.to(dt)
        x = hip.conv_block(self.input_conv, x.contiguous())
        skips = []
        for down in self.down_blocks:
            skips.append(x)
            x = hip.conv_block(down.conv, ops.maxpool3d(x, down.pool.kernel_size))
        ds_feats = []
        for i, up in enumerate(self.up_blocks):
            if self.supports_deep_supervision and (self.depth - i - 1) < len(self.ds_heads):
                ds_feats.append(x)
            x = hip.up_block(up, x, skips.pop())
        out = {"output": hip.norm_act_conv(self.final_norm, self.output_head, x, pad_out=False).float()}
        if self.supports_deep_supervision:
            for i, (ft, head) in enumerate(zip(ds_feats, self.ds_heads)):
                out[f"ds_{i + 1}"] = ops.conv3d(ft, hip._conv_w(head, ft.dtype, (head.out_channels, int(ft.shape[-1]))),
                                                c_out=head.out_channels, kernel=(1, 1, 1),
                                                bias=hip._vec(head, "bias", head.bias)).float()
        return out

    def forward_cl(self, x_cl: torch.Tensor) -> torch.Tensor:
        """Channels-last fast path of the sliding-window engine: (N,D,H,W,C_in) -> (N,D,H,W,C_out) fp32."""
        return self._forward_cl(x_cl)["output"]

    def forward(self, x: torch.Tensor):
        if not x.is_cuda:
            raise RuntimeError("RSUNet (pytorch_connectomics_amd) runs only on an MI355X/ROCm device: "
                               "there is no CPU path. Move the model and input to 'cuda'.")
        if x.dim() != 5:
            raise ValueError(f"RSUNet expects (B, C, D, H, W), got {tuple(x.shape)}")
        bn_batch_stats = self.training and any(isinstance(mod, nn.BatchNorm3d) for mod in self.modules())
        if (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())) or bn_batch_stats:
            # training: autograd Functions whose forward and backward are HIP kernels (training/rsunet_autograd.py); also a
            # train()-mode forward under no_grad with BatchNorm, which uses batch statistics and updates the running buffers
            from ...training.rsunet_autograd import rsunet_train_forward
            res = rsunet_train_forward(self, to_channels_last(x.float()), resolve_compute_dtype(self.compute_dtype))
            out = {k: v.permute(0, 4, 1, 2, 3) for k, v in res.items()}
            return out if self.supports_deep_supervision else out["output"]
        out = {k: to_channels_first(v) for k, v in self._forward_cl(to_channels_last(x.float())).items()}
        return out if self.supports_deep_supervision else out["output"]


@register_architecture("rsunet")
def build_rsunet(cfg) -> RSUNet:
    """RSUNet from cfg.model.rsunet.{width, norm, activation, num_groups, down_factors, depth_2d, kernel_2d,
    act_negative_slope, act_init}; model.loss.deep_supervision enables the multi-scale heads."""
    r = cfg.model.rsunet
    down = getattr(r, "down_factors", None)
    k2d = getattr(r, "kernel_2d", None)
    return RSUNet(in_channels=cfg.model.in_channels, out_channels=cfg.model.out_channels,
                  width=list(getattr(r, "width", [16, 32, 64, 128])), norm=getattr(r, "norm", "batch"),
                  activation=getattr(r, "activation", "relu"), num_groups=getattr(r, "num_groups", 8),
                  deep_supervision=getattr(cfg.model.loss, "deep_supervision", False),
                  down_factors=None if down is None else [tuple(f) for f in down],
                  depth_2d=getattr(r, "depth_2d", 0), kernel_2d=(1, 3, 3) if k2d is None else tuple(k2d),
                  negative_slope=getattr(r, "act_negative_slope", 0.01), init=getattr(r, "act_init", 0.25))


@register_architecture("rsunet_iso")
def build_rsunet_iso(cfg) -> RSUNet:
    """RSUNet with isotropic (2,2,2) down-sampling at every level."""
    r = cfg.model.rsunet
    width = list(getattr(r, "width", [16, 32, 64, 128]))
    return RSUNet(in_channels=cfg.model.in_channels, out_channels=cfg.model.out_channels, width=width,
                  down_factors=[(2, 2, 2)] * (len(width) - 1), norm=getattr(r, "norm", "batch"),
                  activation=getattr(r, "activation", "relu"), num_groups=getattr(r, "num_groups", 8),
                  deep_supervision=getattr(cfg.model.loss, "deep_supervision", False))


__all__ = ["RSUNet", "BilinearUp3d", "build_rsunet", "build_rsunet_iso"]
