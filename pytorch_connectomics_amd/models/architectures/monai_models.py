"""MONAI-style residual U-Net on the MI355X -- counterpart of the `monai_unet` architecture of the reference
(connectomics/models/architectures/monai_models.py: MONAIModelWrapper :29-55, `_infer_spatial_dims` :66-70, `_resolve_norm`
:73-79, UpsampleModeUNet :84-139, build_monai_unet :197-250), which delegates the network itself to the third-party,
un-vendored `monai` package (`monai.networks.nets.UNet`, `monai.networks.blocks.{Convolution, ResidualUnit, ADN}`).

The module tree below restates that published architecture with the SAME child names, so state-dict keys interchange
(`model.model.0.conv.unit0.conv.weight`, `...adn.N.running_mean`, `...adn.A.weight`, `model.model.1.submodule...`,
`model.model.2.0.conv.weight` ...):

    UNet.model   = Sequential(down, SkipConnection(sub-block), up)            channel concat skip
    down         = ResidualUnit(in, c, stride 2, subunits = num_res_units)    conv k3 s2 -> ADN -> conv k3 -> ADN, + residual
                   residual = Conv3d(k3, s2, p1) when strided, Conv3d(k1) when only the width changes, Identity otherwise
    bottom       = ResidualUnit(c, c_last, stride 1)
    up           = Sequential(Convolution(transposed k3 s2 p1 op1 -> ADN), ResidualUnit(out, out, subunits 1,
                   last_conv_only = is_top))
    ADN ("NDA")  = norm (batch | instance | group) -> Dropout(p) -> PReLU (one weight, 0.25)

The torch.nn children are parameter holders.  `forward` runs hand-written gfx950 kernels on NDHWC tensors through the
autograd Functions of training/rsunet_autograd.py (dense conv3d on MFMA; the stride-2 / transposed resampling convs
in csrc/conv3d_strided_kernels.hip; norm statistics, affine + PReLU, their backward kernels), in training and inference.
No CPU path.
"""
from __future__ import annotations

from typing import Optional, Sequence, Union

import torch
import torch.nn as nn

from .base import ConnectomicsModel
from .mednext import resolve_compute_dtype, to_channels_first, to_channels_last
from .registry import register_architecture

MONAI_AVAILABLE = True      # the architecture is built in-repo; nothing to import


def _same_padding(kernel_size: int) -> int:
    if (kernel_size - 1) % 2 == 1:
        raise NotImplementedError(f"Same padding not available for kernel_size={kernel_size}.")
    return (kernel_size - 1) // 2


def _make_norm(norm, channels: int) -> nn.Module:
    """MONAI norm factory for the three spellings `_resolve_norm` can hand over (monai_models.py:73-79)."""
    args = {}
    if isinstance(norm, (tuple, list)):
        norm, args = norm[0], dict(norm[1])
    name = str(norm).lower()
    if name == "batch":
        return nn.BatchNorm3d(channels, **args)
    if name == "instance":
        return nn.InstanceNorm3d(channels, **args)
    if name == "group":
        return nn.GroupNorm(num_channels=channels, **args)
    raise ValueError(f"Unsupported MONAI norm {norm!r} for the MI355X engine (batch, instance, group)")


class ADN(nn.Sequential):
    """norm -> dropout -> activation ("NDA"), children named N / D / A as in monai.networks.blocks.ADN."""

    def __init__(self, channels: int, norm, dropout: Optional[float]):
        super().__init__()
        self.add_module("N", _make_norm(norm, channels))
        if dropout is not None:
            self.add_module("D", nn.Dropout(float(dropout)))
        self.add_module("A", nn.PReLU())


class Convolution(nn.Sequential):
    """conv (or transposed conv) [+ ADN] -- monai.networks.blocks.Convolution for spatial_dims = 3."""

    def __init__(self, in_channels: int, out_channels: int, strides: int = 1, kernel_size: int = 3, norm="instance",
                 dropout: Optional[float] = None, bias: bool = True, conv_only: bool = False, is_transposed: bool = False):
        super().__init__()
        pad = _same_padding(kernel_size)
        if is_transposed:
            conv = nn.ConvTranspose3d(in_channels, out_channels, kernel_size, stride=strides, padding=pad,
                                      output_padding=strides - 1, bias=bias)
        else:
            conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride=strides, padding=pad, bias=bias)
        self.add_module("conv", conv)
        if not conv_only:
            self.add_module("adn", ADN(out_channels, norm, dropout))


class ResidualUnit(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, strides: int = 1, kernel_size: int = 3, subunits: int = 2,
                 norm="instance", dropout: Optional[float] = None, bias: bool = True, last_conv_only: bool = False):
        super().__init__()
        self.conv = nn.Sequential()
        self.residual: nn.Module = nn.Identity()
        pad = _same_padding(kernel_size)
        sch, sst = in_channels, strides
        subunits = max(1, subunits)
        for su in range(subunits):
            conv_only = last_conv_only and su == subunits - 1
            self.conv.add_module(f"unit{su:d}", Convolution(sch, out_channels, strides=sst, kernel_size=kernel_size,
                                                            norm=norm, dropout=dropout, bias=bias, conv_only=conv_only))
            sch, sst = out_channels, 1
        if strides != 1 or in_channels != out_channels:
            rk, rp = kernel_size, pad
            if strides == 1:            # only the width changes: 1x1x1, no padding
                rk, rp = 1, 0
            self.residual = nn.Conv3d(in_channels, out_channels, rk, strides, rp, bias=bias)


class SkipConnection(nn.Module):
    """cat([x, submodule(x)], channel dim) -- monai.networks.layers.SkipConnection(mode='cat')."""

    def __init__(self, submodule: nn.Module):
        super().__init__()
        self.submodule = submodule


class UNet(nn.Module):
    """monai.networks.nets.UNet (spatial_dims = 3, act = PReLU, adn_ordering = 'NDA', up_kernel_size = 3)."""

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int, channels: Sequence[int],
                 strides: Sequence[int], kernel_size: int = 3, up_kernel_size: int = 3, num_res_units: int = 0,
                 norm="instance", dropout: float = 0.0, bias: bool = True):
        super().__init__()
        if spatial_dims != 3:
            raise NotImplementedError("the MI355X engine builds the 3-D MONAI U-Net only (spatial_dims=3)")
        if len(channels) < 2:
            raise ValueError("the length of `channels` should be no less than 2.")
        delta = len(strides) - (len(channels) - 1)
        if delta < 0:
            raise ValueError("the length of `strides` should equal to `len(channels) - 1`.")
        self.dimensions = spatial_dims
        self.in_channels, self.out_channels = in_channels, out_channels
        self.channels, self.strides = tuple(channels), tuple(strides)
        self.kernel_size, self.up_kernel_size = kernel_size, up_kernel_size
        self.num_res_units, self.norm, self.dropout, self.bias = num_res_units, norm, dropout, bias

        def create(inc: int, outc: int, chans: Sequence[int], strs: Sequence[int], is_top: bool) -> nn.Module:
            c, s = chans[0], strs[0]
            if len(chans) > 2:
                sub = create(c, c, chans[1:], strs[1:], False)
                upc = c * 2
            else:
                sub = self._down(c, chans[1], 1)                       # bottom layer
                upc = c + chans[1]
            return nn.Sequential(self._down(inc, c, s), SkipConnection(sub), self._up(upc, outc, s, is_top))

        self.model = create(in_channels, out_channels, self.channels, self.strides, True)

    def _down(self, inc: int, outc: int, strides: int) -> nn.Module:
        """monai UNet._get_down_layer: a ResidualUnit, or with num_res_units = 0 one plain Convolution (conv -> norm -> PReLU)."""
        if self.num_res_units > 0:
            return ResidualUnit(inc, outc, strides=strides, kernel_size=self.kernel_size, subunits=self.num_res_units,
                                norm=self.norm, dropout=self.dropout, bias=self.bias)
        return Convolution(inc, outc, strides=strides, kernel_size=self.kernel_size, norm=self.norm, dropout=self.dropout,
                           bias=self.bias)

    def _up(self, inc: int, outc: int, strides: int, is_top: bool) -> nn.Module:
        """monai UNet._get_up_layer: transposed Convolution (+ a one-subunit ResidualUnit when num_res_units > 0); the top layer ends
        without norm / activation -- on the ResidualUnit's last conv, or with num_res_units = 0 on the transposed conv itself."""
        conv = Convolution(inc, outc, strides=strides, kernel_size=self.up_kernel_size, norm=self.norm,
                           dropout=self.dropout, bias=self.bias, conv_only=is_top and self.num_res_units == 0, is_transposed=True)
        if self.num_res_units <= 0:
            return conv
        ru = ResidualUnit(outc, outc, strides=1, kernel_size=self.kernel_size, subunits=1, norm=self.norm,
                          dropout=self.dropout, bias=self.bias, last_conv_only=is_top)
        return nn.Sequential(conv, ru)

    def forward(self, x):  # pragma: no cover - guard only
        raise RuntimeError("the MONAI-style UNet executes through MONAIModelWrapper.forward (HIP engine); its modules are "
                           "parameter holders")


class UpsampleModeUNet(UNet):
    """Name kept from the reference (monai_models.py:84-139); only the default transposed-conv upsampling is built."""

    def __init__(self, upsample_mode: str = "deconv", upsample_interp_mode: str = "linear",
                 upsample_align_corners: bool = True, **kwargs):
        if upsample_mode and upsample_mode != "deconv":
            raise NotImplementedError(f"monai_unet upsample_mode={upsample_mode!r}: only 'deconv' has HIP kernels")
        self.upsample_mode = upsample_mode
        self.upsample_interp_mode = upsample_interp_mode
        self.upsample_align_corners = upsample_align_corners
        super().__init__(**kwargs)


# ---------------------------------------------------------------------------------------------------- HIP execution
def _adn(adn: ADN, x: torch.Tensor) -> torch.Tensor:
    from ...training.rsunet_autograd import NormActFn
    n = adn.N
    drop = getattr(adn, "D", None)
    if drop is not None and drop.p > 0 and drop.training:
        raise NotImplementedError("dropout > 0 in training mode has no HIP kernel (monai.dropout must be 0.0)")
    if isinstance(n, nn.BatchNorm3d):
        kind, groups, bn = "batch", 1, n
    elif isinstance(n, nn.GroupNorm):
        kind, groups, bn = "group", n.num_groups, None
    else:
        kind, groups, bn = "instance", 1, None
    return NormActFn.apply(x, getattr(n, "weight", None), getattr(n, "bias", None), adn.A.weight, kind, groups,
                           float(n.eps), "prelu", 0.0, bn)


def _convolution(m: Convolution, x: torch.Tensor) -> torch.Tensor:
    from ...training.rsunet_autograd import ResampleConv3dFn
    y = ResampleConv3dFn.apply(x, m.conv.weight, m.conv.bias, int(m.conv.stride[0]), int(m.conv.padding[0]),
                               isinstance(m.conv, nn.ConvTranspose3d))
    return _adn(m.adn, y) if hasattr(m, "adn") else y


def _residual_unit(m: ResidualUnit, x: torch.Tensor) -> torch.Tensor:
    from ...training.rsunet_autograd import AddFn, ResampleConv3dFn
    if isinstance(m.residual, nn.Identity):
        res = x
    else:
        r = m.residual
        res = ResampleConv3dFn.apply(x, r.weight, r.bias, int(r.stride[0]), int(r.padding[0]), False)
    cx = x
    for unit in m.conv:
        cx = _convolution(unit, cx)
    return AddFn.apply(cx, res)


def _run(mod: nn.Module, x: torch.Tensor) -> torch.Tensor:
    if isinstance(mod, ResidualUnit):
        return _residual_unit(mod, x)
    if isinstance(mod, Convolution):
        return _convolution(mod, x)
    if isinstance(mod, SkipConnection):
        y = _run(mod.submodule, x)
        if tuple(y.shape[1:4]) != tuple(x.shape[1:4]):
            # what torch.cat raises inside MONAI's SkipConnection, spelled out
            raise ValueError(f"monai_unet skip connection: spatial size {tuple(x.shape[1:4])} is not restored by the "
                             f"down/up path (got {tuple(y.shape[1:4])}); every spatial size must be divisible by "
                             "the product of the strides")
        return torch.cat([x, y], dim=-1)
    if isinstance(mod, nn.Sequential):
        for child in mod:
            x = _run(child, x)
        return x
    raise TypeError(f"unexpected module in the MONAI U-Net tree: {type(mod).__name__}")


class MONAIModelWrapper(ConnectomicsModel):
    """ConnectomicsModel interface over the MONAI-style network; single-scale output (no deep supervision)."""

    def __init__(self, model: nn.Module):
        super().__init__()
        self.model = model
        self.supports_deep_supervision = False
        self.output_scales = 1
        self.compute_dtype: Optional[torch.dtype] = None       # None -> follow autocast

    def forward_cl(self, x_cl: torch.Tensor) -> torch.Tensor:
        """Channels-last entry of the sliding-window engine: (N,D,H,W,C_in) -> (N,D,H,W,C_out) fp32."""
        if not x_cl.is_cuda:
            raise RuntimeError("monai_unet (pytorch_connectomics_amd) runs only on an MI355X/ROCm device: "
                               "there is no CPU path. Move the model and input to 'cuda'.")
        from ...training.rsunet_autograd import prefetch_prelu, refresh_conv_packs
        prefetch_prelu([m.weight for m in self.model.modules() if isinstance(m, nn.PReLU)])   # one host read, not one per layer
        refresh_conv_packs()          # every conv-weight image whose weight changed since the last forward, in one launch
        dt = resolve_compute_dtype(self.compute_dtype)
        x = x_cl if x_cl.dtype == dt else x_cl.to(dt)
        return _run(self.model.model, x.contiguous()).float()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("monai_unet (pytorch_connectomics_amd) runs only on an MI355X/ROCm device: "
                               "there is no CPU path. Move the model and input to 'cuda'.")
        if x.dim() != 5:
            raise NotImplementedError("the MI355X engine runs the 3-D MONAI U-Net on (B, C, D, H, W) inputs")
        y = self.forward_cl(to_channels_last(x.float()))
        return y.permute(0, 4, 1, 2, 3) if y.requires_grad else to_channels_first(y)


def _infer_spatial_dims(cfg) -> int:
    if hasattr(cfg.model, "input_size") and cfg.model.input_size:
        return len(cfg.model.input_size)
    return getattr(cfg.model.monai, "spatial_dims", 3)


def _resolve_norm(cfg):
    norm_type = getattr(cfg.model.monai, "norm", "batch")
    if norm_type == "group":
        return ("group", {"num_groups": getattr(cfg.model.monai, "num_groups", 8)})
    return norm_type


@register_architecture("monai_unet")
def build_monai_unet(cfg) -> ConnectomicsModel:
    """MONAI UNet with residual units: model.monai.{filters, num_res_units, kernel_size, norm, num_groups, dropout,
    upsample_mode}; 2x down-sampling at every level (strides = [2] * (len(filters) - 1), monai_models.py:228-229)."""
    mc = cfg.model.monai
    channels = list(getattr(mc, "filters", [32, 64, 128, 256, 512]))
    model = UpsampleModeUNet(
        spatial_dims=_infer_spatial_dims(cfg), in_channels=cfg.model.in_channels, out_channels=cfg.model.out_channels,
        channels=channels, strides=[2] * (len(channels) - 1), num_res_units=getattr(mc, "num_res_units", 2),
        kernel_size=getattr(mc, "kernel_size", 3), norm=_resolve_norm(cfg), dropout=getattr(mc, "dropout", 0.0),
        upsample_mode=getattr(mc, "upsample_mode", "deconv"),
        upsample_interp_mode=getattr(mc, "upsample_interp_mode", "linear"),
        upsample_align_corners=getattr(mc, "upsample_align_corners", True))
    return MONAIModelWrapper(model)


__all__ = ["MONAIModelWrapper", "UpsampleModeUNet", "UNet", "ResidualUnit", "Convolution", "ADN", "SkipConnection",
           "build_monai_unet"]
