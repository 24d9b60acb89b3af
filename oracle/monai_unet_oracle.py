"""TEST INFRASTRUCTURE ONLY -- CPU restatement (PyTorch-CPU fp32, torch.nn.functional) of the `monai_unet` architecture.

PARITY UNPINNED: the arithmetic lives in the third-party `monai` package (un-vendored, not installed here; the reference
pins `monai>=1.3` only loosely), which the reference reaches through
connectomics/models/architectures/monai_models.py:231-248 (`UpsampleModeUNet(spatial_dims, in_channels, out_channels,
channels, strides=[2]*(L-1), num_res_units, kernel_size, norm, dropout, upsample_mode='deconv')`).  This file restates
the published `monai.networks.nets.UNet` / `blocks.ResidualUnit` / `blocks.Convolution` / `blocks.ADN` composition:

    block(level)  = down -> cat([x, sub-block(x)]) -> up
    down          = ResidualUnit: cx = (conv k s p=(k-1)/2 -> norm -> dropout -> PReLU) x subunits (stride on the first);
                    res = conv(k, s, p) if strided else conv(1) if widths differ else identity;  out = cx + res
    up            = ConvTranspose(k, s, p, output_padding = s-1) -> norm -> PReLU, then ResidualUnit(subunits 1,
                    last_conv_only at the top level: no norm / activation on the output)
    num_res_units = 0: down = one conv -> norm -> PReLU, up = the transposed conv (-> norm -> PReLU below the top level) alone

anchored on the reference's own shape tests (tests/unit/test_registry_basic.py:64-138: output shape == input spatial
shape, out_channels channels) and on the state-dict key vocabulary.  It walks a state dict by those keys, so it does not
import or call any product code.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F


def _norm(st: Dict[str, torch.Tensor], prefix: str, x: torch.Tensor, kind: str, training: bool, groups: int, eps: float = 1e-5):
    w, b = st.get(prefix + ".weight"), st.get(prefix + ".bias")
    if kind == "batch":
        rm, rv = st[prefix + ".running_mean"], st[prefix + ".running_var"]
        if training:
            return F.batch_norm(x, None, None, w, b, True, 0.1, eps)
        return F.batch_norm(x, rm, rv, w, b, False, 0.1, eps)
    if kind == "instance":
        return F.instance_norm(x, None, None, w, b, True, 0.1, eps)
    if kind == "group":
        return F.group_norm(x, groups, w, b, eps)
    raise ValueError(kind)


def _convolution(st, prefix, x, *, stride, transposed, kind, training, groups):
    w, b = st[prefix + ".conv.weight"], st.get(prefix + ".conv.bias")
    k = w.shape[-1]
    pad = (k - 1) // 2
    if transposed:
        y = F.conv_transpose3d(x, w, b, stride=stride, padding=pad, output_padding=stride - 1)
    else:
        y = F.conv3d(x, w, b, stride=stride, padding=pad)
    if prefix + ".adn.A.weight" in st:
        y = _norm(st, prefix + ".adn.N", y, kind, training, groups)
        y = F.prelu(y, st[prefix + ".adn.A.weight"])
    return y


def _residual_unit(st, prefix, x, *, stride, kind, training, groups):
    res = x
    if prefix + ".residual.weight" in st:
        w = st[prefix + ".residual.weight"]
        k = w.shape[-1]
        res = F.conv3d(x, w, st.get(prefix + ".residual.bias"), stride=stride, padding=(k - 1) // 2 if k > 1 else 0)
    cx = x
    su = 0
    while f"{prefix}.conv.unit{su}.conv.weight" in st:
        cx = _convolution(st, f"{prefix}.conv.unit{su}", cx, stride=stride if su == 0 else 1, transposed=False, kind=kind,
                          training=training, groups=groups)
        su += 1
    return cx + res


def _down(st, prefix, x, *, stride, kind, training, groups):
    """UNet._get_down_layer: a ResidualUnit (keys `<prefix>.conv.unitN...`), or with num_res_units = 0 one Convolution
    (keys `<prefix>.conv.weight`)."""
    if prefix + ".conv.weight" in st:
        return _convolution(st, prefix, x, stride=stride, transposed=False, kind=kind, training=training, groups=groups)
    return _residual_unit(st, prefix, x, stride=stride, kind=kind, training=training, groups=groups)


def _block(st, prefix, x, levels_left: int, *, kind, training, groups):
    """prefix addresses a Sequential(down, SkipConnection(sub), up)."""
    d = _down(st, prefix + ".0", x, stride=2, kind=kind, training=training, groups=groups)
    sub_prefix = prefix + ".1.submodule"
    if levels_left > 1:
        s = _block(st, sub_prefix, d, levels_left - 1, kind=kind, training=training, groups=groups)
    else:                                                   # bottom layer: a stride-1 down layer
        s = _down(st, sub_prefix, d, stride=1, kind=kind, training=training, groups=groups)
    c = torch.cat([d, s], 1)
    if prefix + ".2.conv.weight" in st:                     # num_res_units = 0: the up layer is the transposed Convolution alone
        return _convolution(st, prefix + ".2", c, stride=2, transposed=True, kind=kind, training=training, groups=groups)
    u = _convolution(st, prefix + ".2.0", c, stride=2, transposed=True, kind=kind, training=training, groups=groups)
    return _residual_unit(st, prefix + ".2.1", u, stride=1, kind=kind, training=training, groups=groups)


def forward(st: Dict[str, torch.Tensor], x: torch.Tensor, *, n_levels: int, norm: str = "batch", training: bool = False,
            num_groups: int = 8, prefix: str = "model.model") -> torch.Tensor:
    """st: state dict of MONAIModelWrapper (keys `model.model.0...`); x (B, C, D, H, W) fp32 CPU; n_levels = len(filters);
    training=True uses batch statistics for BatchNorm (running buffers are not updated here)."""
    return _block(st, prefix, x, n_levels - 1, kind=norm, training=training, groups=num_groups)
