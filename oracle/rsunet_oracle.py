"""Oracle: RSUNet forward on the CPU (PyTorch fp32, functional over a state dict).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates
connectomics/models/architectures/rsunet.py: NormAct :73-118, ResBlock :121-154,
ConvBlock :157-198 (pre -> res -> post, each = norm-act then bias-free conv), DownBlock
:201-222 (max-pool then ConvBlock), UpBlock :225-259 (fixed-weight depthwise transposed
conv "bilinear" upsampling :33-70, 1x1 projection, ADD the skip, ConvBlock), trunk
:262-448 (final norm-act + 1x1 head with bias, optional deep-supervision heads).
State-dict keys are the reference module tree's keys, so the reference's own weights can be
fed in directly.  PINNED by tests/golden/rsunet_*.npz (reference outputs).
"""
from __future__ import annotations

import math
from typing import Dict, Sequence

import torch
import torch.nn.functional as F


def bilinear_kernel(channels: int, factor: Sequence[int]) -> torch.Tensor:
    """rsunet.py:41-70 -- kernel (2f - f%2) per axis, weight depends on (h, w) only."""
    ks = [2 * f - f % 2 for f in factor]
    w = torch.zeros(channels, 1, *ks)
    width, height = ks[2], ks[1]
    f = float(math.ceil(width / 2.0))
    c = float(width - 1) / (2.0 * f)
    for iw in range(width):
        for ih in range(height):
            w[..., ih, iw] = (1 - abs(iw / f - c)) * (1 - abs(ih / f - c))
    return w


def _norm_act(x, st, prefix, norm, activation, num_groups, act_kwargs):
    c = x.shape[1]
    if norm == "batch" and act_kwargs.get("bn_training", False):   # training-mode statistics (no running update)
        x = F.batch_norm(x, None, None, st[prefix + ".norm.weight"], st[prefix + ".norm.bias"], True, 0.0, 1e-5)
    elif norm == "batch":   # eval-mode statistics
        x = F.batch_norm(x, st[prefix + ".norm.running_mean"], st[prefix + ".norm.running_var"],
                         st[prefix + ".norm.weight"], st[prefix + ".norm.bias"], False, 0.0, 1e-5)
    elif norm == "group":
        g = min(num_groups, c)
        while c % g:
            g -= 1
        x = F.group_norm(x, g, st[prefix + ".norm.weight"], st[prefix + ".norm.bias"], 1e-5)
    elif norm == "instance":
        x = F.instance_norm(x, eps=1e-5)
    elif norm != "none":
        raise ValueError(f"Unknown normalization: {norm}")
    # rsunet.py:103-113 builds ReLU/LeakyReLU/ELU with inplace=True.  With norm == "none" the
    # norm is an Identity, so the activation overwrites its INPUT tensor -- which the residual
    # branch of ResBlock (:150-154) still references.  Reproduce that aliasing faithfully.
    inplace = norm == "none"
    if activation == "relu":
        return F.relu(x, inplace=inplace)
    if activation == "leakyrelu":
        return F.leaky_relu(x, act_kwargs.get("negative_slope", 0.01), inplace=inplace)
    if activation == "prelu":
        return F.prelu(x, st[prefix + ".act.weight"])
    if activation == "elu":
        return F.elu(x, act_kwargs.get("alpha", 1.0), inplace=inplace)
    raise ValueError(f"Unknown activation: {activation}")


def _conv(x, w, ks):
    return F.conv3d(x, w, None, padding=tuple(k // 2 for k in ks))


def _conv_block(x, st, p, ks, na):
    x = _conv(_norm_act(x, st, p + ".pre.0", *na), st[p + ".pre.1.weight"], ks)
    r = x
    x = _conv(_norm_act(x, st, p + ".res.norm_act1", *na), st[p + ".res.conv1.weight"], ks)
    x = _conv(_norm_act(x, st, p + ".res.norm_act2", *na), st[p + ".res.conv2.weight"], ks) + r
    return _conv(_norm_act(x, st, p + ".post.0", *na), st[p + ".post.1.weight"], ks)


def forward(st: Dict[str, torch.Tensor], x: torch.Tensor, *, width, down_factors=None,
            kernel_sizes=3, norm="batch", activation="relu", num_groups=8,
            deep_supervision=False, depth_2d=0, kernel_2d=(1, 3, 3), **act_kwargs):
    depth = len(width) - 1
    if down_factors is None:
        down_factors = [(1, 2, 2)] * depth
    if isinstance(kernel_sizes, int):
        kernel_sizes = [kernel_sizes] * len(width)
    kernel_sizes = list(kernel_sizes) + [kernel_sizes[-1]] * (len(width) - len(kernel_sizes))
    for i in range(min(depth_2d, len(kernel_sizes))):
        kernel_sizes[i] = tuple(kernel_2d)
    ks = [(k, k, k) if isinstance(k, int) else tuple(k) for k in kernel_sizes]
    na = (norm, activation, num_groups, act_kwargs)

    x = _conv_block(x, st, "input_conv", ks[0], na)
    skips = []
    for d in range(depth):
        skips.append(x)
        x = F.max_pool3d(x, tuple(down_factors[d]))
        x = _conv_block(x, st, f"down_blocks.{d}.conv", ks[d + 1], na)
    n_ds = min(4, depth) if deep_supervision else 0
    ds_feats = []
    for i, d in enumerate(reversed(range(depth))):
        if deep_supervision and (depth - i - 1) < n_ds:
            ds_feats.append(x)
        fac = tuple(down_factors[d])
        pad = [int(math.ceil((f - 1) / 2.0)) for f in fac]
        wk = st.get(f"up_blocks.{i}.up.weight")
        if wk is None:
            wk = bilinear_kernel(x.shape[1], fac)
        x = F.conv_transpose3d(x, wk, stride=fac, padding=pad, groups=x.shape[1])
        x = F.conv3d(x, st[f"up_blocks.{i}.proj.weight"]) + skips.pop()
        x = _conv_block(x, st, f"up_blocks.{i}.conv", ks[d], na)
    x = _norm_act(x, st, "final_norm", *na)
    out = F.conv3d(x, st["output_head.weight"], st["output_head.bias"])
    if not deep_supervision:
        return out
    res = {"output": out}
    for i, ft in enumerate(ds_feats[:n_ds]):
        res[f"ds_{i + 1}"] = F.conv3d(ft, st[f"ds_heads.{i}.weight"], st[f"ds_heads.{i}.bias"])
    return res
