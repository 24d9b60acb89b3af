"""Oracle: MedNeXt forward on the CPU (PyTorch fp32, functional, NCDHW).

TEST INFRASTRUCTURE (see oracle/__init__.py).  **Parity unpinned**: the reference delegates
this arithmetic to the third-party ``nnunet_mednext`` package
(connectomics/models/architectures/mednext_models.py:24-25; factory call :374-380; full
constructor :449-479), which is neither vendored nor version-pinned nor installed here.
This file restates the *published* architecture (Roy et al., "MedNeXt", MICCAI 2023,
arXiv 2303.09975) using the attribute names the reference reads from the package:

  trunk attrs   stem, enc_block_{0..3}, down_{0..3}, bottleneck, up_{3..0}, dec_block_{3..0},
                out_{0..4}, dummy_tensor, do_ds                 (mednext_models.py:215-231, 79-87)
  block attrs   conv1 (depthwise k^3), norm (GroupNorm(C,C) | channel LayerNorm), conv2 (C->rC),
                conv3 (rC->C_out), do_res, grn_{beta,gamma}      (mednext_models.py:104-126)
  down block    conv1 stride 2; optional res_conv = Conv3d(C, 2C, 1, stride 2)
  up block      conv1 = depthwise ConvTranspose3d(k, stride 2, pad k//2); optional
                res_conv = ConvTranspose3d(C, C/2, 1, stride 2); both results get one voxel of
                zero padding in FRONT of every spatial axis so that the size doubles exactly
  out block     ConvTranspose3d(C, n_classes, 1)  (== a 1x1x1 convolution)

Known-answer pins used by tests: parameter counts S/B/M/L x k3/k5
(mednext_models.py:309-312), forward_output(forward_features(x)) == forward(x)
(tests/unit/test_mednext_features.py:39), deep supervision returns 5 tensors (:52-55).
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch
import torch.nn.functional as F

# size table of create_mednext_v1 (assumed from the upstream project)
SIZES = {
    "S": dict(exp_r=[2] * 9, block_counts=[2] * 9, checkpoint=None),
    "B": dict(exp_r=[2, 3, 4, 4, 4, 4, 4, 3, 2], block_counts=[2] * 9, checkpoint=None),
    "M": dict(exp_r=[2, 3, 4, 4, 4, 4, 4, 3, 2], block_counts=[3, 4, 4, 4, 4, 4, 4, 4, 3],
              checkpoint="outside_block"),
    "L": dict(exp_r=[3, 4, 8, 8, 8, 8, 8, 4, 3], block_counts=[3, 4, 8, 8, 8, 8, 8, 4, 3],
              checkpoint="outside_block"),
}


def topology(n_channels: int, exp_r, block_counts: Sequence[int]):
    """Ordered description of the trunk: list of (kind, name, c_in, c_out, exp_r, n_blocks)."""
    if isinstance(exp_r, int):
        exp_r = [exp_r] * 9
    n = n_channels
    t = []
    for lvl in range(4):
        t.append(("blocks", f"enc_block_{lvl}", n << lvl, n << lvl, exp_r[lvl], block_counts[lvl]))
        t.append(("down", f"down_{lvl}", n << lvl, n << (lvl + 1), exp_r[lvl + 1], 1))
    t.append(("blocks", "bottleneck", n << 4, n << 4, exp_r[4], block_counts[4]))
    for i, lvl in enumerate((3, 2, 1, 0)):
        t.append(("up", f"up_{lvl}", n << (lvl + 1), n << lvl, exp_r[5 + i], 1))
        t.append(("blocks", f"dec_block_{lvl}", n << lvl, n << lvl, exp_r[5 + i], block_counts[5 + i]))
    return t


def init_state(in_channels=1, n_channels=32, n_classes=1, exp_r=2, kernel_size=3,
               block_counts=(2,) * 9, deep_supervision=False, do_res_up_down=True,
               grn=False, norm_type="group", seed=0, dim="3d") -> Dict[str, torch.Tensor]:
    """Random parameters with the upstream key names and PyTorch-layout shapes.  dim='2d': the Conv2d /
    ConvTranspose2d twin the constructor builds for `dim='2d'` (mednext_models.py:449-467); every function
    below picks the 2-D or 3-D functional op from the rank of the weight it is given."""
    g = torch.Generator().manual_seed(seed)
    k = int(kernel_size)
    nd = {"2d": 2, "3d": 3}[dim]
    st: Dict[str, torch.Tensor] = {}

    def rnd(*shape, scale=1.0):
        return (torch.rand(*shape, generator=g) * 2 - 1) * scale

    def conv(name, cout, cin_per_group, ks, fan_in):
        b = 1.0 / (fan_in ** 0.5)
        st[name + ".weight"] = rnd(cout, cin_per_group, *([ks] * nd), scale=b)
        st[name + ".bias"] = rnd(cout, scale=b)

    def block(prefix, cin, cout, r, transposed_res=None):
        conv(prefix + ".conv1", cin, 1, k, k ** 3)
        st[prefix + ".norm.weight"] = 1.0 + 0.1 * rnd(cin)
        st[prefix + ".norm.bias"] = 0.1 * rnd(cin)
        conv(prefix + ".conv2", r * cin, cin, 1, cin)
        conv(prefix + ".conv3", cout, r * cin, 1, r * cin)
        if grn:
            st[prefix + ".grn_beta"] = 0.1 * rnd(1, r * cin, *([1] * nd))
            st[prefix + ".grn_gamma"] = 0.1 * rnd(1, r * cin, *([1] * nd))

    conv("stem", n_channels, in_channels, 1, in_channels)
    for kind, name, cin, cout, r, nb in topology(n_channels, exp_r, list(block_counts)):
        if kind == "blocks":
            for i in range(nb):
                block(f"{name}.{i}", cin, cout, r)
        else:
            block(name, cin, cout, r)
            if do_res_up_down:
                if kind == "down":
                    conv(name + ".res_conv", cout, cin, 1, cin)
                else:  # ConvTranspose3d weight layout is (C_in, C_out, 1,1,1)
                    b = 1.0 / (cout ** 0.5)
                    st[name + ".res_conv.weight"] = rnd(cin, cout, *([1] * nd), scale=b)
                    st[name + ".res_conv.bias"] = rnd(cout, scale=b)
    heads = [0] + ([1, 2, 3, 4] if deep_supervision else [])
    for h in heads:
        c = n_channels << h
        b = 1.0 / (n_classes ** 0.5)
        st[f"out_{h}.conv_out.weight"] = rnd(c, n_classes, *([1] * nd), scale=b)  # ConvTranspose layout
        st[f"out_{h}.conv_out.bias"] = rnd(n_classes, scale=b)
    st["dummy_tensor"] = torch.ones(1)
    return st


def _conv(x, w, b, **kw):
    return (F.conv2d if w.dim() == 4 else F.conv3d)(x, w, b, **kw)


def _convT(x, w, b, **kw):
    return (F.conv_transpose2d if w.dim() == 4 else F.conv_transpose3d)(x, w, b, **kw)


def _front_pad(y):
    return F.pad(y, (1, 0) * (y.dim() - 2))


def _norm(t, st, prefix, norm_type):
    c = t.shape[1]
    if norm_type == "group":
        return F.group_norm(t, c, st[prefix + ".norm.weight"], st[prefix + ".norm.bias"], eps=1e-5)
    # channels_first LayerNorm over C per voxel (ConvNeXt style, eps 1e-5)
    u = t.mean(1, keepdim=True)
    s = (t - u).pow(2).mean(1, keepdim=True)
    tn = (t - u) / torch.sqrt(s + 1e-5)
    bc = (1, -1) + (1,) * (t.dim() - 2)
    return st[prefix + ".norm.weight"].view(bc) * tn + st[prefix + ".norm.bias"].view(bc)


def _mlp(t, st, prefix, norm_type, grn):
    h = F.gelu(_conv(_norm(t, st, prefix, norm_type), st[prefix + ".conv2.weight"],
                     st[prefix + ".conv2.bias"]))
    if grn:
        gx = torch.norm(h, p=2, dim=tuple(range(2, h.dim())), keepdim=True)
        nx = gx / (gx.mean(dim=1, keepdim=True) + 1e-6)
        h = st[prefix + ".grn_gamma"] * (h * nx) + st[prefix + ".grn_beta"] + h
    return _conv(h, st[prefix + ".conv3.weight"], st[prefix + ".conv3.bias"])


def block_forward(x, st, prefix, k, do_res=True, norm_type="group", grn=False):
    t = _conv(x, st[prefix + ".conv1.weight"], st[prefix + ".conv1.bias"], padding=k // 2,
              groups=x.shape[1])
    y = _mlp(t, st, prefix, norm_type, grn)
    return x + y if do_res else y


def down_forward(x, st, prefix, k, norm_type="group", grn=False):
    t = _conv(x, st[prefix + ".conv1.weight"], st[prefix + ".conv1.bias"], stride=2,
              padding=k // 2, groups=x.shape[1])
    y = _mlp(t, st, prefix, norm_type, grn)
    if prefix + ".res_conv.weight" in st:
        y = y + _conv(x, st[prefix + ".res_conv.weight"], st[prefix + ".res_conv.bias"], stride=2)
    return y


def up_forward(x, st, prefix, k, norm_type="group", grn=False):
    t = _convT(x, st[prefix + ".conv1.weight"], st[prefix + ".conv1.bias"], stride=2,
               padding=k // 2, groups=x.shape[1])
    y = _front_pad(_mlp(t, st, prefix, norm_type, grn))
    if prefix + ".res_conv.weight" in st:
        r = _convT(x, st[prefix + ".res_conv.weight"], st[prefix + ".res_conv.bias"], stride=2)
        y = y + _front_pad(r)
    return y


def forward_features(st, x, *, n_channels=32, exp_r=2, kernel_size=3, block_counts=(2,) * 9,
                     do_res=True, norm_type="group", grn=False, collect: List | None = None):
    """stem -> encoder -> bottleneck -> decoder; returns the full-resolution feature map.
    If `collect` is a list, decoder-level features (deepest first: bottleneck, dec_3, dec_2,
    dec_1) are appended for the deep-supervision heads."""
    k = int(kernel_size)
    x = _conv(x, st["stem.weight"], st["stem.bias"])
    skips = {}
    for kind, name, cin, cout, r, nb in topology(n_channels, exp_r, list(block_counts)):
        if kind == "blocks":
            for i in range(nb):
                x = block_forward(x, st, f"{name}.{i}", k, do_res, norm_type, grn)
            if name.startswith("enc_block_"):
                skips[int(name[-1])] = x
            elif collect is not None and name != "dec_block_0":
                collect.append(x)
        elif kind == "down":
            x = down_forward(x, st, name, k, norm_type, grn)
        else:
            x = skips.pop(int(name[-1])) + up_forward(x, st, name, k, norm_type, grn)
    return x


def forward_output(st, feat, head=0):
    return _convT(feat, st[f"out_{head}.conv_out.weight"], st[f"out_{head}.conv_out.bias"])


def forward(st, x, *, deep_supervision=False, **kw):
    feats: List = [] if deep_supervision else None
    f = forward_features(st, x, collect=feats, **kw)
    out = forward_output(st, f, 0)
    if not deep_supervision:
        return out
    # feats = [bottleneck, dec_3, dec_2, dec_1] -> heads out_4, out_3, out_2, out_1
    ds = [forward_output(st, ft, h) for ft, h in zip(feats, (4, 3, 2, 1))]
    return [out, ds[3], ds[2], ds[1], ds[0]]


def param_count(st) -> int:
    return sum(int(v.numel()) for v in st.values())
