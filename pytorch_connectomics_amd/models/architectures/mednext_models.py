"""MedNeXt wrappers and registry builders -- same public surface as the reference's
connectomics/models/architectures/mednext_models.py (MedNeXtWrapper :38-89, MedNeXtTaskHead
:129-194, MedNeXtMultiHeadWrapper :197-273, build_mednext :303-397, build_mednext_custom
:400-483, upkern_load_weights :487-537), built on the in-repo MI355X trunk
(``.mednext``) instead of the external ``nnunet_mednext`` package.
"""
from __future__ import annotations

import os

from dataclasses import dataclass
from typing import Any, Dict, Mapping, Optional, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from .base import ConnectomicsModel
from .mednext import (MedNeXt as MedNeXtBase, MedNeXtBlock, create_mednext_v1, to_channels_first,
                      to_channels_last)
from .registry import register_architecture

MEDNEXT_AVAILABLE = True


class MedNeXtWrapper(ConnectomicsModel):
    """Calls the trunk; turns the deep-supervision list of 5 into {"output", "ds_1".."ds_4"}."""

    def __init__(self, model: nn.Module, deep_supervision: bool = False):
        super().__init__()
        self.model = model
        self.supports_deep_supervision = deep_supervision
        self.output_scales = 5 if deep_supervision else 1

    def forward(self, x: torch.Tensor) -> Union[torch.Tensor, Dict[str, torch.Tensor]]:
        outputs = self.model(x)
        if self.supports_deep_supervision and isinstance(outputs, list):
            return {"output": outputs[0], "ds_1": outputs[1], "ds_2": outputs[2], "ds_3": outputs[3],
                    "ds_4": outputs[4]}
        return outputs

    def forward_cl(self, x_cl: torch.Tensor) -> torch.Tensor:
        """Channels-last fast path used by the on-device sliding-window engine."""
        return self.model.forward_cl(x_cl)


def _cfg_value(cfg: Any, key: str, default: Any = None) -> Any:
    return cfg.get(key, default) if isinstance(cfg, Mapping) else getattr(cfg, key, default)


_NORM_NAME = {nn.GroupNorm: "group"}


def _infer_mednext_head_block_kwargs(model: nn.Module) -> dict:
    """Constructor arguments of a task-head block, read off the trunk's last decoder block (mednext_models.py:99-126)."""
    blocks = getattr(model, "dec_block_0", None)
    if blocks is None or len(blocks) == 0:
        raise ValueError("MedNeXt trunk must expose a non-empty dec_block_0 to build task heads.")
    ref = blocks[0]
    if not isinstance(ref, MedNeXtBlock):
        raise TypeError("Expected MedNeXt dec_block_0 to contain MedNeXtBlock instances for multi-head reuse.")
    k = ref.conv1.kernel_size
    return dict(exp_r=ref.conv2.out_channels // ref.conv2.in_channels, kernel_size=int(k[0] if isinstance(k, tuple) else k),
                do_res=ref.do_res, norm_type=_NORM_NAME.get(type(ref.norm), "layer"), dim=ref.dim, grn=ref.grn)


# narrow task heads run as one block-diagonal 32-channel head at inference (MedNeXtMultiHeadWrapper._merged_heads); 0 = one pass per head
MERGE_NARROW_HEADS = os.environ.get("PYTC_MERGE_HEADS", "1") != "0"
FUSE_MERGED_HEAD_PROJECTION = os.environ.get("PYTC_FUSE_MERGED_HEAD", "1") != "0"


@dataclass(frozen=True)
class _HeadSpec:
    """One entry of `model.heads`: a mapping / namespace with out_channels [, num_blocks, hidden_channels], or a bare int."""
    out_channels: int
    num_blocks: int = 0
    hidden_channels: Optional[int] = None

    @classmethod
    def parse(cls, entry: Any) -> "_HeadSpec":
        hidden = _cfg_value(entry, "hidden_channels", None)
        return cls(int(_cfg_value(entry, "out_channels", entry)), int(_cfg_value(entry, "num_blocks", 0)),
                   None if hidden is None else int(hidden))

    def as_dict(self, feature_channels: int) -> dict:
        return {"out_channels": self.out_channels, "num_blocks": self.num_blocks,
                "hidden_channels": self.hidden_channels or feature_channels}


class MedNeXtTaskHead(nn.Module):
    """Optional 1x1 in-projection -> N MedNeXt blocks -> 1x1 out-projection on the shared features."""

    def __init__(self, in_channels: int, out_channels: int, num_blocks: int, hidden_channels: int | None = None,
                 *, exp_r: int, kernel_size: int, do_res: bool, norm_type: str, dim: str, grn: bool):
        super().__init__()
        width = in_channels if hidden_channels is None else hidden_channels
        # (condition that must hold, message) in the order the reference raises them
        for ok, msg in (
                (num_blocks >= 0, f"MedNeXt task head num_blocks must be >= 0, got {num_blocks}"),
                (out_channels > 0, f"MedNeXt task head out_channels must be positive, got {out_channels}"),
                (width > 0, f"MedNeXt task head hidden_channels must be positive, got {width}"),
                (width <= in_channels, "MedNeXt task head hidden_channels must not exceed the shared feature width "
                                       f"({width} > {in_channels})"),
                (dim in ("2d", "3d"), f"MedNeXt task head dim must be '2d' or '3d', got {dim}")):
            if not ok:
                raise ValueError(msg)
        conv = nn.Conv3d if dim == "3d" else nn.Conv2d
        block_kw = dict(exp_r=exp_r, kernel_size=kernel_size, do_res=do_res, norm_type=norm_type, dim=dim, grn=grn)
        self.input_projection = nn.Identity() if width == in_channels else conv(in_channels, width, kernel_size=1)
        self.blocks = (nn.Sequential(*(MedNeXtBlock(width, width, **block_kw) for _ in range(num_blocks)))
                       if num_blocks else nn.Identity())
        self.projection = conv(width, out_channels, kernel_size=1)
        self.hidden_channels = width

    def forward_cl(self, hip, feat_cl: torch.Tensor) -> torch.Tensor:
        x = feat_cl
        if not isinstance(self.input_projection, nn.Identity):
            x = hip.pointwise(x, self.input_projection)
        if not isinstance(self.blocks, nn.Identity):
            for blk in self.blocks:
                x = hip.block(blk, x)
        return hip.pointwise(x, self.projection, out_dtype=torch.float32)

    def forward(self, x: torch.Tensor) -> torch.Tensor:  # pragma: no cover - guard only
        raise RuntimeError("MedNeXtTaskHead executes through MedNeXtMultiHeadWrapper.forward (HIP engine)")


class MedNeXtMultiHeadWrapper(ConnectomicsModel):
    """Output contract {"output": {head_name: tensor}}; no deep supervision (reference v1)."""

    def __init__(self, model: nn.Module, heads: Mapping[str, Any], *, primary_head: str | None = None):
        super().__init__()
        if getattr(model, "do_ds", False):
            raise ValueError("MedNeXtMultiHeadWrapper does not support deep supervision yet. "
                             "Disable deep supervision for the trunk first.")
        if not hasattr(model, "forward_features"):
            raise ValueError("MedNeXt trunk must expose forward_features() before using MedNeXtMultiHeadWrapper.")
        if not heads:
            raise ValueError("MedNeXtMultiHeadWrapper requires at least one named task head.")
        self.model = model
        self.supports_deep_supervision = False
        self.output_scales = 1
        self.feature_channels = int(model.stem.out_channels)
        self.head_block_kwargs = _infer_mednext_head_block_kwargs(model)
        specs = {name: _HeadSpec.parse(entry) for name, entry in heads.items()}
        self.heads = nn.ModuleDict({name: MedNeXtTaskHead(self.feature_channels, sp.out_channels, sp.num_blocks,
                                                          sp.hidden_channels, **self.head_block_kwargs)
                                    for name, sp in specs.items()})
        self.head_specs = {name: sp.as_dict(self.feature_channels) for name, sp in specs.items()}
        self.primary_head = primary_head if primary_head else next(iter(specs))
        if self.primary_head not in self.heads:
            raise ValueError(f"primary_head '{self.primary_head}' is not one of the configured heads: "
                             f"{sorted(self.heads.keys())}")

    def forward_features(self, x: torch.Tensor) -> torch.Tensor:
        return self.model.forward_features(x)

    def forward_heads_cl(self, feat_cl: torch.Tensor) -> Dict[str, torch.Tensor]:
        hip = self.model._hip
        y = self._merged_heads_cl(feat_cl)
        if y is not None:
            outs, c0 = {}, 0
            for name, head in self.heads.items():
                c1 = c0 + head.projection.out_channels
                outs[name] = y[..., c0:c1]
                c0 = c1
            return outs
        return {name: head.forward_cl(hip, feat_cl) for name, head in self.heads.items()}

    def _merged_heads_cl(self, feat_cl: torch.Tensor) -> Optional[torch.Tensor]:
        """All heads' outputs concatenated along C in declaration order through the merged head, or None (see _merged_heads)."""
        merged = self._merged_heads(feat_cl) if (MERGE_NARROW_HEADS and not torch.is_grad_enabled()) else None
        if merged is None:
            return None
        return self._run_merged(merged, hip_in=None, feat_cl=feat_cl)

    def _run_merged(self, merged, hip_in: Optional[torch.Tensor], feat_cl: Optional[torch.Tensor]) -> torch.Tensor:
        """the merged head on the trunk's features, or (hip_in) on features the trunk's last mixer has already projected"""
        hip = self.model._hip
        x = hip_in if hip_in is not None else hip.pointwise(feat_cl, merged.input_projection)
        for i, blk in enumerate(merged.blocks):
            # the out-projection rides in the last mixer's epilogue where the fused kernel can carry it (the trunk's own output conv does the
            # same): (None, logits fp32) comes back and the block's 64 B / voxel are never written
            x = hip.block(blk, x, head=merged.projection_t if (i + 1 == len(merged.blocks) and FUSE_MERGED_HEAD_PROJECTION) else None)
        if isinstance(x, tuple):
            return x[1]
        return hip.pointwise(x, merged.projection, out_dtype=torch.float32)

    def _merged_heads(self, feat_cl: torch.Tensor):
        """Narrow task heads (MitoEM: three heads of 8 hidden channels, one block each -- tutorials/mitoEM/common.yaml:10-39) as ONE
        32-channel head with block-diagonal weights: head i owns channels [i*w, (i+1)*w) of the merged in-projection / depthwise conv /
        GroupNorm(C, C) (per-channel statistics: grouping heads changes nothing) and the matching diagonal blocks of conv2 / conv3 / the
        out-projection; padded channels and hidden units carry zero weights and biases (GELU(0) = 0).  Same function, but on the kernels of
        a level-0 trunk block (matrix-core depthwise conv, MFMA mixer 32 -> 32 exp_r -> 32) instead of three passes of 8-channel
        kernels over the full-resolution tensor: MedNeXt-L 2 x 160^3: 4.4 -> 1.1 ms of a 19 ms forward (profiles/r05_mednext_l_*).
        bf16 inference only; returns None when the heads are not of that shape (then each head runs on its own)."""
        heads = list(self.heads.values())
        if feat_cl.dtype != torch.bfloat16 or feat_cl.dim() != 5 or len(heads) < 2:
            return None
        kw = self.head_block_kwargs
        w, nb = heads[0].hidden_channels, (len(heads[0].blocks) if not isinstance(heads[0].blocks, nn.Identity) else 0)
        feat, Wp = self.feature_channels, 32
        if (feat != 32 or kw["dim"] != "3d" or kw["grn"] or kw["norm_type"] != "group" or kw["kernel_size"] != 3 or nb < 1
                or len(heads) * w > Wp or (kw["exp_r"] * Wp) not in (64, 96, 128)
                or any(isinstance(h.input_projection, nn.Identity) or h.hidden_channels != w
                       or (0 if isinstance(h.blocks, nn.Identity) else len(h.blocks)) != nb for h in heads)):
            return None
        params = [p_ for h in heads for p_ in h.parameters()]
        key = tuple((p_.data_ptr(), p_._version) for p_ in params) + (str(feat_cl.device),)
        cached = self.__dict__.get("_merged_cache")
        if cached is not None and cached[0] == key:
            return cached[1]
        e, dev = kw["exp_r"] * w, feat_cl.device
        with torch.no_grad():
            m = nn.Module()
            m.input_projection = nn.Conv3d(feat, Wp, kernel_size=1).to(dev)
            m.projection = nn.Conv3d(Wp, sum(h.projection.out_channels for h in heads), kernel_size=1).to(dev)
            m.blocks = nn.ModuleList(MedNeXtBlock(Wp, Wp, **kw).to(dev) for _ in range(nb))
            for t in m.parameters():
                t.zero_()
            c0 = 0
            for i, h in enumerate(heads):
                ch, hid = slice(i * w, (i + 1) * w), slice(i * e, (i + 1) * e)
                m.input_projection.weight[ch] = h.input_projection.weight
                m.input_projection.bias[ch] = h.input_projection.bias
                for mb, hb in zip(m.blocks, h.blocks):
                    mb.conv1.weight[ch] = hb.conv1.weight
                    mb.conv1.bias[ch] = hb.conv1.bias
                    mb.norm.weight[ch], mb.norm.bias[ch] = hb.norm.weight, hb.norm.bias
                    mb.conv2.weight[hid, ch] = hb.conv2.weight
                    mb.conv2.bias[hid] = hb.conv2.bias
                    mb.conv3.weight[ch, hid] = hb.conv3.weight
                    mb.conv3.bias[ch] = hb.conv3.bias
                c1 = c0 + h.projection.out_channels
                m.projection.weight[c0:c1, ch] = h.projection.weight
                m.projection.bias[c0:c1] = h.projection.bias
                c0 = c1
            # the same projection in the layout of the trunk's output conv (ConvTranspose3d: weight (in, out, 1, 1, 1)), which the fused head takes
            m.projection_t = nn.ConvTranspose3d(Wp, m.projection.out_channels, kernel_size=1).to(dev)
            m.projection_t.weight.copy_(m.projection.weight.reshape(m.projection.out_channels, Wp).t().reshape(Wp, -1, 1, 1, 1))
            m.projection_t.bias.copy_(m.projection.bias)
            m.eval()
            for t in m.parameters():
                t.requires_grad_(False)
        self.__dict__["_merged_cache"] = (key, m)          # not a sub-module: no entry in state_dict(), rebuilt when a head parameter changes
        return m

    def forward_heads(self, features: torch.Tensor) -> Dict[str, torch.Tensor]:
        from .mednext import resolve_compute_dtype
        dt = resolve_compute_dtype(self.model.compute_dtype)
        features, flat = self.model._lift(features)
        outs = self.forward_heads_cl(to_channels_last(features).to(dt))
        return {k: self.model._drop(to_channels_first(v), flat) for k, v in outs.items()}

    def forward(self, x: torch.Tensor) -> Dict[str, Dict[str, torch.Tensor]]:
        self.model._check_input(x)
        x, flat = self.model._lift(x)
        drop = self.model._drop
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # training: trunk and heads through the autograd Functions of training/autograd.py (HIP forward + backward)
            from ...training.autograd import mednext_multihead_train_forward
            from .mednext import resolve_compute_dtype
            outs = mednext_multihead_train_forward(self, to_channels_last(x.float()), resolve_compute_dtype(self.model.compute_dtype))
            return {"output": {k: drop(to_channels_first(v), flat) for k, v in outs.items()}}
        feat_cl = self.model.features_cl(to_channels_last(x.float()))
        return {"output": {k: drop(to_channels_first(v), flat) for k, v in self.forward_heads_cl(feat_cl).items()}}

    def forward_cl(self, x_cl: torch.Tensor) -> torch.Tensor:
        """Channels-last fast path: all heads concatenated along C in declaration order."""
        from .mednext import resolve_compute_dtype
        merged = None
        if MERGE_NARROW_HEADS and FUSE_MERGED_HEAD_PROJECTION and not torch.is_grad_enabled():
            # the merged input projection rides in the epilogue of the trunk's last mixer where the fused kernel can carry it: the features
            # themselves (64 B / voxel written, read back) never exist
            probe = torch.empty((0, 1, 1, 1, self.feature_channels), dtype=resolve_compute_dtype(self.model.compute_dtype), device=x_cl.device)
            merged = self._merged_heads(probe)
        if merged is not None:
            feat_cl = self.model.features_cl(x_cl, proj=merged.input_projection)
            if isinstance(feat_cl, tuple):
                return self._run_merged(merged, hip_in=feat_cl[1], feat_cl=None)
            return self._run_merged(merged, hip_in=None, feat_cl=feat_cl)
        feat_cl = self.model.features_cl(x_cl)
        y = self._merged_heads_cl(feat_cl)
        if y is not None:
            return y
        outs = self.forward_heads_cl(feat_cl)
        return torch.cat(list(outs.values()), dim=-1) if len(outs) > 1 else next(iter(outs.values()))


def _get_mednext_heads_cfg(cfg):
    heads = getattr(cfg.model, "heads", None)
    return (dict(heads), getattr(cfg.model, "primary_head", None)) if heads else ({}, None)


def _resolve_mednext_num_classes(cfg, head_cfg: Mapping[str, Any]) -> int:
    """Width of the trunk's own output projection: the heads' channels together (at least 1), else model.out_channels."""
    if not head_cfg:
        return int(cfg.model.out_channels)
    return max(1, sum(int(_cfg_value(spec, "out_channels", 0)) for spec in head_cfg.values()))


def _finish(model: nn.Module, head_cfg, primary_head, deep_supervision: bool) -> ConnectomicsModel:
    if head_cfg:
        return MedNeXtMultiHeadWrapper(model, head_cfg, primary_head=primary_head)
    return MedNeXtWrapper(model, deep_supervision=deep_supervision)


_SIZE_HELP = ("Model sizes:\n  - S (Small): 5.6M params\n  - B (Base): 10.5M params\n"
              "  - M (Medium): 17.6M params\n  - L (Large): 61.8M params")


@register_architecture("mednext")
def build_mednext(cfg) -> ConnectomicsModel:
    """MedNeXt with a predefined size: model.mednext.size in S/B/M/L (5.6/10.5/17.6/61.8 M params at k=3),
    model.mednext.kernel_size in 3/5/7, model.loss.deep_supervision, optional model.heads; model.mednext.checkpoint_style
    'outside_block' turns on per-block activation checkpointing (policy: MedNeXt.checkpoint_policy)."""
    mn = cfg.model.mednext
    size, ksize = getattr(mn, "size", "S"), getattr(mn, "kernel_size", 3)
    ds = getattr(getattr(cfg.model, "loss", None), "deep_supervision", False)
    head_cfg, primary_head = _get_mednext_heads_cfg(cfg)
    if size not in ("S", "B", "M", "L"):
        raise ValueError(f"MedNeXt model_size must be 'S', 'B', 'M', or 'L'. Got: {size}\n{_SIZE_HELP}")
    if ksize not in (3, 5, 7):
        raise ValueError(f"MedNeXt kernel_size must be 3, 5, or 7. Got: {ksize}\nRecommended: Start with kernel_size=3")
    model = create_mednext_v1(num_input_channels=cfg.model.in_channels, num_classes=_resolve_mednext_num_classes(cfg, head_cfg),
                              model_id=size, kernel_size=ksize, deep_supervision=ds)
    style = getattr(mn, "checkpoint_style", None)
    if style is not None:
        if style != "outside_block":
            raise ValueError(f"model.mednext.checkpoint_style must be None or 'outside_block', got: {style!r}")
        model.outside_block_checkpointing = True
    return _finish(model, head_cfg, primary_head, ds)


# constructor argument -> (config key under model.mednext, default), mednext_models.py:449-463
_CUSTOM_ARGS = (("n_channels", "base_channels", 32), ("exp_r", "exp_r", 4), ("kernel_size", "kernel_size", 7),
                ("do_res", "do_res", True), ("do_res_up_down", "do_res_up_down", True), ("block_counts", "block_counts", [2] * 9),
                ("checkpoint_style", "checkpoint_style", None), ("norm_type", "norm", "group"), ("dim", "dim", "3d"),
                ("grn", "grn", False))


@register_architecture("mednext_custom")
def build_mednext_custom(cfg) -> ConnectomicsModel:
    """MedNeXt with explicit base_channels / exp_r / kernel_size / block_counts / norm / dim / grn."""
    head_cfg, primary_head = _get_mednext_heads_cfg(cfg)
    params = {arg: getattr(cfg.model.mednext, key, default) for arg, key, default in _CUSTOM_ARGS}
    params.update(in_channels=cfg.model.in_channels, n_classes=_resolve_mednext_num_classes(cfg, head_cfg),
                  deep_supervision=getattr(cfg.model.loss, "deep_supervision", False))
    for ok, msg in ((params["dim"] in ("2d", "3d"), f"mednext_dim must be '2d' or '3d', got: {params['dim']}"),
                    (params["norm_type"] in ("group", "layer"), f"mednext_norm must be 'group' or 'layer', got: {params['norm_type']}"),
                    (len(params["block_counts"]) == 9, "mednext_block_counts must have exactly 9 elements (one per level), "
                                                       f"got {len(params['block_counts'])}")):
        if not ok:
            raise ValueError(msg)
    params["block_counts"] = list(params["block_counts"])
    if not isinstance(params["exp_r"], int):
        params["exp_r"] = list(params["exp_r"])
    return _finish(MedNeXtBase(**params), head_cfg, primary_head, params["deep_supervision"])


def upkern_load_weights(target_model: MedNeXtWrapper, source_model: MedNeXtWrapper) -> MedNeXtWrapper:
    """UpKern (mednext_models.py:487-537 -> nnunet_mednext.run.load_weights): initialise a large-kernel model from a trained
    small-kernel one.  Tensors of equal shape are copied; depthwise kernels (same channels, different spatial size) are
    resized with trilinear interpolation (align_corners=False, as F.interpolate defaults); anything else is an error."""
    tgt, src = target_model.model.state_dict(), source_model.model.state_dict()
    missing = [k for k in tgt if k not in src]
    if missing:
        raise KeyError(f"UpKern: key {missing[0]} missing in the source model")
    new = {}
    for k, v in tgt.items():
        s = src[k]
        if s.shape == v.shape:
            new[k] = s.clone()
        elif s.dim() == v.dim() == 5 and s.shape[:2] == v.shape[:2]:
            new[k] = F.interpolate(s.float(), size=tuple(v.shape[2:]), mode="trilinear").to(v.dtype)
        else:
            raise ValueError(f"UpKern: incompatible shapes for {k}: {tuple(s.shape)} -> {tuple(v.shape)}")
    target_model.model.load_state_dict(new)
    return target_model


__all__ = ["MedNeXtMultiHeadWrapper", "MedNeXtTaskHead", "MedNeXtWrapper", "build_mednext",
           "build_mednext_custom", "upkern_load_weights"]
