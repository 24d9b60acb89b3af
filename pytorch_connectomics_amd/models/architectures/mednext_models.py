"""MedNeXt wrappers and registry builders -- same public surface as the reference's
connectomics/models/architectures/mednext_models.py (MedNeXtWrapper :38-89, MedNeXtTaskHead
:129-194, MedNeXtMultiHeadWrapper :197-273, build_mednext :303-397, build_mednext_custom
:400-483, upkern_load_weights :487-537), built on the in-repo MI355X trunk
(``.mednext``) instead of the external ``nnunet_mednext`` package.
"""
from __future__ import annotations

from typing import Any, Dict, Mapping, Union

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
    if isinstance(cfg, Mapping):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


def _infer_mednext_head_block_kwargs(model: nn.Module) -> dict:
    if not hasattr(model, "dec_block_0") or len(model.dec_block_0) == 0:
        raise ValueError("MedNeXt trunk must expose a non-empty dec_block_0 to build task heads.")
    ref_block = model.dec_block_0[0]
    if not isinstance(ref_block, MedNeXtBlock):
        raise TypeError("Expected MedNeXt dec_block_0 to contain MedNeXtBlock instances for multi-head reuse.")
    kernel_size = ref_block.conv1.kernel_size
    if isinstance(kernel_size, tuple):
        kernel_size = kernel_size[0]
    return {
        "exp_r": ref_block.conv2.out_channels // ref_block.conv2.in_channels,
        "kernel_size": int(kernel_size),
        "do_res": ref_block.do_res,
        "norm_type": "group" if isinstance(ref_block.norm, nn.GroupNorm) else "layer",
        "dim": ref_block.dim,
        "grn": ref_block.grn,
    }


class MedNeXtTaskHead(nn.Module):
    """Optional 1x1 in-projection -> N MedNeXt blocks -> 1x1 out-projection on the shared features."""

    def __init__(self, in_channels: int, out_channels: int, num_blocks: int, hidden_channels: int | None = None,
                 *, exp_r: int, kernel_size: int, do_res: bool, norm_type: str, dim: str, grn: bool):
        super().__init__()
        if num_blocks < 0:
            raise ValueError(f"MedNeXt task head num_blocks must be >= 0, got {num_blocks}")
        if out_channels <= 0:
            raise ValueError(f"MedNeXt task head out_channels must be positive, got {out_channels}")
        if hidden_channels is None:
            hidden_channels = in_channels
        if hidden_channels <= 0:
            raise ValueError(f"MedNeXt task head hidden_channels must be positive, got {hidden_channels}")
        if hidden_channels > in_channels:
            raise ValueError("MedNeXt task head hidden_channels must not exceed the shared feature width "
                             f"({hidden_channels} > {in_channels})")
        if dim == "2d":
            conv = nn.Conv2d
        elif dim == "3d":
            conv = nn.Conv3d
        else:
            raise ValueError(f"MedNeXt task head dim must be '2d' or '3d', got {dim}")
        self.input_projection = (conv(in_channels, hidden_channels, kernel_size=1)
                                 if hidden_channels != in_channels else nn.Identity())
        blocks = [MedNeXtBlock(hidden_channels, hidden_channels, exp_r=exp_r, kernel_size=kernel_size,
                               do_res=do_res, norm_type=norm_type, dim=dim, grn=grn) for _ in range(num_blocks)]
        self.blocks = nn.Sequential(*blocks) if blocks else nn.Identity()
        self.projection = conv(hidden_channels, out_channels, kernel_size=1)
        self.hidden_channels = hidden_channels

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
        self.feature_channels = int(self.model.stem.out_channels)
        self.head_block_kwargs = _infer_mednext_head_block_kwargs(model)
        task_heads, head_specs = {}, {}
        for head_name, head_cfg in heads.items():
            out_channels = int(_cfg_value(head_cfg, "out_channels", head_cfg))
            num_blocks = int(_cfg_value(head_cfg, "num_blocks", 0))
            hidden_channels = _cfg_value(head_cfg, "hidden_channels", None)
            hidden_channels = int(hidden_channels) if hidden_channels is not None else None
            task_heads[head_name] = MedNeXtTaskHead(self.feature_channels, out_channels, num_blocks,
                                                    hidden_channels, **self.head_block_kwargs)
            head_specs[head_name] = {"out_channels": out_channels, "num_blocks": num_blocks,
                                     "hidden_channels": hidden_channels or self.feature_channels}
        self.heads = nn.ModuleDict(task_heads)
        self.head_specs = head_specs
        resolved = primary_head or next(iter(self.heads.keys()))
        if resolved not in self.heads:
            raise ValueError(f"primary_head '{resolved}' is not one of the configured heads: "
                             f"{sorted(self.heads.keys())}")
        self.primary_head = resolved

    def forward_features(self, x: torch.Tensor) -> torch.Tensor:
        return self.model.forward_features(x)

    def forward_heads_cl(self, feat_cl: torch.Tensor) -> Dict[str, torch.Tensor]:
        hip = self.model._hip
        return {name: head.forward_cl(hip, feat_cl) for name, head in self.heads.items()}

    def forward_heads(self, features: torch.Tensor) -> Dict[str, torch.Tensor]:
        from .mednext import resolve_compute_dtype
        dt = resolve_compute_dtype(self.model.compute_dtype)
        outs = self.forward_heads_cl(to_channels_last(features).to(dt))
        return {k: to_channels_first(v) for k, v in outs.items()}

    def forward(self, x: torch.Tensor) -> Dict[str, Dict[str, torch.Tensor]]:
        self.model._check_input(x)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # training: trunk and heads through the autograd Functions of training/autograd.py (HIP forward + backward)
            from ...training.autograd import mednext_multihead_train_forward
            from .mednext import resolve_compute_dtype
            outs = mednext_multihead_train_forward(self, to_channels_last(x.float()), resolve_compute_dtype(self.model.compute_dtype))
            return {"output": {k: to_channels_first(v) for k, v in outs.items()}}
        feat_cl = self.model.features_cl(to_channels_last(x.float()))
        return {"output": {k: to_channels_first(v) for k, v in self.forward_heads_cl(feat_cl).items()}}

    def forward_cl(self, x_cl: torch.Tensor) -> torch.Tensor:
        """Channels-last fast path: all heads concatenated along C in declaration order."""
        outs = self.forward_heads_cl(self.model.features_cl(x_cl))
        return torch.cat(list(outs.values()), dim=-1) if len(outs) > 1 else next(iter(outs.values()))


def _get_mednext_heads_cfg(cfg):
    raw_heads = getattr(cfg.model, "heads", None)
    if not raw_heads:
        return {}, None
    return dict(raw_heads), getattr(cfg.model, "primary_head", None)


def _resolve_mednext_num_classes(cfg, head_cfg: Mapping[str, Any]) -> int:
    if head_cfg:
        total = sum(int(_cfg_value(spec, "out_channels", 0)) for spec in head_cfg.values())
        return max(1, total)
    return int(cfg.model.out_channels)


@register_architecture("mednext")
def build_mednext(cfg) -> ConnectomicsModel:
    """MedNeXt with a predefined size: model.mednext.size in S/B/M/L (5.6/10.5/17.6/61.8 M params at k=3),
    model.mednext.kernel_size in 3/5/7, model.loss.deep_supervision, optional model.heads."""
    in_channels = cfg.model.in_channels
    model_size = getattr(cfg.model.mednext, "size", "S")
    kernel_size = getattr(cfg.model.mednext, "kernel_size", 3)
    loss_cfg = getattr(cfg.model, "loss", None)
    deep_supervision = getattr(loss_cfg, "deep_supervision", False)
    head_cfg, primary_head = _get_mednext_heads_cfg(cfg)
    out_channels = _resolve_mednext_num_classes(cfg, head_cfg)
    if model_size not in ["S", "B", "M", "L"]:
        raise ValueError(f"MedNeXt model_size must be 'S', 'B', 'M', or 'L'. Got: {model_size}\n"
                         "Model sizes:\n  - S (Small): 5.6M params\n  - B (Base): 10.5M params\n"
                         "  - M (Medium): 17.6M params\n  - L (Large): 61.8M params")
    if kernel_size not in [3, 5, 7]:
        raise ValueError(f"MedNeXt kernel_size must be 3, 5, or 7. Got: {kernel_size}\n"
                         "Recommended: Start with kernel_size=3")
    model = create_mednext_v1(num_input_channels=in_channels, num_classes=out_channels, model_id=model_size,
                              kernel_size=kernel_size, deep_supervision=deep_supervision)
    checkpoint_style = getattr(cfg.model.mednext, "checkpoint_style", None)
    if checkpoint_style is not None:
        if checkpoint_style != "outside_block":
            raise ValueError("model.mednext.checkpoint_style must be None or 'outside_block', "
                             f"got: {checkpoint_style!r}")
        model.outside_block_checkpointing = True
    if head_cfg:
        return MedNeXtMultiHeadWrapper(model, head_cfg, primary_head=primary_head)
    return MedNeXtWrapper(model, deep_supervision=deep_supervision)


@register_architecture("mednext_custom")
def build_mednext_custom(cfg) -> ConnectomicsModel:
    """MedNeXt with explicit base_channels / exp_r / kernel_size / block_counts / norm / dim / grn."""
    head_cfg, primary_head = _get_mednext_heads_cfg(cfg)
    params = {
        "in_channels": cfg.model.in_channels,
        "n_channels": getattr(cfg.model.mednext, "base_channels", 32),
        "n_classes": _resolve_mednext_num_classes(cfg, head_cfg),
        "exp_r": getattr(cfg.model.mednext, "exp_r", 4),
        "kernel_size": getattr(cfg.model.mednext, "kernel_size", 7),
        "deep_supervision": getattr(cfg.model.loss, "deep_supervision", False),
        "do_res": getattr(cfg.model.mednext, "do_res", True),
        "do_res_up_down": getattr(cfg.model.mednext, "do_res_up_down", True),
        "block_counts": getattr(cfg.model.mednext, "block_counts", [2] * 9),
        "checkpoint_style": getattr(cfg.model.mednext, "checkpoint_style", None),
        "norm_type": getattr(cfg.model.mednext, "norm", "group"),
        "dim": getattr(cfg.model.mednext, "dim", "3d"),
        "grn": getattr(cfg.model.mednext, "grn", False),
    }
    if params["dim"] not in ["2d", "3d"]:
        raise ValueError(f"mednext_dim must be '2d' or '3d', got: {params['dim']}")
    if params["norm_type"] not in ["group", "layer"]:
        raise ValueError(f"mednext_norm must be 'group' or 'layer', got: {params['norm_type']}")
    if len(params["block_counts"]) != 9:
        raise ValueError("mednext_block_counts must have exactly 9 elements (one per level), "
                         f"got {len(params['block_counts'])}")
    params["block_counts"] = list(params["block_counts"])
    if not isinstance(params["exp_r"], int):
        params["exp_r"] = list(params["exp_r"])
    model = MedNeXtBase(**params)
    if head_cfg:
        return MedNeXtMultiHeadWrapper(model, head_cfg, primary_head=primary_head)
    return MedNeXtWrapper(model, deep_supervision=params["deep_supervision"])


def upkern_load_weights(target_model: MedNeXtWrapper, source_model: MedNeXtWrapper) -> MedNeXtWrapper:
    """UpKern: initialise a large-kernel model from a trained small-kernel one -- every tensor is
    copied, depthwise kernels of differing size are trilinearly resized (one-off, load time)."""
    tgt, src = target_model.model.state_dict(), source_model.model.state_dict()
    new = {}
    for k, v in tgt.items():
        if k not in src:
            raise KeyError(f"UpKern: key {k} missing in the source model")
        s = src[k]
        if s.shape == v.shape:
            new[k] = s.clone()
        elif s.dim() == 5 and s.shape[:2] == v.shape[:2]:
            new[k] = F.interpolate(s.float(), size=tuple(v.shape[2:]), mode="trilinear").to(v.dtype)
        else:
            raise ValueError(f"UpKern: incompatible shapes for {k}: {tuple(s.shape)} -> {tuple(v.shape)}")
    target_model.model.load_state_dict(new)
    return target_model


__all__ = ["MedNeXtMultiHeadWrapper", "MedNeXtTaskHead", "MedNeXtWrapper", "build_mednext",
           "build_mednext_custom", "upkern_load_weights"]
