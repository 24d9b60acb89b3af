"""Lightning-free counterpart of the reference's ConnectomicsModule (training/lightning/model.py:74-1256) for
the hot path: forward / training_step / validation_step / configure_optimizers with the same method names,
a two-term loss (weighted BCE-with-logits + sigmoid Dice, profiles/loss_profiles.yaml:2-9) with deep-supervision
weights [1, .5, .25, .125, .0625] (schema/model.py:46-51), AdamW with the no-weight-decay-on-norm grouping
(training/optimization/build.py:73-112), WarmupCosineLR (lr_scheduler.py:48-82), global-norm gradient clipping,
DDP over RCCL, and Lightning-layout checkpoints ("state_dict" with the "model." prefix, model.py:244-297).

The network forward/backward run the HIP kernels; on a GPU the BCE + Dice loss and the clip + AdamW (+ EMA) update
are HIP kernels too (training/fused.py); other loss functions and optimizers are PyTorch
device ops.
"""
from __future__ import annotations

import math
from typing import Any, Dict, Mapping, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..models import build_model
from ..utils.channel_slices import resolve_channel_indices
from ..utils.model_outputs import resolve_head_target_slice, unwrap_main_output

_NORM_TYPES = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d, nn.SyncBatchNorm, nn.GroupNorm, nn.InstanceNorm1d,
               nn.InstanceNorm2d, nn.InstanceNorm3d, nn.LayerNorm, nn.LocalResponseNorm)
DS_WEIGHTS = [1.0, 0.5, 0.25, 0.125, 0.0625]


def dice_loss_sigmoid(logits: torch.Tensor, target: torch.Tensor, smooth_nr: float = 1e-5, smooth_dr: float = 1e-5):
    """MONAI DiceLoss(sigmoid=True) semantics: per (B, C) dice over spatial dims, mean over B and C."""
    p = torch.sigmoid(logits.float())
    t = target.float()
    dims = tuple(range(2, p.dim()))
    inter = (p * t).sum(dims)
    den = p.sum(dims) + t.sum(dims)
    return (1.0 - (2.0 * inter + smooth_nr) / (den + smooth_dr)).mean()


def _mask_for_unweighted_loss(pred, target, mask, clamp_min: float):
    """Losses without a spatial-weight argument (the MONAI ones) see a mask through their INPUTS
    (training/losses/orchestrator.py:648-655): invalid voxels get the clamp floor as logit and 0 as target."""
    if mask is None:
        return pred, target
    valid = (mask > 0).expand_as(pred)
    return pred.masked_fill(~valid, float(clamp_min)), target * (mask > 0).to(target.dtype)


def _drop_background(p, t, include_background: bool):
    if include_background or p.shape[1] == 1:          # MONAI: "single channel prediction, include_background=False ignored"
        return p, t
    return p[:, 1:], t[:, 1:]


def monai_dice_loss(logits, target, *, sigmoid: bool = False, include_background: bool = True, smooth_nr=1e-5, smooth_dr=1e-5,
                    squared_pred: bool = False, **_unused):
    """monai.losses.DiceLoss as the reference instantiates it (models/losses/build.py:58-61; reduction 'mean', batch=False):
    1 - (2 sum(p t) + smooth_nr) / (sum(p) + sum(t) + smooth_dr) per (B, C) over the spatial dims, mean over B and C.
    `sigmoid` defaults to False as in MONAI (a config that omits it trains Dice on the raw logits).  Parity unpinned: MONAI
    is not in the image; restated from its documented formula."""
    p = torch.sigmoid(logits.float()) if sigmoid else logits.float()
    t = target.float()
    p, t = _drop_background(p, t, bool(include_background))
    dims = tuple(range(2, p.dim()))
    inter = (p * t).sum(dims)
    den = ((p * p).sum(dims) + (t * t).sum(dims)) if squared_pred else (p.sum(dims) + t.sum(dims))
    return (1.0 - (2.0 * inter + float(smooth_nr)) / (den + float(smooth_dr))).mean()


def monai_tversky_loss(logits, target, *, sigmoid: bool = False, include_background: bool = True, alpha: float = 0.5,
                       beta: float = 0.5, smooth_nr=1e-5, smooth_dr=1e-5, **_unused):
    """monai.losses.TverskyLoss: tp = sum(p t), fp = alpha sum(p (1 - t)), fn = beta sum((1 - p) t) per (B, C);
    1 - (tp + smooth_nr) / (tp + fp + fn + smooth_dr), mean.  Parity unpinned (see monai_dice_loss)."""
    p = torch.sigmoid(logits.float()) if sigmoid else logits.float()
    t = target.float()
    p, t = _drop_background(p, t, bool(include_background))
    dims = tuple(range(2, p.dim()))
    tp = (p * t).sum(dims)
    fp = float(alpha) * (p * (1.0 - t)).sum(dims)
    fn = float(beta) * ((1.0 - p) * t).sum(dims)
    return (1.0 - (tp + float(smooth_nr)) / (tp + fp + fn + float(smooth_dr))).mean()


def monai_focal_loss(logits, target, *, gamma: float = 2.0, alpha=None, include_background: bool = True, **_unused):
    """monai.losses.FocalLoss (sigmoid form, use_softmax=False, reduction 'mean'): BCE-with-logits * (1 - p_t)^gamma, with
    alpha: * (alpha t + (1 - alpha)(1 - t)); mean over every element.  Parity unpinned (see monai_dice_loss)."""
    x, t = logits.float(), target.float()
    x, t = _drop_background(x, t, bool(include_background))
    bce = F.binary_cross_entropy_with_logits(x, t, reduction="none")
    # (1 - p_t)^gamma = exp(gamma * logsigmoid(-x (2t - 1))): stable for large |x|
    mod = torch.exp(float(gamma) * F.logsigmoid(-x * (2.0 * t - 1.0)))
    loss = mod * bce
    if alpha is not None:
        loss = loss * (float(alpha) * t + (1.0 - float(alpha)) * (1.0 - t))
    return loss.mean()


def auto_pos_weight_scalar(target, mask=None, cap: float = 10.0) -> torch.Tensor:
    """`pos_weight: auto` of a weighted-BCE term (orchestrator.py:180-197): min(neg / pos, 10) over the valid voxels, 1 when
    either class is absent -- computed on the device, no host synchronisation."""
    valid = torch.ones_like(target, dtype=torch.bool) if mask is None else (mask > 0).expand_as(target)
    pos = ((target > 0) & valid).sum().float()
    neg = ((target <= 0) & valid).sum().float()
    ratio = torch.clamp(neg / pos.clamp_min(1.0), max=float(cap))
    return torch.where((pos > 0) & (neg > 0), ratio, torch.ones_like(ratio)).reshape(1)


def class_balance_weight(target: torch.Tensor, pos_weight=None, valid_mask=None, cap: float = 10.0) -> torch.Tensor:
    """The spatial weight map the reference's orchestrator hands to every `weight`-taking loss except the weighted BCE when a term
    carries no mask of its own (orchestrator.py:129-178, :606-612).  `pos_weight` None / "auto": foreground (target > 0) and
    background weights in the ratio neg : pos, scaled so that the valid voxels average 1, each capped at `cap`; ones when a class
    is absent.  A number: that weight on the foreground, 1 elsewhere.  Counts stay on the device (no host synchronisation)."""
    t = target
    ones = torch.ones_like(t)
    if pos_weight is not None and not isinstance(pos_weight, str):
        fg = float(pos_weight)
        if fg <= 0:
            raise ValueError(f"pos_weight must be > 0, got {fg}")
        return torch.where(t > 0, ones * fg, ones)
    if isinstance(pos_weight, str) and pos_weight != "auto":
        raise ValueError(f"Unsupported pos_weight mode: {pos_weight!r}. Expected a positive number or 'auto'.")
    valid = torch.ones_like(t, dtype=torch.bool) if valid_mask is None else (valid_mask > 0).expand_as(t)
    pos = ((t > 0) & valid).sum().to(t.dtype if t.is_floating_point() else torch.float32)
    neg = ((t <= 0) & valid).sum().to(pos.dtype)
    both = (pos > 0) & (neg > 0)
    count = pos + neg
    fg = torch.clamp(count / (2.0 * pos.clamp_min(1.0)), max=float(cap))
    bg = torch.clamp(count / (2.0 * neg.clamp_min(1.0)), max=float(cap))
    balanced = torch.where(t > 0, ones * fg, ones * bg)
    return torch.where(both, balanced, ones)


# losses whose spatial argument is a WEIGHT map (reference models/losses/metadata.py:38-48); every other loss sees a mask as
# "logits at the clamp minimum, target 0" outside it
_WEIGHT_TAKING = {"SmoothL1Loss", "WeightedBCEWithLogitsLoss", "PerChannelBCEWithLogitsLoss", "WeightedMSELoss", "WeightedMAELoss"}
_OWN_CLASS_BALANCE = {"WeightedBCEWithLogitsLoss", "PerChannelBCEWithLogitsLoss"}      # pos_weight defaults to 1 there (plan.py:105-112)


def per_channel_bce_with_logits(logits, target, weight=None, *, auto_pos_weight: bool = True, max_pos_weight: float = 10.0,
                                reduction: str = "mean"):
    """models/losses/losses.py:269-351: BCE per channel with its own class-balancing weight min(n_neg / n_pos, max_pos_weight)
    (1 for a channel without positives) from the valid voxels of the current batch, reduced per channel (weighted-valid mean or
    sum), summed over the channels."""
    x, t = logits.float(), target.float()
    C = x.shape[1]
    pw = None
    if auto_pos_weight:
        valid = (weight > 0).expand_as(t) if weight is not None else torch.ones_like(t, dtype=torch.bool)
        dims = (0,) + tuple(range(2, t.dim()))
        pos = ((t > 0) & valid).sum(dims).float()
        neg = ((t <= 0) & valid).sum(dims).float()
        ratio = torch.clamp(neg / pos.clamp_min(1.0), max=float(max_pos_weight))
        pw = torch.where(pos > 0, ratio, torch.ones_like(ratio)).reshape(1, C, *([1] * (t.dim() - 2)))
    bce = F.binary_cross_entropy_with_logits(x, t, pos_weight=pw, reduction="none")
    total = x.new_zeros(())
    w = None if weight is None else weight.float().expand_as(bce)
    for c in range(C):
        b = bce[:, c:c + 1]
        if w is None:
            total = total + (b.mean() if reduction == "mean" else b.sum())
        else:
            wc = w[:, c:c + 1]
            bw = b * wc
            valid_c = wc > 0
            total = total + ((bw * valid_c).sum() / valid_c.sum().clamp_min(1) if reduction == "mean" else bw.sum())
    return total


def weighted_bce_with_logits(logits, target, weight=None, pos_weight=None):
    """models/losses/losses.py:17-44,190-266 (reduction='mean'; with a weight map: the mean of weight * bce over the
    voxels whose weight is > 0, the map broadcast to the logits' shape; 0 when no voxel is valid)."""
    if pos_weight is None:
        pw = None
    elif isinstance(pos_weight, str):
        if pos_weight != "auto":
            raise ValueError(f"Unsupported pos_weight mode: {pos_weight!r}. Expected a positive number or 'auto'.")
        pw = auto_pos_weight_scalar(target, weight).to(logits.device)
    elif isinstance(pos_weight, torch.Tensor):
        pw = pos_weight.to(device=logits.device, dtype=torch.float32)
    else:
        if float(pos_weight) <= 0:
            raise ValueError(f"pos_weight must be > 0, got {float(pos_weight)}")
        pw = torch.as_tensor([float(pos_weight)], device=logits.device, dtype=torch.float32)
    bce = F.binary_cross_entropy_with_logits(logits.float(), target.float(), pos_weight=pw, reduction="none")
    if weight is None:
        return bce.mean()
    w = weight.float().expand_as(bce)
    valid = w > 0
    return (bce * w * valid).sum() / valid.sum().clamp_min(1)


def _reduce_weighted(loss: torch.Tensor, weight: Optional[torch.Tensor]) -> torch.Tensor:
    """losses.py:17-44 with reduction='mean': plain mean, or the mean of weight * loss over the voxels with weight > 0."""
    if weight is None:
        return loss.mean()
    w = weight.float().expand_as(loss)
    valid = w > 0
    return (loss * w * valid).sum() / valid.sum().clamp_min(1)


def weighted_regression_loss(kind: str, pred, target, weight=None, *, tanh: bool = False, beta: float = 1.0):
    """WeightedMSELoss / WeightedMAELoss / SmoothL1Loss of the reference (losses.py:140-187, 725-800): optional tanh on
    the prediction (distance-transform targets in [-1, 1]), elementwise loss, weighted-valid mean."""
    p = pred.float()
    if tanh:
        p = torch.tanh(p)
    t = target.float()
    if kind == "mse":
        loss = (p - t) ** 2
    elif kind == "mae":
        loss = (p - t).abs()
    else:
        loss = F.smooth_l1_loss(p, t, beta=float(beta), reduction="none")
    return _reduce_weighted(loss, weight)


_LOSSES = {
    "WeightedMSELoss": lambda p, t, **kw: weighted_regression_loss("mse", p, t, kw.get("weight"), tanh=bool(kw.get("tanh", False))),
    "WeightedMAELoss": lambda p, t, **kw: weighted_regression_loss("mae", p, t, kw.get("weight"), tanh=bool(kw.get("tanh", False))),
    "SmoothL1Loss": lambda p, t, **kw: weighted_regression_loss("huber", p, t, kw.get("weight"), tanh=bool(kw.get("tanh", False)),
                                                                beta=float(kw.get("beta", 1.0))),
    "DiceLoss": lambda p, t, **kw: monai_dice_loss(*_mask_for_unweighted_loss(p, t, kw.get("weight"), kw.get("clamp_min", -20.0)),
                                                   **{k: v for k, v in kw.items() if k not in ("weight", "pos_weight", "clamp_min")}),
    "TverskyLoss": lambda p, t, **kw: monai_tversky_loss(*_mask_for_unweighted_loss(p, t, kw.get("weight"), kw.get("clamp_min", -20.0)),
                                                         **{k: v for k, v in kw.items() if k not in ("weight", "pos_weight", "clamp_min")}),
    "FocalLoss": lambda p, t, **kw: monai_focal_loss(*_mask_for_unweighted_loss(p, t, kw.get("weight"), kw.get("clamp_min", -20.0)),
                                                     **{k: v for k, v in kw.items() if k not in ("weight", "pos_weight", "clamp_min")}),
    "PerChannelBCEWithLogitsLoss": lambda p, t, **kw: per_channel_bce_with_logits(
        p, t, kw.get("weight"), auto_pos_weight=bool(kw.get("auto_pos_weight", True)),
        max_pos_weight=float(kw.get("max_pos_weight", 10.0)), reduction=str(kw.get("reduction", "mean"))),
    "WeightedBCEWithLogitsLoss": lambda p, t, **kw: weighted_bce_with_logits(p, t, kw.get("weight"), kw.get("pos_weight")),
    # torch's own losses take no spatial argument: a mask reaches them as "logit at the clamp minimum, target 0" outside it
    "BCEWithLogitsLoss": lambda p, t, **kw: F.binary_cross_entropy_with_logits(
        *(v.float() for v in _mask_for_unweighted_loss(p, t, kw.get("weight"), kw.get("clamp_min", -20.0)))),
    "MSELoss": lambda p, t, **kw: F.mse_loss(*(v.float() for v in _mask_for_unweighted_loss(p, t, kw.get("weight"), kw.get("clamp_min", -20.0)))),
}


def match_target_to_output(target: torch.Tensor, output: torch.Tensor) -> torch.Tensor:
    """Deep-supervision target at an output's resolution (training/losses/orchestrator.py:892-952): integer labels by nearest
    neighbour (kept integer), dense float targets by trilinear interpolation (align_corners=False) and clamped back to
    [-1, 1] / [0, 1] when the original lies in that range (interpolation overshoot of tanh-SDT / sigmoid targets)."""
    if target.shape == output.shape:
        return target
    if target.dtype in (torch.long, torch.int, torch.int32, torch.int64, torch.uint8):
        return F.interpolate(target.float(), size=output.shape[2:], mode="nearest").long()
    # (B, C, H, W) targets of a dim='2d' model: the bilinear twin (the reference names "trilinear" unconditionally, which
    # torch rejects for 4-D tensors, so its 2-D + deep-supervision configs cannot reach this line)
    out = F.interpolate(target, size=output.shape[2:], mode="trilinear" if target.dim() == 5 else "bilinear", align_corners=False)
    lo, hi = float(target.min()), float(target.max())
    if lo >= -1.5 and hi <= 1.5:
        out = torch.clamp(out, -1.0, 1.0)
    elif lo >= 0.0 and hi <= 1.5:
        out = torch.clamp(out, 0.0, 1.0)
    return out


def resize_class_index_to_output(t: torch.Tensor, output: torch.Tensor) -> torch.Tensor:
    """Nearest-neighbour resize regardless of dtype (orchestrator.py:277-291): batch masks and class-index targets."""
    if t.shape[2:] == output.shape[2:]:
        return t
    r = F.interpolate(t.float(), size=output.shape[2:], mode="nearest")
    return r.to(t.dtype) if t.dtype.is_floating_point else r.long()


class WarmupCosineLR(torch.optim.lr_scheduler.LRScheduler):
    """lr = eta_min + (base * warmup(t) - eta_min) * 0.5 (1 + cos(pi t / max_iters))."""

    def __init__(self, optimizer, max_iters: int, warmup_factor: float = 0.001, warmup_iters: int = 1000,
                 eta_min: float = 0.0, last_epoch: int = -1):
        self.max_iters, self.warmup_factor, self.warmup_iters, self.eta_min = max_iters, warmup_factor, warmup_iters, eta_min
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        it = self.last_epoch
        wf = 1.0
        if it < self.warmup_iters:
            alpha = it / max(1, self.warmup_iters)
            wf = self.warmup_factor * (1 - alpha) + alpha
        cos = 0.5 * (1.0 + math.cos(math.pi * it / self.max_iters))
        return [self.eta_min + (b * wf - self.eta_min) * cos for b in self.base_lrs]


def build_optimizer(cfg, model: nn.Module) -> torch.optim.Optimizer:
    oc = cfg.optimization.optimizer
    name = str(getattr(oc, "name", "adamw")).lower()
    lr = float(getattr(oc, "lr", 1e-4))
    wd = float(getattr(oc, "weight_decay", 1e-4))
    wd_norm = float(getattr(oc, "weight_decay_norm", 0.0))
    wd_bias = float(getattr(oc, "weight_decay_bias", wd))
    bias_lr = float(getattr(oc, "bias_lr_factor", 1.0))
    # the reference (optimization/build.py:73-112) emits one group per parameter; parameters with identical
    # hyper-parameters are merged here (same update rule, but the multi-tensor optimizer then runs a handful of fused
    # launches per step instead of ~8 per parameter -- 1 800 tiny launches / 40 ms of host time for MedNeXt-S)
    merged, seen, seen_order = {}, set(), []
    for module in model.modules():
        for key, p in module.named_parameters(recurse=False):
            if not p.requires_grad or p in seen:
                continue
            seen.add(p)
            seen_order.append(p)
            g_lr, g_wd = lr, wd
            if isinstance(module, _NORM_TYPES):
                g_wd = wd_norm
            elif key == "bias":
                g_lr, g_wd = lr * bias_lr, wd_bias
            merged.setdefault((g_lr, g_wd), []).append(p)
    groups = [{"params": ps, "lr": k[0], "weight_decay": k[1]} for k, ps in merged.items()]
    model_order = {id(p): i for i, p in enumerate(p for p in seen_order)}
    betas = tuple(getattr(oc, "betas", (0.9, 0.999)))
    eps = float(getattr(oc, "eps", 1e-8))
    if name == "adamw":
        fused = bool(getattr(oc, "fused", True)) and all(p.is_cuda for g in groups for p in g["params"])
        if fused:
            # one multi-tensor HIP launch for the clip norm and one for the update (training/fused.py); the clip value
            # of optimization.gradient_clip_val is applied inside, so fit() skips torch's clip_grad_norm_
            from .fused import FusedAdamW
            ema = getattr(cfg.optimization, "ema", None)
            ema_on = bool(getattr(ema, "enabled", False)) if ema is not None else False
            opt = FusedAdamW(groups, lr=lr, betas=betas, eps=eps, weight_decay=wd,
                             max_grad_norm=float(getattr(cfg.optimization, "gradient_clip_val", 0.0) or 0.0),
                             ema_decay=float(getattr(ema, "decay", 0.999)) if ema_on else None,
                             ema_warmup_steps=int(getattr(ema, "warmup_steps", 0) or 0) if ema_on else 0)
        else:
            opt = torch.optim.AdamW(groups, lr=lr, betas=betas, eps=eps, weight_decay=wd)
    elif name == "adam":
        opt = torch.optim.Adam(groups, lr=lr, betas=betas, eps=eps)
    elif name == "sgd":
        opt = torch.optim.SGD(groups, lr=lr, momentum=float(getattr(oc, "momentum", 0.9)), weight_decay=wd,
                              nesterov=bool(getattr(oc, "nesterov", False)))
    else:
        raise ValueError(f"Unknown optimizer: '{name}'. Supported optimizers: adamw, adam, sgd")
    opt._pytc_model_order = model_order          # position of every parameter in registration order (checkpoint remapping)
    return opt


def _sched_param(sc, key, default, *, specific=False):
    """optimization/build.py:30-44: `scheduler.params[key]` first; then (unless scheduler-specific) the direct field."""
    params = getattr(sc, "params", None)
    if isinstance(params, Mapping) and key in params and params[key] is not None:
        return params[key]
    if specific:
        return default
    v = getattr(sc, key, default)
    return default if v is None else v


def build_lr_scheduler(cfg, optimizer):
    """The reference's scheduler table (optimization/build.py:155-324): cosineannealinglr (default, also for unknown names),
    cosineannealingwarmrestarts, steplr, multisteplr, reducelronplateau, warmupcosine[lr], constant[lr]; parameters from
    `scheduler.params` then the typed fields; `name: null` = no scheduler.  WarmupCosineLR also accepts the round-1 spellings
    (max_iters / warmup_iters / warmup_factor)."""
    from torch.optim import lr_scheduler as LS
    sc = getattr(cfg.optimization, "scheduler", None)
    if sc is None or getattr(sc, "name", "cosineannealinglr") is None:
        return None
    name = str(_sched_param(sc, "name", "cosineannealinglr")).lower()
    max_epochs = int(getattr(cfg.optimization, "max_epochs", 100) or 100)
    if name in ("cosineannealingwarmrestarts", "cosinewarmrestarts"):
        return LS.CosineAnnealingWarmRestarts(optimizer, T_0=int(_sched_param(sc, "T_0", 200, specific=True)),
                                              T_mult=int(_sched_param(sc, "T_mult", 1, specific=True)),
                                              eta_min=float(_sched_param(sc, "min_lr", 1e-5)))
    if name == "steplr":
        return LS.StepLR(optimizer, step_size=int(_sched_param(sc, "step_size", 30, specific=True)),
                         gamma=float(_sched_param(sc, "gamma", 0.1, specific=True)))
    if name == "multisteplr":
        return LS.MultiStepLR(optimizer, milestones=list(_sched_param(sc, "milestones", [30, 60, 90], specific=True)),
                              gamma=float(_sched_param(sc, "gamma", 0.1, specific=True)))
    if name == "reducelronplateau":
        return LS.ReduceLROnPlateau(optimizer, mode=_sched_param(sc, "mode", "min"), factor=float(_sched_param(sc, "factor", 0.1)),
                                    patience=int(_sched_param(sc, "patience", 10)), threshold=float(_sched_param(sc, "threshold", 1e-4)),
                                    cooldown=int(_sched_param(sc, "cooldown", 0)), eps=float(_sched_param(sc, "eps", 1e-8)),
                                    min_lr=float(_sched_param(sc, "min_lr", 1e-6)))
    if name in ("warmupcosine", "warmupcosinelr", "warmup_cosine", "warmup_cosine_lr"):
        max_iter = _sched_param(sc, "max_iter", None, specific=True) or getattr(sc, "max_iters", None) or max_epochs
        warm = getattr(sc, "warmup_iters", None)
        factor = getattr(sc, "warmup_factor", None)
        return WarmupCosineLR(optimizer, int(max_iter),
                              warmup_factor=float(factor if factor is not None else _sched_param(sc, "warmup_start_lr", 0.001)),
                              warmup_iters=int(warm if warm is not None else _sched_param(sc, "warmup_epochs", 5)),
                              eta_min=float(_sched_param(sc, "min_lr", 0.0)))
    if name in ("constant", "constantlr"):
        return LS.LambdaLR(optimizer, lr_lambda=lambda epoch: 1.0)
    if name != "cosineannealinglr":
        import logging
        logging.getLogger(__name__).warning("unknown scheduler %r: using CosineAnnealingLR like the reference builder", name)
    return LS.CosineAnnealingLR(optimizer, T_max=int(_sched_param(sc, "t_max", max_epochs, specific=True)),
                                eta_min=float(_sched_param(sc, "min_lr", 1e-6)))


class ConnectomicsModule(nn.Module):
    def __init__(self, cfg, model: Optional[nn.Module] = None):
        super().__init__()
        self.cfg = cfg
        self.model = model if model is not None else build_model(cfg)
        loss_cfg = getattr(cfg.model, "loss", None)
        self.deep_supervision = bool(getattr(loss_cfg, "deep_supervision", False))
        self.ds_weights = list(getattr(loss_cfg, "deep_supervision_weights", None) or DS_WEIGHTS)
        terms = getattr(loss_cfg, "losses", None)
        if not terms:
            terms = [{"function": "WeightedBCEWithLogitsLoss", "weight": 1.0},
                     {"function": "DiceLoss", "weight": 1.0, "kwargs": {"sigmoid": True}}]
        self.loss_terms = []
        for t in terms:
            get = (lambda k, d=None, _t=t: _t.get(k, d)) if isinstance(t, dict) else (lambda k, d=None, _t=t: getattr(_t, k, d))
            fn = get("function")
            if fn not in _LOSSES:
                raise ValueError(f"Unknown loss function {fn!r}; available: {sorted(_LOSSES)}")
            pos_weight = get("pos_weight")
            if isinstance(pos_weight, str):
                if pos_weight.strip().lower() != "auto":
                    raise ValueError(f"losses[{len(self.loss_terms)}] pos_weight must be a positive number or 'auto', got {pos_weight!r}")
                pos_weight = "auto"
            elif pos_weight is not None and float(pos_weight) <= 0:
                raise ValueError(f"losses[{len(self.loss_terms)}] pos_weight must be > 0, got {float(pos_weight)}")
            if pos_weight is not None and fn not in _WEIGHT_TAKING:
                raise ValueError(f"losses[{len(self.loss_terms)}] pos_weight is only supported for losses with "
                                 f"spatial_weight_arg='weight' (got {fn})")
            # which head a term reads is settled when the module is built (training/losses/plan.py:183-229): pred_head, else
            # model.primary_head, else the only head; a term that names no target channels takes that head's `target_slice`
            heads = getattr(cfg.model, "heads", None)
            heads = heads if isinstance(heads, Mapping) else {}
            where, pred_head, primary = f"losses[{len(self.loss_terms)}]", get("pred_head"), getattr(cfg.model, "primary_head", None)
            if heads:
                if pred_head is not None and pred_head not in heads:
                    raise ValueError(f"{where} pred_head={pred_head!r} is not one of the configured model.heads {sorted(heads)}")
                if pred_head is None and primary is None and len(heads) > 1:
                    raise ValueError(f"{where} must define pred_head or model.primary_head when model.heads has multiple entries "
                                     f"{sorted(heads)}")
            elif pred_head is not None:
                raise ValueError(f"{where} uses pred_head/pred2_head but model.heads is not configured.")
            target_slice = get("target_slice", get("target"))
            if target_slice is None:
                head = pred_head or primary or (next(iter(heads)) if len(heads) == 1 else None)
                if head is not None:
                    target_slice = resolve_head_target_slice(cfg, head)
            # the reference's spellings (training/losses/plan.py:126-147): coefficient = weight, pred / target / mask = *_slice
            self.loss_terms.append({"fn": fn, "weight": float(get("coefficient", get("weight", 1.0))), "pred_head": get("pred_head"),
                                    "pred_slice": get("pred_slice", get("pred")),
                                    "target_slice": target_slice, "pos_weight": pos_weight,
                                    "mask_slice": get("mask_slice", get("mask")),
                                    "apply_deep_supervision": bool(get("apply_deep_supervision", True)),
                                    "kwargs": dict(get("kwargs", None) or {})})
        # adaptive loss balancing (reference training/losses/balancing.py:64-222; tutorials/mitoEM/common.yaml:54-55): a trainable
        # sub-module, every loss entry is one task.  `uncertainty` rides on the fused loss kernel (the learned task coefficients
        # enter it as host scalars, _balanced_scale_loss); `gradnorm` needs every task's own graph and takes the generic path
        from .balancing import build_loss_weighter
        self.loss_weighter = build_loss_weighter(cfg, len(self.loss_terms), self.model)
        self.fused_loss = bool(getattr(loss_cfg, "fused", True))
        # every prediction is clamped before its loss, at every scale (orchestrator.py:95-96,574; schema/model.py:50-51)
        self.clamp_min = float(getattr(loss_cfg, "deep_supervision_clamp_min", -20.0))
        self.clamp_max = float(getattr(loss_cfg, "deep_supervision_clamp_max", 20.0))
        self.global_step = 0
        self.current_epoch = 0

    # ---- reference-visible methods ----------------------------------------------------------------
    def forward(self, x: torch.Tensor):
        return self.model(x)

    _FUSABLE = {"WeightedBCEWithLogitsLoss": "bce", "BCEWithLogitsLoss": "bce", "DiceLoss": "dice"}

    def _term_is_fusable(self, t, pred) -> bool:
        """The fused kernel computes mean-reduced BCE-with-logits (numeric pos_weight) and sigmoid Dice with MONAI's default
        smoothing over all channels; any other variant of those terms takes the generic path."""
        if t["fn"] not in self._FUSABLE or t.get("mask_slice") is not None:
            return False
        kw = t["kwargs"]
        if self._FUSABLE[t["fn"]] == "bce":
            return not isinstance(t["pos_weight"], str) and str(kw.get("reduction", "mean")) == "mean" and \
                not (set(kw) - {"reduction"})
        smooth_ok = abs(float(kw.get("smooth_nr", 1e-5)) - 1e-5) < 1e-12 and abs(float(kw.get("smooth_dr", 1e-5)) - 1e-5) < 1e-12
        bg_ok = bool(kw.get("include_background", True)) or pred.shape[1] == 1
        return bool(kw.get("sigmoid", False)) and smooth_ok and bg_ok and not kw.get("squared_pred", False) and \
            not (set(kw) - {"sigmoid", "smooth_nr", "smooth_dr", "include_background", "squared_pred"})

    def _fused_term_loss(self, pred, target, mask, terms):
        """All terms are BCE-with-logits / sigmoid-Dice: one fused HIP reduction per (pred_slice, target_slice) pair."""
        from .fused import bce_dice_loss
        groups = {}
        for i, t in terms:
            key = (str(t["pred_slice"]), str(t["target_slice"]))
            g = groups.setdefault(key, {"bce": None, "dice": None, "t": t})
            kind = self._FUSABLE[t["fn"]]
            if g[kind] is not None:
                return None                   # two terms of one kind on the same slices: take the generic path
            g[kind] = (i, t)
        total, parts = 0.0, {}
        for g in groups.values():
            t = g["t"]
            p, y = pred, target
            if t["pred_slice"] is not None:
                p = pred[:, resolve_channel_indices(t["pred_slice"], num_channels=pred.shape[1], context="pred_slice")]
            if t["target_slice"] is not None:
                y = target[:, resolve_channel_indices(t["target_slice"], num_channels=target.shape[1], context="target_slice")]
            wb = g["bce"][1]["weight"] if g["bce"] else 0.0
            wd = g["dice"][1]["weight"] if g["dice"] else 0.0
            pw = g["bce"][1]["pos_weight"] if g["bce"] and g["bce"][1]["fn"] == "WeightedBCEWithLogitsLoss" else None
            v, out = bce_dice_loss(p, y, mask, w_bce=wb, w_dice=wd, pos_weight=pw)
            if g["bce"]:
                parts[f"loss_{g['bce'][0]}_{g['bce'][1]['fn']}"] = out[1]
            if g["dice"]:
                parts[f"loss_{g['dice'][0]}_{g['dice'][1]['fn']}"] = out[2]
            total = total + v
        return total, parts

    def _fusable(self, pred, mask, terms) -> bool:
        # (torch's BCEWithLogitsLoss under a mask averages over ALL voxels -- masked ones as logit -20 / target 0 --, which the fused
        # kernel's valid-voxel mean is not)
        plain_bce_masked = mask is not None and any(t["fn"] == "BCEWithLogitsLoss" for _, t in terms)
        return bool(pred.is_cuda and self.fused_loss and not plain_bce_masked and all(self._term_is_fusable(t, pred) for _, t in terms))

    def _term_loss(self, pred, target, mask=None, terms=None, tasks=None):
        """Weighted sum of the loss terms `terms` (list of (index, term); default: all) on one prediction tensor.
        tasks: a dict that receives {term index: static weight x raw value, WITH its graph} (adaptive balancing: the caller
        combines the tasks); the fused kernel steps aside then -- it returns one scalar for all its terms."""
        terms = list(enumerate(self.loss_terms)) if terms is None else terms
        pred = torch.clamp(pred, min=self.clamp_min, max=self.clamp_max)
        if tasks is None and self._fusable(pred, mask, terms):
            res = self._fused_term_loss(pred, target, mask, terms)      # finiteness is checked where fit() reads the value
            if res is not None:
                return res
        total, parts = 0.0, {}
        for i, t in terms:
            p, y = pred, target
            if t["pred_slice"] is not None:
                p = pred[:, resolve_channel_indices(t["pred_slice"], num_channels=pred.shape[1], context="pred_slice")]
            if t["target_slice"] is not None:
                y = target[:, resolve_channel_indices(t["target_slice"], num_channels=target.shape[1], context="target_slice")]
            # what the loss sees as its spatial argument (reference orchestrator.py:566-646): a term's own mask channels
            # (`mask_slice` of the labels) x the batch mask; a `weight`-taking loss without a mask of its own gets the
            # class-balancing map instead (not the weighted BCE, whose pos_weight is a scalar of its own)
            own = None
            if t.get("mask_slice") is not None:
                own = target[:, resolve_channel_indices(t["mask_slice"], num_channels=target.shape[1], context="mask_slice")]
            valid = own if mask is None else (mask if own is None else own * mask)
            fn = t["fn"]
            if fn in _WEIGHT_TAKING:
                spatial = own
                if spatial is None and fn != "WeightedBCEWithLogitsLoss":
                    spatial = class_balance_weight(y, 1.0 if (t["pos_weight"] is None and fn in _OWN_CLASS_BALANCE) else t["pos_weight"],
                                                   valid_mask=valid)
                if mask is not None:
                    spatial = mask if spatial is None else spatial * mask
                pw = t["pos_weight"] if fn == "WeightedBCEWithLogitsLoss" else None
                if isinstance(pw, str):
                    pw = auto_pos_weight_scalar(y, valid)
                elif pw is not None and float(pw) == 1.0:
                    pw = None
                v = _LOSSES[fn](p, y, weight=spatial, pos_weight=pw, clamp_min=self.clamp_min, **t["kwargs"])
            else:
                v = _LOSSES[fn](p, y, weight=valid, pos_weight=None, clamp_min=self.clamp_min, **t["kwargs"])
            if not torch.isfinite(v):
                raise FloatingPointError(f"loss term {t['fn']} is not finite")
            parts[f"loss_{i}_{t['fn']}"] = v.detach()
            if tasks is not None:
                tasks[i] = t["weight"] * v
            total = total + t["weight"] * v
        return total, parts

    def _balanced_scale_loss(self, items, stage: str):
        """One output scale under adaptive loss balancing (orchestrator.py:110-127, 779-790): `items` = [(pred, target, mask, terms)]
        (one entry per head); every term is a task whose loss is its static weight x raw value, the weighter combines ALL tasks of
        the scale in one call.  Uncertainty weighting on fusable terms keeps the fused HIP loss: the per-task coefficients
        0.5 * exp(-s_i) multiply the static weights as host scalars (one device -> host read of T floats per scale), the kernel's
        scalar is sum_i coef_i * task_i exactly, and the gradient of the log-variances -- which the kernel knows nothing of -- comes
        from a zero-valued term built on the detached task values."""
        from .balancing import UncertaintyLossWeighter
        w = self.loss_weighter
        idx = sorted(i for _, _, _, terms in items for i, _ in terms)
        if idx != list(range(len(self.loss_terms))):
            raise ValueError(f"adaptive loss balancing combines all {len(self.loss_terms)} loss terms at every scale it sees; this scale "
                             f"carries terms {idx} (apply_deep_supervision: false on a term is not compatible with loss_balancing)")
        names = [f"loss_{i}_{self.loss_terms[i]['fn']}" for i in idx]
        parts: Dict[str, Any] = {}
        if isinstance(w, UncertaintyLossWeighter) and all(self._fusable(torch.clamp(p, self.clamp_min, self.clamp_max), m, terms)
                                                           for p, _, m, terms in items):
            coef = 0.5 * torch.exp(-w.log_vars)
            host = [float(c) for c in coef.detach().cpu().tolist()]
            fused_total, ok = 0.0, True
            raw: Dict[int, torch.Tensor] = {}
            for pred, target, mask, terms in items:
                scaled = [(i, {**t, "weight": t["weight"] * host[i]}) for i, t in terms]
                res = self._fused_term_loss(torch.clamp(pred, min=self.clamp_min, max=self.clamp_max), target, mask, scaled)
                if res is None:
                    ok = False
                    break
                fused_total = fused_total + res[0]
                for i, t in terms:
                    raw[i] = res[1][f"loss_{i}_{t['fn']}"]
                parts.update(res[1])
            if ok:
                tasks = torch.stack([self.loss_terms[i]["weight"] * raw[i].detach().float().reshape(()) for i in idx])
                total = fused_total + ((coef - coef.detach()) * tasks).sum() + 0.5 * w.log_vars.sum()
                wts = torch.exp(-w.log_vars).detach()
                for n, wt in zip(names, wts):
                    parts[f"{n}_balance_weight"] = wt
                    parts[f"{stage}_loss_uncertainty/{n}_weight"] = wt
                parts[f"{stage}_loss_uncertainty/reg"] = (0.5 * w.log_vars.sum()).detach()
                return total, parts
            parts = {}
        tasks: Dict[int, torch.Tensor] = {}
        for pred, target, mask, terms in items:
            _, pr = self._term_loss(pred, target, mask, terms, tasks=tasks)
            parts.update(pr)
        total, wts, logs = w.combine([tasks[i] for i in idx], names, stage)
        for n, wt in zip(names, wts):
            parts[f"{n}_balance_weight"] = wt
        parts.update(logs)
        return total, parts

    def _head_of(self, term_index: int, term, heads) -> str:
        """Which named head a term reads (training/losses/orchestrator.py:328-378): its pred_head, else
        model.primary_head, else the only head; anything else is a configuration error."""
        want = term["pred_head"]
        if want is None:
            primary = getattr(self.cfg.model, "primary_head", None)
            if primary is not None and primary in heads:
                want = primary
            elif len(heads) == 1:
                want = next(iter(heads))
            else:
                raise ValueError(f"Loss term 'loss_{term_index}_{term['fn']}' did not specify pred_head and model.primary_head "
                                 f"is unset, but model output has multiple heads {sorted(heads)}.")
        if want not in heads:
            raise ValueError(f"Loss term 'loss_{term_index}_{term['fn']}' requested pred_head='{want}', but available output "
                             f"heads are {sorted(heads)}.")
        if not isinstance(heads[want], torch.Tensor):
            raise TypeError(f"Output head '{want}' must be a tensor, got {type(heads[want]).__name__}.")
        return want

    def _compute_loss(self, outputs, labels, mask=None):
        stage = "train" if self.training else "val"
        main = unwrap_main_output(outputs)
        if isinstance(main, dict):
            # named heads: every term reads its own head; the terms of one head share a (fused) reduction
            if not main:
                raise ValueError("Named-head model output mapping is empty.")
            by_head: Dict[str, list] = {}
            for i, t in enumerate(self.loss_terms):
                by_head.setdefault(self._head_of(i, t, main), []).append((i, t))
            total, parts = 0.0, {}
            if self.loss_weighter is not None:
                total, parts = self._balanced_scale_loss([(main[head], labels, mask, terms) for head, terms in by_head.items()], stage)
            else:
                for head, terms in by_head.items():
                    v, pr = self._term_loss(main[head], labels, mask, terms)
                    total = total + v
                    parts.update(pr)
            total = self.ds_weights[0] * total
            parts["train_loss_total"] = total.detach()
            return total, parts
        for i, t in enumerate(self.loss_terms):
            if t["pred_head"] is not None:
                raise ValueError(f"Loss term 'loss_{i}_{t['fn']}' requested pred_head='{t['pred_head']}' but the model "
                                 "output is a single tensor.")
        all_terms = list(enumerate(self.loss_terms))
        if self.loss_weighter is not None:
            total, parts = self._balanced_scale_loss([(main, labels, mask, all_terms)], stage)
        else:
            total, parts = self._term_loss(main, labels, mask)
        total = self.ds_weights[0] * total
        if self.deep_supervision and isinstance(outputs, dict):
            for i in range(1, 5):
                ds = outputs.get(f"ds_{i}")
                if ds is None:
                    continue
                tgt = match_target_to_output(labels, ds)
                m = None if mask is None else resize_class_index_to_output(mask, ds)
                on_scales = [(j, t) for j, t in enumerate(self.loss_terms) if t.get("apply_deep_supervision", True)]
                if not on_scales:
                    continue
                if self.loss_weighter is not None:
                    li, _ = self._balanced_scale_loss([(ds, tgt, m, on_scales)], stage)
                else:
                    li, _ = self._term_loss(ds, tgt, m, on_scales)
                total = total + self.ds_weights[i] * li
        parts["train_loss_total"] = total.detach()
        return total, parts

    def training_step(self, batch: Dict[str, torch.Tensor], batch_idx: int = 0) -> torch.Tensor:
        outputs = self(batch["image"])
        loss, self.last_log = self._compute_loss(outputs, batch["label"], batch.get("mask"))
        return loss

    @torch.no_grad()
    def validation_step(self, batch, batch_idx: int = 0) -> Dict[str, torch.Tensor]:
        outputs = self(batch["image"])
        loss, _ = self._compute_loss(outputs, batch["label"], batch.get("mask"))
        main = unwrap_main_output(outputs)
        p = torch.sigmoid(main[:, :1].float()) > 0.5
        t = batch["label"][:, :1] > 0
        union = (p | t).sum().clamp_min(1)
        return {"val_loss_total": loss, "val_jaccard": (p & t).sum().float() / union}

    def configure_optimizers(self):
        # with an adaptive loss weighter the optimizer takes the whole module (its task weights train with the network), as the
        # reference does (lightning/model.py:1160-1162)
        opt = build_optimizer(self.cfg, self if self.loss_weighter is not None else self.model)
        return opt, build_lr_scheduler(self.cfg, opt)

    # ---- checkpoints in the Lightning layout ------------------------------------------------------
    def checkpoint_dict(self, optimizer=None) -> Dict[str, Any]:
        sd = {"model." + k: v.detach().cpu() for k, v in self.model.state_dict().items()}
        if self.loss_weighter is not None:      # a sub-module of the LightningModule in the reference: same key prefix
            sd.update({"loss_weighter." + k: v.detach().cpu() for k, v in self.loss_weighter.state_dict().items() if v is not None})
            if getattr(self.loss_weighter, "initial_losses", None) is not None:
                sd["loss_weighter.initial_losses"] = self.loss_weighter.initial_losses.detach().cpu()
        ck = {"state_dict": sd,
              "global_step": self.global_step,
              "pytc_metadata": {"format_version": 1, "model_arch": str(getattr(self.cfg.model.arch, "type", ""))}}
        ck["epoch"] = int(getattr(self, "current_epoch", 0))
        sched = getattr(self, "_scheduler", None)
        if sched is not None:
            ck["lr_schedulers"] = [sched.state_dict()]
        if optimizer is not None:
            ck["optimizer_states"] = [optimizer.state_dict()]
            if getattr(optimizer, "ema", None):
                # the reference persists the EMA under its callback's state (callbacks.py:732-790: "ema_state",
                # "updates", "decay"); same vocabulary here so either side can resume / evaluate with it
                ck["callbacks"] = {"EMAWeightsCallback": {
                    "ema_state": {k: v.cpu() for k, v in optimizer.ema_state_dict(self.model).items()},
                    "updates": int(optimizer.ema_updates), "decay": float(optimizer.ema_decay)}}
        return ck

    def load_checkpoint_dict(self, ck: Dict[str, Any]) -> None:
        """Model weights + counters now; optimizer / scheduler / EMA state is kept and applied by `restore_training_state`
        once `fit` has built them (a true resume: Adam moments, the LR schedule position and the EMA shadow continue)."""
        sd = {k[len("model."):]: v for k, v in ck["state_dict"].items()
              if k.startswith("model.") and not k.startswith("model.loss_functions.")}
        wsd = {k[len("loss_weighter."):]: v for k, v in ck["state_dict"].items() if k.startswith("loss_weighter.")}
        if wsd and self.loss_weighter is None:
            import warnings
            warnings.warn(f"checkpoint carries adaptive loss-balancing state ({len(wsd)} tensors, e.g. 'loss_weighter.{sorted(wsd)[0]}') but "
                          "model.loss.loss_balancing is not configured: the learned task weights are NOT restored", RuntimeWarning, stacklevel=2)
        elif wsd:
            if "initial_losses" in wsd and hasattr(self.loss_weighter, "initial_losses"):
                # (a None buffer cannot be load_state_dict'ed; a checkpointed reference point always replaces the live one)
                self.loss_weighter.initial_losses = wsd["initial_losses"].clone()
            bad = self.loss_weighter.load_state_dict({k: v for k, v in wsd.items() if k != "initial_losses"}, strict=False)
            missing_w = [k for k in bad.missing_keys if k != "initial_losses"]
            if bad.unexpected_keys or missing_w:
                raise RuntimeError(f"checkpoint loss_weighter state does not match the configured strategy: unexpected "
                                   f"{bad.unexpected_keys}, missing {missing_w}")
        missing, unexpected = self.model.load_state_dict(sd, strict=False)
        # a reference checkpoint may carry deep-supervision heads of a trunk built with them (`out_1..out_4`); anything else
        # missing / unexpected is an architecture mismatch
        bad_unexpected = [k for k in unexpected if not any(f".out_{i}." in "." + k for i in (1, 2, 3, 4))]
        if missing or bad_unexpected:
            raise RuntimeError(f"checkpoint does not match the model: missing {missing[:5]}, unexpected {bad_unexpected[:5]}")
        self.global_step = int(ck.get("global_step", 0))
        self.current_epoch = int(ck.get("epoch", 0))
        self._resume = {k: ck[k] for k in ("optimizer_states", "lr_schedulers", "callbacks") if k in ck}

    def restore_training_state(self, optimizer, scheduler=None) -> bool:
        """Apply the optimizer / scheduler / EMA state of a loaded checkpoint.  Optimizer states written with the reference's
        one-group-per-parameter layout (or torch's) are remapped by parameter order onto this optimizer's merged groups."""
        res = getattr(self, "_resume", None)
        if not res:
            return False
        if res.get("optimizer_states"):
            _load_optimizer_state(optimizer, res["optimizer_states"][0])
        if scheduler is not None and res.get("lr_schedulers"):
            scheduler.load_state_dict(res["lr_schedulers"][0])
        ema = _ema_callback_entry(res)
        if ema and getattr(optimizer, "ema_decay", None) is not None and hasattr(optimizer, "load_ema_state_dict"):
            optimizer.load_ema_state_dict(self.model, ema[EMA_STATE_KEY], int(ema.get("updates", 0)))
        self._resume = None
        return True


EMA_STATE_KEY = "ema_state"                    # the reference callback's vocabulary (callbacks.py:752 `EMAWeightsCallback.EMA_STATE_KEY`)


def _ema_callback_entry(checkpoint: Dict[str, Any]) -> Optional[Dict[str, Any]]:
    """The EMA callback's state inside a Lightning-layout checkpoint: Lightning keys callback states by `state_key`, the class name
    optionally followed by its arguments, hence the prefix match (reference callbacks.py:54-59)."""
    for key, state in (checkpoint.get("callbacks") or {}).items():
        if isinstance(state, dict) and str(key).startswith("EMAWeightsCallback") and state.get(EMA_STATE_KEY):
            return state
    return None


def load_ema_state_dict(checkpoint: Dict[str, Any]) -> Optional[Dict[str, torch.Tensor]]:
    """The EMA weights stored in `checkpoint`, keyed like `model.state_dict()`; None when the run had EMA off or the checkpoint
    carries none (reference training/lightning/callbacks.py:46-60 -- same contract, so `--ema` evaluation reads either side's files)."""
    entry = _ema_callback_entry(checkpoint)
    return dict(entry[EMA_STATE_KEY]) if entry else None


def _load_optimizer_state(optimizer, state: Dict[str, Any]) -> None:
    """optimizer.load_state_dict that tolerates a different GROUPING of the same parameters in the same order: torch maps
    saved state to parameters by position, so the per-parameter state (exp_avg, exp_avg_sq, step) is re-keyed onto this
    optimizer's parameter order and the groups keep their own hyper-parameters (the checkpoint's lr is adopted per
    parameter when the group counts match, as torch would)."""
    saved_groups = state["param_groups"]
    own_groups = optimizer.state_dict()["param_groups"]
    n_saved = sum(len(g["params"]) for g in saved_groups)
    n_own = sum(len(g["params"]) for g in own_groups)
    if n_saved != n_own:
        raise ValueError(f"optimizer state holds {n_saved} parameters, the optimizer {n_own}: different models")
    if [len(g["params"]) for g in saved_groups] == [len(g["params"]) for g in own_groups]:
        optimizer.load_state_dict(state)
        return
    # position of every parameter in the model-order enumeration both layouts were built from
    order_own = [pid for g in own_groups for pid in g["params"]]
    if len(saved_groups) == n_saved:                 # reference layout: one group per parameter, in model order
        order_saved = [g["params"][0] for g in saved_groups]
        model_pos_of_own = _model_order_positions(optimizer)
        remap = {order_saved[model_pos_of_own[i]]: order_own[i] for i in range(n_own)}
    else:
        raise ValueError("optimizer state has a parameter grouping this loader cannot remap "
                         f"({len(saved_groups)} groups for {n_saved} parameters)")
    new_state = {remap[k]: v for k, v in state["state"].items() if k in remap}
    optimizer.load_state_dict({"state": new_state, "param_groups": own_groups})


def _model_order_positions(optimizer):
    """For the i-th parameter in the optimizer's flattened group order: its index in model (registration) order."""
    order = getattr(optimizer, "_pytc_model_order", None)
    if order is None:
        raise ValueError("the optimizer does not know the model order of its parameters (built outside build_optimizer)")
    flat = [p for g in optimizer.param_groups for p in g["params"]]
    return [order[id(p)] for p in flat]


def precision_to_dtype(precision) -> torch.dtype:
    """Lightning precision strings: 16-mixed / bf16-mixed -> bf16 storage on this engine; 32 -> fp32."""
    return torch.bfloat16 if any(s in str(precision) for s in ("16", "bf16")) else torch.float32


def resolve_training_steps(cfg, *, fast_dev_run: int = 0, dataset_steps_per_epoch: Optional[int] = None) -> tuple[int, int]:
    """(total optimizer steps, steps per epoch) from optimization.{max_steps, max_epochs, n_steps_per_epoch}.  The reference's
    default n_steps_per_epoch = -1 means "from the dataset size" (schema/optimization.py:96): with an iterable synthetic /
    sampled source there is no such size, so `dataset_steps_per_epoch` (when the caller knows one) or an explicit error."""
    oc = cfg.optimization
    if fast_dev_run:
        return int(fast_dev_run), int(fast_dev_run)
    per_epoch = getattr(oc, "n_steps_per_epoch", None)
    per_epoch = int(per_epoch) if per_epoch is not None else -1
    max_steps = getattr(oc, "max_steps", None)
    if per_epoch <= 0 and (dataset_steps_per_epoch is None or dataset_steps_per_epoch <= 0) \
            and max_steps is not None and int(max_steps) > 0:
        # no epoch length from either side, but a total: the run is one "epoch" of max_steps optimizer steps
        return int(max_steps), int(max_steps)
    if per_epoch <= 0:
        if dataset_steps_per_epoch is None or dataset_steps_per_epoch <= 0:
            raise ValueError("optimization.n_steps_per_epoch is -1 / unset (auto from the dataset size) but the training source "
                             "is an endless sampler: set optimization.n_steps_per_epoch (or optimization.max_steps)")
        per_epoch = int(dataset_steps_per_epoch)
    total = per_epoch * max(1, int(getattr(oc, "max_epochs", 1) or 1))
    if max_steps is not None and int(max_steps) > 0:
        total = min(total, int(max_steps)) if getattr(oc, "max_epochs", None) else int(max_steps)
    return total, per_epoch


def fit(module: ConnectomicsModule, batches, *, max_steps: int, device, log_every: int = 10, ddp: bool = False,
        log=print, steps_per_epoch: Optional[int] = None):
    """Minimal training loop: forward (HIP) -> loss -> backward (HIP) -> clip -> optimizer (+ scheduler).
    `batches` yields {"image","label"[,"mask"]} dicts of (B,C,D,H,W) tensors.  `max_steps` is the TOTAL step count of the run:
    after `load_checkpoint_dict` the loop continues from `module.global_step` with the optimizer moments, the scheduler
    position and the EMA shadow of the checkpoint.  The scheduler steps per `optimization.scheduler.interval` ('epoch', the
    reference default, every `steps_per_epoch` optimizer steps; or 'step') and `frequency` (lightning/model.py:1174-1200)."""
    cfg = module.cfg
    module.to(device).train()
    inner = getattr(module.model, "model", module.model)
    dt = precision_to_dtype(getattr(cfg.optimization, "precision", "32"))
    for mod in (module.model, inner):
        if hasattr(mod, "compute_dtype"):
            mod.compute_dtype = dt
    net = module
    weighter_params = []
    if ddp:
        from torch.nn.parallel import DistributedDataParallel as DDP
        dev_ids = [device.index] if device.type == "cuda" else None
        # the MedNeXt trunk carries an unused `dummy_tensor` parameter (trainer.py:241-253 uses the same flag).
        # DDP wraps the NETWORK, not the whole module: the loss -- and with adaptive loss balancing the weighter's own parameters
        # (log-variances / task weights) -- is computed after the wrapped forward returns, where DDP's reducer would first mark those
        # parameters unused and then see their gradient hooks fire ("Expected to mark a variable ready only once").  The
        # reference computes its loss inside the wrapped training_step (lightning/model.py:863-910); here the weighter's few gradients
        # are averaged over the ranks by hand below, which gives every rank the same update DDP would.
        net = DDP(module.model, device_ids=dev_ids, find_unused_parameters=True)
        if module.loss_weighter is not None:
            weighter_params = [p for p in module.loss_weighter.parameters() if p.requires_grad]
    opt, sched = module.configure_optimizers()
    module._scheduler = sched
    resumed = module.restore_training_state(opt, sched)
    sc = getattr(cfg.optimization, "scheduler", None)
    interval = str(getattr(sc, "interval", "epoch") or "epoch").lower()
    frequency = max(1, int(getattr(sc, "frequency", 1) or 1))
    if interval not in ("epoch", "step"):
        raise ValueError(f"optimization.scheduler.interval must be 'epoch' or 'step', got {interval!r}")
    per_epoch = int(steps_per_epoch) if steps_per_epoch else max_steps
    plateau = isinstance(sched, torch.optim.lr_scheduler.ReduceLROnPlateau)
    clip = float(getattr(cfg.optimization, "gradient_clip_val", 0.0) or 0.0)
    accum = max(1, int(getattr(cfg.optimization, "accumulate_grad_batches", 1) or 1))
    history = []
    it = iter(batches)
    first = module.global_step if resumed else 0
    epoch_loss, epoch_n = 0.0, 0
    for step in range(first, max_steps):
        opt.zero_grad(set_to_none=True)
        for _ in range(accum):
            batch = {k: v.to(device, non_blocking=True) for k, v in next(it).items()}
            out = net(batch["image"])
            loss, logs = module._compute_loss(out, batch["label"], batch.get("mask"))
            (loss / accum).backward()
        if weighter_params:
            world = torch.distributed.get_world_size()
            flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in weighter_params])
            torch.distributed.all_reduce(flat)
            at = 0
            for p in weighter_params:
                p.grad = (flat[at:at + p.numel()] / world).view_as(p).clone()
                at += p.numel()
        if clip > 0 and not hasattr(opt, "max_grad_norm"):     # FusedAdamW clips inside its update kernel
            # every parameter the optimizer owns (with a loss weighter: its parameters too), as Lightning clips (and as FusedAdamW does)
            torch.nn.utils.clip_grad_norm_([p for g in opt.param_groups for p in g["params"]], clip)
        opt.step()
        module.global_step = step + 1
        history.append(float(loss.detach()))
        epoch_loss, epoch_n = epoch_loss + history[-1], epoch_n + 1
        end_of_epoch = (step + 1) % per_epoch == 0
        if end_of_epoch:
            module.current_epoch += 1
        if sched is not None:
            if interval == "step":
                if (step + 1) % frequency == 0:
                    sched.step(history[-1]) if plateau else sched.step()
            elif end_of_epoch and module.current_epoch % frequency == 0:
                sched.step(epoch_loss / max(1, epoch_n)) if plateau else sched.step()
        if end_of_epoch:
            epoch_loss, epoch_n = 0.0, 0
        if step - first == 2:
            from ..utils.hostgc import quiesce_gc
            quiesce_gc()       # the cyclic GC's full passes cost milliseconds per step in a launch-bound loop
        if not math.isfinite(history[-1]):
            raise FloatingPointError(f"training loss is not finite at step {step}")
        if log and (step % log_every == 0 or step == max_steps - 1):
            log(f"step {step}: loss {history[-1]:.4f} lr {opt.param_groups[0]['lr']:.2e}")
    return history, opt


def synthetic_batches(batch_size: int, patch, *, in_channels=1, out_channels=1, seed=42, device="cpu"):
    """The reference's random demo data (data_factory.py:273-276): image U[0,1), label rand > 0.85."""
    g = torch.Generator(device=device).manual_seed(seed)
    g2 = torch.Generator(device=device).manual_seed(seed + 1000)
    while True:
        img = torch.rand((batch_size, in_channels, *patch), generator=g, device=device)
        lab = (torch.rand((batch_size, out_channels, *patch), generator=g2, device=device) > 0.85).float()
        yield {"image": img, "label": lab}


__all__ = ["ConnectomicsModule", "build_optimizer", "build_lr_scheduler", "WarmupCosineLR", "fit", "resolve_training_steps",
           "match_target_to_output", "synthetic_batches",
           "dice_loss_sigmoid", "weighted_bce_with_logits", "precision_to_dtype"]
