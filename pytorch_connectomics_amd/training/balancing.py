"""Adaptive balancing of the loss terms -- counterpart of the reference's connectomics/training/losses/balancing.py
(:18-222; wired in lightning/model.py:155-169 and orchestrator.py:110-127, 779-790).  Every loss entry of
`model.loss.losses` is one task (plan.py:281-291); the task losses (raw value x its static weight) are combined by

* `uncertainty` (Kendall et al. 2018): sum_i 0.5 exp(-s_i) L_i + 0.5 s_i with a learned log-variance s_i per task;
* `gradnorm` (Chen et al. 2018): sum_i w_i L_i with w = relu(raw) * T / sum(relu(raw)), plus lambda * L1 between the
  weighted gradient norms G_i = w_i ||dL_i/dW|| (W = a small shared parameter set: the last / first / all trainable
  parameters of the model) and their targets mean(G) * (r_i / mean(r))^alpha, r_i = L_i / L_i(first step).  The gradient
  norms are constants of that auxiliary loss: it trains the task weights only.

Plain PyTorch on device tensors: the terms are scalars, there is nothing here for a kernel.  The weighter is a sub-module of
`ConnectomicsModule`, so its parameters are optimised with the network (`configure_optimizers` hands the optimizer the whole
module when a weighter exists, as the reference does)."""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

__all__ = ["UncertaintyLossWeighter", "GradNormLossWeighter", "build_loss_weighter", "select_shared_parameters"]


def select_shared_parameters(model: nn.Module, strategy: str = "last") -> List[nn.Parameter]:
    params = [p for p in model.parameters() if p.requires_grad]
    if not params:
        return []
    strategy = (strategy or "last").lower()
    if strategy == "first":
        return params[:1]
    if strategy == "all":
        return params
    return params[-1:]


class UncertaintyLossWeighter(nn.Module):
    def __init__(self, num_tasks: int):
        super().__init__()
        self.log_vars = nn.Parameter(torch.zeros(num_tasks))

    def combine(self, losses: Sequence[torch.Tensor], names: Sequence[str], stage: str) -> Tuple[torch.Tensor, torch.Tensor, dict]:
        stacked = torch.stack(list(losses))
        weights = torch.exp(-self.log_vars)
        reg = 0.5 * self.log_vars
        total = (0.5 * weights * stacked).sum() + reg.sum()
        logs = {f"{stage}_loss_uncertainty/{n}_weight": w for n, w in zip(names, weights.detach())}
        logs[f"{stage}_loss_uncertainty/reg"] = reg.sum().detach()
        return total, weights.detach(), logs


class GradNormLossWeighter(nn.Module):
    def __init__(self, num_tasks: int, alpha: float = 0.5, gradnorm_lambda: float = 1.0,
                 shared_parameters: Optional[Iterable[nn.Parameter]] = None):
        super().__init__()
        self.alpha, self.gradnorm_lambda = float(alpha), float(gradnorm_lambda)
        self.task_weights = nn.Parameter(torch.ones(num_tasks))
        self.register_buffer("initial_losses", None)
        self.shared_parameters = list(shared_parameters) if shared_parameters is not None else []   # not registered: the model owns them

    def _normalized_weights(self) -> torch.Tensor:
        raw = torch.relu(self.task_weights)
        return raw * (len(raw) / (raw.sum() + 1e-6))

    def combine(self, losses: Sequence[torch.Tensor], names: Sequence[str], stage: str) -> Tuple[torch.Tensor, torch.Tensor, dict]:
        stacked = torch.stack(list(losses))
        weights = self._normalized_weights()
        total = (weights * stacked).sum()
        logs = {f"{stage}_loss_gradnorm/{n}_weight": w for n, w in zip(names, weights.detach())}
        if not self.training or stage != "train" or not self.shared_parameters or len(stacked) == 0:
            return total, weights.detach(), logs
        if self.initial_losses is None:
            self.initial_losses = stacked.detach()
        norms = []
        for loss in losses:       # one backward per task, down to the shared parameters only (the head: a few kernels)
            grads = torch.autograd.grad(loss, self.shared_parameters, retain_graph=True, allow_unused=True)
            vals = [g.norm() for g in grads if g is not None]
            norms.append((torch.stack(vals).mean() if vals else torch.zeros((), device=loss.device)).detach())
        base = torch.stack(norms)
        weighted = weights * base
        ratios = stacked.detach() / (self.initial_losses + 1e-12)
        target = weighted.mean() * (ratios / ratios.mean()) ** self.alpha
        aux = F.l1_loss(weighted, target)
        logs[f"{stage}_loss_gradnorm/reg"] = aux.detach()
        return total + self.gradnorm_lambda * aux, weights.detach(), logs


def build_loss_weighter(cfg, num_tasks: int, model: Optional[nn.Module] = None) -> Optional[nn.Module]:
    """balancing.py:174-222: `model.loss.loss_balancing.strategy` (older flat spelling: `model.loss.strategy`) in
    {uncertainty, gradnorm}; None = static weights."""
    loss_cfg = getattr(getattr(cfg, "model", None), "loss", None)
    if loss_cfg is None:
        return None
    lb = getattr(loss_cfg, "loss_balancing", None)
    if lb is None or getattr(lb, "strategy", None) is None:
        lb = loss_cfg
    strategy = getattr(lb, "strategy", None)
    if strategy is None:
        return None
    strategy = str(strategy).lower()
    if strategy in ("", "none", "static", "fixed"):           # static weights, as configs of rounds 1-4 of this package spelled it
        return None
    if strategy == "uncertainty":
        return UncertaintyLossWeighter(num_tasks)
    if strategy == "gradnorm":
        shared = select_shared_parameters(model, getattr(lb, "gradnorm_parameter_strategy", "last")) if model is not None else []
        return GradNormLossWeighter(num_tasks, alpha=getattr(lb, "gradnorm_alpha", 0.5) if getattr(lb, "gradnorm_alpha", None) is not None else 0.5,
                                    gradnorm_lambda=getattr(lb, "gradnorm_lambda", 1.0) if getattr(lb, "gradnorm_lambda", None) is not None else 1.0,
                                    shared_parameters=shared)
    raise ValueError(f"Unknown loss balancing strategy: {strategy}")
