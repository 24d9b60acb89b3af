"""Base model interface (contract of the reference's models/architectures/base.py:17-87)."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any, Dict, Union

import torch
import torch.nn as nn


class ConnectomicsModel(nn.Module, ABC):
    """forward(x: (B,C,D,H,W)) -> Tensor | {"output": T, "ds_1".."ds_4": T} | {"output": {head: T}}."""

    supports_deep_supervision: bool
    output_scales: int

    def __init__(self):
        super().__init__()
        self.supports_deep_supervision, self.output_scales = False, 1

    @abstractmethod
    def forward(self, x: torch.Tensor) -> Union[torch.Tensor, Dict[str, torch.Tensor]]:
        raise NotImplementedError

    def get_model_info(self) -> Dict[str, Any]:
        counts = [0, 0]                                   # all, trainable
        for p in self.parameters():
            counts[0] += p.numel()
            counts[1] += p.numel() if p.requires_grad else 0
        return dict(name=type(self).__name__, deep_supervision=self.supports_deep_supervision, output_scales=self.output_scales,
                    parameters=counts[0], trainable_parameters=counts[1])

    def __repr__(self) -> str:
        return "{name}(parameters={parameters:,}, deep_supervision={deep_supervision})".format(**self.get_model_info())


__all__ = ["ConnectomicsModel"]
