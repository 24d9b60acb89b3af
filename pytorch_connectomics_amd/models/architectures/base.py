"""Base model interface (contract of the reference's models/architectures/base.py:17-87)."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any, Dict, Union

import torch
import torch.nn as nn


class ConnectomicsModel(nn.Module, ABC):
    """forward(x: (B,C,D,H,W)) -> Tensor | {"output": T, "ds_1".."ds_4": T} | {"output": {head: T}}."""

    def __init__(self):
        super().__init__()
        self.supports_deep_supervision = False
        self.output_scales = 1

    @abstractmethod
    def forward(self, x: torch.Tensor) -> Union[torch.Tensor, Dict[str, torch.Tensor]]:
        raise NotImplementedError

    def get_model_info(self) -> Dict[str, Any]:
        total = sum(p.numel() for p in self.parameters())
        trainable = sum(p.numel() for p in self.parameters() if p.requires_grad)
        return {"name": self.__class__.__name__, "deep_supervision": self.supports_deep_supervision,
                "output_scales": self.output_scales, "parameters": total, "trainable_parameters": trainable}

    def __repr__(self) -> str:
        info = self.get_model_info()
        return f"{info['name']}(parameters={info['parameters']:,}, deep_supervision={info['deep_supervision']})"


__all__ = ["ConnectomicsModel"]
