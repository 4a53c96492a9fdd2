"""Denoiser pre-conditioning (reference: sgm/modules/diffusionmodules/denoiser_scaling.py:51-59)."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Tuple

import torch

from ....ops import get_ops


class DenoiserScaling(ABC):
    @abstractmethod
    def __call__(self, sigma: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        pass


class VScalingWithEDMcNoise(DenoiserScaling):
    """c_skip = 1/(s^2+1), c_out = -s/sqrt(s^2+1), c_in = 1/sqrt(s^2+1), c_noise = ln(s)/4 — one HIP kernel."""

    def __call__(self, sigma: torch.Tensor):
        shape = sigma.shape
        outs = get_ops().edm_scalings(sigma.reshape(-1).contiguous().float())
        return tuple(o.reshape(shape) for o in outs)
