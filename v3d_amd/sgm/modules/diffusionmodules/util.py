"""Small building blocks shared by the U-Net owners (reference: sgm/modules/diffusionmodules/util.py:259-369)."""
from __future__ import annotations

import torch
import torch.nn as nn


def zero_module(module: nn.Module) -> nn.Module:
    for p in module.parameters():
        p.detach().zero_()
    return module


def normalization(channels: int) -> nn.GroupNorm:
    """GroupNorm32: 32 groups, eps 1e-5, statistics in fp32 (util.py:259-276) — the HIP GroupNorm always uses fp32 stats."""
    return nn.GroupNorm(32, channels)


def conv_nd(dims, *args, **kwargs):
    if dims == 1:
        return nn.Conv1d(*args, **kwargs)
    if dims == 2:
        return nn.Conv2d(*args, **kwargs)
    if dims == 3:
        return nn.Conv3d(*args, **kwargs)
    raise ValueError(f"unsupported dimensions: {dims}")


def linear(*args, **kwargs):
    return nn.Linear(*args, **kwargs)


class AlphaBlender(nn.Module):
    """Owns `mix_factor`; alpha = sigmoid(mix_factor) (or 1 where image_only_indicator is set) is turned into the
    epilogue coefficients of the temporal branch's last GEMM by the v3d_blend_coefs kernel (util.py:341-369)."""

    strategies = ["learned", "fixed", "learned_with_images"]

    def __init__(self, alpha: float, merge_strategy: str = "learned_with_images", rearrange_pattern: str = "b t -> (b t) 1 1"):
        super().__init__()
        self.merge_strategy = merge_strategy
        self.rearrange_pattern = rearrange_pattern
        assert merge_strategy in self.strategies, f"merge_strategy needs to be in {self.strategies}"
        if merge_strategy == "fixed":
            self.register_buffer("mix_factor", torch.tensor([float(alpha)], dtype=torch.float32))
        else:
            self.register_parameter("mix_factor", nn.Parameter(torch.tensor([float(alpha)], dtype=torch.float32)))

    def alpha_value(self) -> float:
        """Host scalar alpha for image_only_indicator == 0 (pack-time; one-off sync)."""
        m = float(self.mix_factor.detach().float().cpu())
        if self.merge_strategy == "fixed":
            return m
        return float(torch.sigmoid(torch.tensor(m)))
