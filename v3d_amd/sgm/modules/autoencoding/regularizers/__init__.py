"""DiagonalGaussianRegularizer (reference: sgm/modules/autoencoding/regularizers/__init__.py:13-31;
sgm/modules/distributions/distributions.py:25-41).  Encode-side only; the decode hot path never touches it."""
from __future__ import annotations

from typing import Any, Tuple

import torch
import torch.nn as nn


class DiagonalGaussianRegularizer(nn.Module):
    def __init__(self, sample: bool = True):
        super().__init__()
        self.sample = sample

    def get_trainable_parameters(self) -> Any:
        yield from ()

    def forward(self, z: torch.Tensor) -> Tuple[torch.Tensor, dict]:
        mean, logvar = torch.chunk(z, 2, dim=1)
        logvar = torch.clamp(logvar, -30.0, 20.0)
        if self.sample:
            # CPU generator then .to(device), exactly the reference's RNG order (distributions.py:37-41)
            out = mean + torch.exp(0.5 * logvar) * torch.randn(mean.shape).to(device=z.device)
        else:
            out = mean
        kl = 0.5 * torch.sum(mean.pow(2) + logvar.exp() - 1.0 - logvar, dim=[1, 2, 3])
        return out, {"kl_loss": torch.sum(kl) / kl.shape[0]}
