"""Sigma schedules (reference: sgm/modules/diffusionmodules/discretizer.py:17-39).

The schedule is a handful of host scalars per sample, so it is evaluated with torch on the host exactly as the
reference evaluates it on `device="cpu"`; the sampler keeps it as host floats (no device sync inside the loop).
"""
from __future__ import annotations

from abc import abstractmethod

import torch

from ...util import append_zero


class Discretization:
    def __call__(self, n, do_append_zero=True, device="cpu", flip=False):
        sigmas = self.get_sigmas(n, device=device)
        sigmas = append_zero(sigmas) if do_append_zero else sigmas
        return sigmas if not flip else torch.flip(sigmas, (0,))

    @abstractmethod
    def get_sigmas(self, n, device):
        raise NotImplementedError


class EDMDiscretization(Discretization):
    """Karras rho-schedule: sigma_i = (smax^(1/rho) + i/(n-1) (smin^(1/rho) - smax^(1/rho)))^rho  (discretizer.py:28-39)."""

    def __init__(self, sigma_min=0.002, sigma_max=80.0, rho=7.0):
        self.sigma_min = sigma_min
        self.sigma_max = sigma_max
        self.rho = rho

    def get_sigmas(self, n, device="cpu"):
        ramp = torch.linspace(0, 1, n, device=device)
        lo = self.sigma_min ** (1 / self.rho)
        hi = self.sigma_max ** (1 / self.rho)
        return (hi + ramp * (lo - hi)) ** self.rho
