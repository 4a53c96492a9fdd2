"""EDM samplers (reference: sgm/modules/diffusionmodules/sampling.py:24-133,214-237; sampling_utils.py:34-35).

Sampler state `x` stays fp32 on the device; every elementwise update is a HIP kernel from libv3d_hip.so.  The
sigma schedule lives on the host, so the loop issues no device->host sync (the reference's `sigmas[i]` compares
and `torch.sum(next_sigma)` checks each force one).
"""
from __future__ import annotations

from typing import Dict, Union

import torch

from ....ops import get_ops
from ...util import default, instantiate_from_config

DEFAULT_GUIDER = {"target": "v3d_amd.sgm.modules.diffusionmodules.guiders.IdentityGuider"}


class BaseDiffusionSampler:
    def __init__(self, discretization_config: Dict, num_steps: Union[int, None] = None,
                 guider_config: Union[Dict, None] = None, verbose: bool = False, device: str = "cuda"):
        self.num_steps = num_steps
        self.discretization = instantiate_from_config(discretization_config)
        self.guider = instantiate_from_config(default(guider_config, DEFAULT_GUIDER))
        self.verbose = verbose
        self.device = device

    def prepare_sampling_loop(self, x, cond, uc=None, num_steps=None):
        sigmas = self.discretization(self.num_steps if num_steps is None else num_steps, device="cpu").float()
        uc = default(uc, cond)
        # x *= sqrt(1 + sigma_0^2), in place on the caller's tensor like the reference (sampling.py:50)
        get_ops().axpb_f32(x, float(torch.sqrt(1.0 + sigmas[0] ** 2.0)), 0.0, out=x)
        num_sigmas = len(sigmas)
        s_in = x.new_ones([x.shape[0]])
        return x, s_in, sigmas, num_sigmas, cond, uc

    def denoise(self, x, denoiser, sigma, cond, uc):
        denoised = denoiser(*self.guider.prepare_inputs(x, sigma, cond, uc))
        return self.guider(denoised, sigma)

    def get_sigma_gen(self, num_sigmas):
        gen = range(num_sigmas - 1)
        if self.verbose:
            from tqdm import tqdm
            print("#" * 30, " Sampling setting ", "#" * 30)
            print(f"Sampler: {self.__class__.__name__}")
            print(f"Discretization: {self.discretization.__class__.__name__}")
            print(f"Guider: {self.guider.__class__.__name__}")
            gen = tqdm(gen, total=num_sigmas, desc=f"Sampling with {self.__class__.__name__} for {num_sigmas} steps")
        return gen


class SingleStepDiffusionSampler(BaseDiffusionSampler):
    def sampler_step(self, sigma, next_sigma, denoiser, x, cond, uc, *args, **kwargs):
        raise NotImplementedError


class EDMSampler(SingleStepDiffusionSampler):
    def __init__(self, s_churn=0.0, s_tmin=0.0, s_tmax=float("inf"), s_noise=1.0, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.s_churn = s_churn
        self.s_tmin = s_tmin
        self.s_tmax = s_tmax
        self.s_noise = s_noise

    def sampler_step(self, sigma, next_sigma, denoiser, x, cond, uc=None, gamma=0.0):
        ops = get_ops()
        sigma_hat = sigma if gamma == 0.0 else ops.axpb_f32(sigma, gamma + 1.0, 0.0)
        if gamma > 0:
            # churn: x += eps * s_noise * sqrt(sigma_hat^2 - sigma^2)   (sampling.py:98-100); V3D_512 uses s_churn = 0
            eps = torch.randn_like(x) * self.s_noise
            x = x + eps * ((sigma_hat ** 2 - sigma ** 2) ** 0.5).reshape((-1,) + (1,) * (x.dim() - 1))
        denoised = self.denoise(x, denoiser, sigma_hat, cond, uc)
        # d = (x - denoised) / sigma_hat ; euler: x + (next_sigma - sigma_hat) d   — one fused kernel
        x = x.contiguous()
        denoised = denoised.contiguous()
        euler = ops.euler_step(x, denoised, sigma_hat, next_sigma)
        # reference signature (euler_step, x, d, dt, next_sigma, ...).  d and dt are never materialised: a sampler that needs them
        # (Heun) gets the tensors they are made of in `_step_ctx` and fuses its update into one kernel
        self._step_ctx = (denoised, sigma_hat)
        out = self.possible_correction_step(euler, x, None, None, next_sigma, denoiser, cond, uc)
        self._step_ctx = None
        return out

    def __call__(self, denoiser, x, cond, uc=None, num_steps=None):
        ops = get_ops()
        x, s_in, sigmas, num_sigmas, cond, uc = self.prepare_sampling_loop(x, cond, uc, num_steps)
        sig = [float(s) for s in sigmas]
        for i in self.get_sigma_gen(num_sigmas):
            gamma = min(self.s_churn / (num_sigmas - 1), 2 ** 0.5 - 1) if self.s_tmin <= sig[i] <= self.s_tmax else 0.0
            self._next_sigma_is_zero = sig[i + 1] < 1e-14 / max(1, x.shape[0])   # host copy of `torch.sum(next_sigma) < 1e-14`
            x = self.sampler_step(ops.axpb_f32(s_in, sig[i], 0.0), ops.axpb_f32(s_in, sig[i + 1], 0.0),
                                  denoiser, x, cond, uc, gamma)
        return x


class EulerEDMSampler(EDMSampler):
    def possible_correction_step(self, euler_step, x, d, dt, next_sigma, denoiser, cond, uc):
        return euler_step


class HeunEDMSampler(EDMSampler):
    """2nd-order correction (sampling.py:221-237): one more guided evaluation at (euler_step, next_sigma), then
    x + dt (d + d_new) / 2 where next_sigma > 0.  The "all noise levels are 0 -> skip the evaluation" test of the reference
    (`torch.sum(next_sigma) < 1e-14`, a device sync there) is decided on the host copy of the sigma schedule."""

    def possible_correction_step(self, euler_step, x, d, dt, next_sigma, denoiser, cond, uc):
        if getattr(self, "_next_sigma_is_zero", False):
            return euler_step
        denoised, sigma_hat = self._step_ctx
        denoised2 = self.denoise(euler_step, denoiser, next_sigma, cond, uc)
        return get_ops().heun_step(x, denoised, euler_step, denoised2.contiguous(), sigma_hat, next_sigma)
