"""Denoiser (reference: sgm/modules/diffusionmodules/denoiser.py:12-39): D = F(x c_in, c_noise, cond) c_out + x c_skip."""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn as nn

from ....ops import get_ops
from ...util import append_dims, instantiate_from_config


class Denoiser(nn.Module):
    def __init__(self, scaling_config: Dict):
        super().__init__()
        self.scaling = instantiate_from_config(scaling_config)

    def possibly_quantize_sigma(self, sigma: torch.Tensor) -> torch.Tensor:
        return sigma

    def possibly_quantize_c_noise(self, c_noise: torch.Tensor) -> torch.Tensor:
        return c_noise

    def forward(self, network: nn.Module, input: torch.Tensor, sigma: torch.Tensor, cond: Dict,
                **additional_model_inputs) -> torch.Tensor:
        ops = get_ops()
        sigma = self.possibly_quantize_sigma(sigma)
        sigma_shape = sigma.shape
        c_skip, c_out, c_in, c_noise = self.scaling(append_dims(sigma, input.ndim))
        c_noise = self.possibly_quantize_c_noise(c_noise.reshape(sigma_shape))
        n = input.shape[0]
        c_in_v, c_out_v, c_skip_v = (t.reshape(n).contiguous() for t in (c_in, c_out, c_skip))
        if hasattr(network, "forward_scaled"):
            # fused boundary: `input * c_in` and the wrapper's channel concat happen inside one packing kernel
            net = network.forward_scaled(input, c_in_v, c_noise, cond, **additional_model_inputs)
        else:
            net = network(input * c_in, c_noise, cond, **additional_model_inputs)
        if net.dim() == 4 and net.stride(1) == 1 and net.dtype == torch.float32:
            # channels-last fp32 network output ([n, C, H, W] view of [n*H*W, C]) -> fused combine kernel
            cl = net.permute(0, 2, 3, 1)
            if cl.is_contiguous():
                return ops.denoise_combine(cl.reshape(-1, net.shape[1]), input.contiguous(), c_out_v, c_skip_v)
        return net * c_out + input * c_skip
