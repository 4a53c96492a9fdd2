"""Network wrappers (reference: sgm/modules/diffusionmodules/wrappers.py:8-34)."""
from __future__ import annotations

import torch
import torch.nn as nn

OPENAIUNETWRAPPER = "v3d_amd.sgm.modules.diffusionmodules.wrappers.OpenAIWrapper"


class IdentityWrapper(nn.Module):
    def __init__(self, diffusion_model, compile_model: bool = False):
        super().__init__()
        if compile_model:
            raise ValueError("compile_model=True is not supported: the network is already hand-written HIP kernels")
        self.diffusion_model = diffusion_model

    def forward(self, *args, **kwargs):
        return self.diffusion_model(*args, **kwargs)


class OpenAIWrapper(IdentityWrapper):
    """Concatenates c["concat"] on the channel axis and maps crossattn/vector to the U-Net kwargs (wrappers.py:23-34)."""

    def forward(self, x: torch.Tensor, t: torch.Tensor, c: dict, **kwargs) -> torch.Tensor:
        return self.forward_scaled(x, None, t, c, **kwargs)

    def forward_scaled(self, x, scale, t, c: dict, **kwargs):
        """Same as forward(x * scale[:, None, None, None], t, c): scale and concat are folded into the U-Net's
        input-packing kernel when the wrapped network supports it."""
        concat = c.get("concat", None)
        model = self.diffusion_model
        if hasattr(model, "forward_fused"):
            return model.forward_fused(x, scale, concat, timesteps=t, context=c.get("crossattn", None),
                                       y=c.get("vector", None), **kwargs)
        if scale is not None:
            x = x * scale.reshape((-1,) + (1,) * (x.dim() - 1))
        if concat is not None:
            x = torch.cat((x, concat.type_as(x)), dim=1)
        return model(x, timesteps=t, context=c.get("crossattn", None), y=c.get("vector", None), **kwargs)
