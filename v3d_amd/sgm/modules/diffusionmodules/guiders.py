"""Guiders (reference: sgm/modules/diffusionmodules/guiders.py:13-146)."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Dict, List, Optional, Tuple, Union

import torch

from ....ops import get_ops
from ...util import default


class Guider(ABC):
    @abstractmethod
    def __call__(self, x: torch.Tensor, sigma: float) -> torch.Tensor:
        pass

    def prepare_inputs(self, x: torch.Tensor, s: float, c: Dict, uc: Dict) -> Tuple[torch.Tensor, float, Dict]:
        pass


def _cat_cond(c: Dict, uc: Dict, keys) -> Dict:
    out = dict()
    for k in c:
        if k in keys:
            out[k] = torch.cat((uc[k], c[k]), 0)   # batch order [uc ; c] (guiders.py:95)
        else:
            if k == "rgb":
                continue
            assert c[k] == uc[k]
            out[k] = c[k]
    return out


class IdentityGuider(Guider):
    def __call__(self, x, sigma):
        return x

    def prepare_inputs(self, x, s, c, uc):
        return x, s, {k: c[k] for k in c}


class VanillaCFG(Guider):
    def __init__(self, scale: float):
        self.scale = scale

    def __call__(self, x, sigma):
        n = x.shape[0] // 2
        sc = torch.full((1,), float(self.scale), dtype=torch.float32, device=x.device)
        return get_ops().cfg_combine(x.contiguous(), sc, 1)

    def prepare_inputs(self, x, s, c, uc):
        return torch.cat([x] * 2), torch.cat([s] * 2), _cat_cond(c, uc, ["vector", "crossattn", "concat"])


class LinearPredictionGuider(Guider):
    """x_u + s_t (x_c - x_u) with s = linspace(min, max, T) per frame (guiders.py:61-86); one HIP kernel."""

    def __init__(self, max_scale: float, num_frames: int, min_scale: float = 1.0,
                 additional_cond_keys: Optional[Union[List[str], str]] = None):
        self.min_scale = min_scale
        self.max_scale = max_scale
        self.num_frames = num_frames
        self.scale = torch.linspace(min_scale, max_scale, num_frames).unsqueeze(0)
        additional_cond_keys = default(additional_cond_keys, [])
        if isinstance(additional_cond_keys, str):
            additional_cond_keys = [additional_cond_keys]
        self.additional_cond_keys = additional_cond_keys
        self._scale_dev = None

    def _scale_on(self, device):
        if self._scale_dev is None or self._scale_dev.device != device or self._scale_src is not self.scale:
            self._scale_src = self.scale
            self._scale_dev = self.scale.reshape(-1).to(device=device, dtype=torch.float32).contiguous()
        return self._scale_dev

    def __call__(self, x: torch.Tensor, sigma: torch.Tensor) -> torch.Tensor:
        return get_ops().cfg_combine(x.contiguous(), self._scale_on(x.device), self.num_frames)

    def prepare_inputs(self, x, s, c, uc):
        keys = ["vector", "crossattn", "concat"] + self.additional_cond_keys
        return torch.cat([x] * 2), torch.cat([s] * 2), _cat_cond(c, uc, keys)


class CentralPredictionGuider(LinearPredictionGuider):
    """Triangular per-frame scale peaking at the central frame (guiders.py:104-146): linspace(min, 2 max, T) mirrored over the
    second half.  Same kernel as the linear guider - only the scale vector differs."""

    def __init__(self, max_scale: float, num_frames: int, min_scale: float = 1.0,
                 additional_cond_keys: Optional[Union[List[str], str]] = None):
        super().__init__(max_scale, num_frames, min_scale, additional_cond_keys)
        scale = torch.linspace(min_scale, 2 * max_scale, num_frames)
        scale[num_frames // 2:] = 2 * max_scale - scale[num_frames // 2:]
        self.scale = scale.unsqueeze(0)
