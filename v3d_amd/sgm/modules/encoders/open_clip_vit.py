"""Parameter owner with open_clip's `VisionTransformer` attribute / state-dict names (open_clip is a third-party dependency of the
reference - `open-clip-torch` in requirements.txt, absent from its tree; the names below are what
`conditioner.embedders.0.open_clip.model.visual.*` of svd_xt.safetensors / V3D_512.ckpt carries).  The forward runs on the HIP
kernels through v3d_amd.engine.clip; there is no torch fallback."""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn as nn

# open_clip model_configs/ViT-H-14.json (vision_cfg) + embed_dim
VIT_H_14 = dict(image_size=224, patch_size=14, width=1280, layers=32, heads=16, mlp_ratio=4.0, embed_dim=1024)


class _AttentionParams(nn.Module):
    """nn.MultiheadAttention's parameter names (in_proj_weight / in_proj_bias / out_proj.*)."""

    def __init__(self, width: int):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * width, width))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * width))
        self.out_proj = nn.Linear(width, width)
        nn.init.xavier_uniform_(self.in_proj_weight)


class _ResidualAttentionBlock(nn.Module):
    def __init__(self, width: int, mlp_width: int):
        super().__init__()
        self.ln_1 = nn.LayerNorm(width)
        self.attn = _AttentionParams(width)
        self.ln_2 = nn.LayerNorm(width)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(width, mlp_width)), ("gelu", nn.GELU()), ("c_proj", nn.Linear(mlp_width, width))]))


class _Transformer(nn.Module):
    def __init__(self, width: int, layers: int, mlp_width: int):
        super().__init__()
        self.resblocks = nn.ModuleList([_ResidualAttentionBlock(width, mlp_width) for _ in range(layers)])


class VisionTransformer(nn.Module):
    def __init__(self, image_size=224, patch_size=14, width=1280, layers=32, heads=16, mlp_ratio=4.0, embed_dim=1024):
        super().__init__()
        assert image_size % patch_size == 0 and width % heads == 0
        self.image_size, self.patch_size, self.width, self.heads, self.output_tokens = image_size, patch_size, width, heads, False
        g = image_size // patch_size
        scale = width ** -0.5
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn(g * g + 1, width))
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = _Transformer(width, layers, int(width * mlp_ratio))
        self.ln_post = nn.LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, embed_dim))
        self._packed = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_packed())

    def invalidate_packed(self):
        self._packed = None

    def _apply(self, fn, *args, **kwargs):
        self._packed = None
        return super()._apply(fn, *args, **kwargs)

    def packed(self):
        if self._packed is None:
            from ....engine.clip import pack_clip_visual
            self._packed = pack_clip_visual(self)
        return self._packed


class CLIPVisualOnly(nn.Module):
    """Stands where open_clip's CLIP model stands in the reference (`self.model`, its text transformer deleted,
    encoders/modules.py:614-620): only `.visual` is ever called."""

    def __init__(self, **vision_cfg):
        super().__init__()
        self.visual = VisionTransformer(**vision_cfg)
