"""Spatial transformer building blocks — parameter owners with the reference's state-dict layout
(reference: sgm/modules/attention.py:92-118 GEGLU/FeedForward, 260-349 CrossAttention, 461-577
BasicTransformerBlock, 623-730 SpatialTransformer).

These modules own the weights (same names/shapes as the reference, so its checkpoints load by key) and describe
the block; the arithmetic is executed by v3d_amd.engine.unet on the HIP kernels, not by per-module forwards.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from ..util import default, exists


def zero_module(module: nn.Module) -> nn.Module:
    for p in module.parameters():
        p.detach().zero_()
    return module


def Normalize(in_channels: int) -> nn.GroupNorm:
    return nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)


class _EngineOnly(nn.Module):
    """Marker base: forward of an inner block is not a public entry point of this build."""

    def forward(self, *args, **kwargs):
        raise RuntimeError(
            f"{self.__class__.__name__} is a parameter owner; run it through its parent network "
            "(VideoUNet / VideoDecoder) or v3d_amd.engine block functions")


class GEGLU(_EngineOnly):
    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class FeedForward(_EngineOnly):
    def __init__(self, dim, dim_out=None, mult=4, glu=False, dropout=0.0):
        super().__init__()
        if not glu:
            raise NotImplementedError("only the gated (GEGLU) feed-forward used by SVD/V3D is implemented")
        inner_dim = int(dim * mult)
        dim_out = default(dim_out, dim)
        self.net = nn.Sequential(GEGLU(dim, inner_dim), nn.Dropout(dropout), nn.Linear(inner_dim, dim_out))

    def forward(self, x):
        """The module on its own (attention.py:82-113); inside the networks it runs fused with its LayerNorm and residual (v3d_ff_fused)."""
        from ...engine.standalone import feed_forward_module
        return feed_forward_module(self, x)


class CrossAttention(_EngineOnly):
    """to_q/to_k/to_v (no bias) + to_out.0 (bias).  Registered under both reference mode keys below."""

    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.0, backend=None, **kwargs):
        super().__init__()
        inner_dim = dim_head * heads
        context_dim = default(context_dim, query_dim)
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.dim_head = dim_head
        self.to_q = nn.Linear(query_dim, inner_dim, bias=False)
        self.to_k = nn.Linear(context_dim, inner_dim, bias=False)
        self.to_v = nn.Linear(context_dim, inner_dim, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner_dim, query_dim), nn.Dropout(dropout))

    def forward(self, x, context=None, mask=None, **kwargs):
        """The module on its own (attention.py:286-349): self-attention over x [B, N, C], or cross-attention to ONE context token."""
        from ...engine.standalone import cross_attention_module
        return cross_attention_module(self, x, context, mask)


MemoryEfficientCrossAttention = CrossAttention  # same maths; the HIP flash kernel serves both mode strings


class BasicTransformerBlock(_EngineOnly):
    ATTENTION_MODES = {"softmax": CrossAttention, "softmax-xformers": MemoryEfficientCrossAttention}

    def __init__(self, dim, n_heads, d_head, dropout=0.0, context_dim=None, gated_ff=True, checkpoint=True,
                 disable_self_attn=False, attn_mode="softmax", sdp_backend=None):
        super().__init__()
        assert attn_mode in self.ATTENTION_MODES
        if disable_self_attn:
            raise NotImplementedError("disable_self_attn is not used by V3D/SVD and not implemented")
        attn_cls = self.ATTENTION_MODES[attn_mode]
        self.disable_self_attn = disable_self_attn
        self.attn1 = attn_cls(query_dim=dim, heads=n_heads, dim_head=d_head, dropout=dropout, context_dim=None)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff)
        self.attn2 = attn_cls(query_dim=dim, context_dim=context_dim, heads=n_heads, dim_head=d_head, dropout=dropout)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)
        self.checkpoint = checkpoint

    def forward(self, x, context=None, **kwargs):
        """The block on its own (attention.py:556-577), x [B, N, C]; inside the networks it is part of run_svt's fused sequence."""
        from ...engine.standalone import basic_transformer_block
        return basic_transformer_block(self, x, context)


class SpatialTransformer(_EngineOnly):
    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0.0, context_dim=None, disable_self_attn=False,
                 use_linear=False, attn_type="softmax", use_checkpoint=True, sdp_backend=None):
        super().__init__()
        if exists(context_dim) and not isinstance(context_dim, (list, tuple)):
            context_dim = [context_dim]
        if exists(context_dim):
            if depth != len(context_dim):
                assert all(c == context_dim[0] for c in context_dim), "need homogenous context_dim to match depth automatically"
                context_dim = depth * [context_dim[0]]
        else:
            context_dim = [None] * depth
        if not use_linear:
            raise NotImplementedError("use_linear_in_transformer=False (1x1-conv projections) is not used by V3D/SVD")
        self.in_channels = in_channels
        self.n_heads, self.d_head = n_heads, d_head
        inner_dim = n_heads * d_head
        self.norm = Normalize(in_channels)
        self.proj_in = nn.Linear(in_channels, inner_dim)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner_dim, n_heads, d_head, dropout=dropout, context_dim=context_dim[d],
                                  disable_self_attn=disable_self_attn, attn_mode=attn_type, checkpoint=use_checkpoint,
                                  sdp_backend=sdp_backend) for d in range(depth)])
        self.proj_out = zero_module(nn.Linear(inner_dim, in_channels))
        self.use_linear = use_linear
