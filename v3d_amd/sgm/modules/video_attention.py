"""Temporal transformer blocks — parameter owners (reference: sgm/modules/video_attention.py:15-140
VideoTransformerBlock, 143-301 SpatialVideoTransformer).  Executed by v3d_amd.engine.unet.run_svt."""
from __future__ import annotations

import torch.nn as nn

from .attention import (BasicTransformerBlock, CrossAttention, FeedForward, MemoryEfficientCrossAttention,
                        SpatialTransformer, _EngineOnly)
from .diffusionmodules.util import AlphaBlender


class VideoTransformerBlock(_EngineOnly):
    ATTENTION_MODES = {"softmax": CrossAttention, "softmax-xformers": MemoryEfficientCrossAttention}

    def __init__(self, dim, n_heads, d_head, dropout=0.0, context_dim=None, gated_ff=True, checkpoint=True,
                 timesteps=None, ff_in=False, inner_dim=None, attn_mode="softmax", disable_self_attn=False,
                 disable_temporal_crossattention=False, switch_temporal_ca_to_sa=False):
        super().__init__()
        attn_cls = self.ATTENTION_MODES[attn_mode]
        self.ff_in = ff_in or inner_dim is not None
        if inner_dim is None:
            inner_dim = dim
        assert int(n_heads * d_head) == inner_dim
        self.is_res = inner_dim == dim
        if disable_self_attn or switch_temporal_ca_to_sa:
            raise NotImplementedError("disable_self_attn / switch_temporal_ca_to_sa are not used by V3D/SVD")
        if self.ff_in:
            self.norm_in = nn.LayerNorm(dim)
            self.ff_in = FeedForward(dim, dim_out=inner_dim, dropout=dropout, glu=gated_ff)
        self.timesteps = timesteps
        self.disable_self_attn = disable_self_attn
        self.attn1 = attn_cls(query_dim=inner_dim, heads=n_heads, dim_head=d_head, dropout=dropout)
        self.ff = FeedForward(inner_dim, dim_out=dim, dropout=dropout, glu=gated_ff)
        if disable_temporal_crossattention:
            self.attn2 = None
        else:
            self.norm2 = nn.LayerNorm(inner_dim)
            self.attn2 = attn_cls(query_dim=inner_dim, context_dim=context_dim, heads=n_heads, dim_head=d_head,
                                  dropout=dropout)
        self.norm1 = nn.LayerNorm(inner_dim)
        self.norm3 = nn.LayerNorm(inner_dim)
        self.switch_temporal_ca_to_sa = switch_temporal_ca_to_sa
        self.checkpoint = checkpoint


class SpatialVideoTransformer(SpatialTransformer):
    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0.0, use_linear=False, context_dim=None,
                 use_spatial_context=False, timesteps=None, merge_strategy: str = "fixed", merge_factor: float = 0.5,
                 time_context_dim=None, ff_in=False, checkpoint=False, time_depth=1, attn_mode="softmax",
                 disable_self_attn=False, disable_temporal_crossattention=False, max_time_embed_period: int = 10000):
        super().__init__(in_channels, n_heads, d_head, depth=depth, dropout=dropout, attn_type=attn_mode,
                         use_checkpoint=checkpoint, context_dim=context_dim, use_linear=use_linear,
                         disable_self_attn=disable_self_attn)
        self.time_depth = time_depth
        self.depth = depth
        self.max_time_embed_period = max_time_embed_period
        inner_dim = n_heads * d_head
        if use_spatial_context:
            time_context_dim = context_dim
        self.time_stack = nn.ModuleList([
            VideoTransformerBlock(inner_dim, n_heads, d_head, dropout=dropout, context_dim=time_context_dim,
                                  timesteps=timesteps, checkpoint=checkpoint, ff_in=ff_in, inner_dim=inner_dim,
                                  attn_mode=attn_mode, disable_self_attn=disable_self_attn,
                                  disable_temporal_crossattention=disable_temporal_crossattention)
            for _ in range(self.depth)])
        assert len(self.time_stack) == len(self.transformer_blocks)
        self.use_spatial_context = use_spatial_context
        time_embed_dim = self.in_channels * 4
        self.time_pos_embed = nn.Sequential(nn.Linear(self.in_channels, time_embed_dim), nn.SiLU(),
                                            nn.Linear(time_embed_dim, self.in_channels))
        self.time_mixer = AlphaBlender(alpha=merge_factor, merge_strategy=merge_strategy)

    def forward(self, x, context=None, time_context=None, timesteps=None, image_only_indicator=None):
        """The block on its own (video_attention.py:230-301): x [(b t), C, H, W], context [(b t), 1, context_dim] -> [(b t), C, H, W].
        use_spatial_context: time_context = context[::timesteps].  Inside VideoUNet the same executor runs (v3d_amd.engine.unet.run_svt)."""
        from ...engine.standalone import spatial_video_transformer
        assert self.use_spatial_context, "only use_spatial_context=True (V3D / SVD) is implemented"
        return spatial_video_transformer(self, x, context, time_context, timesteps, image_only_indicator)
