"""U-Net building blocks — parameter owners (reference: sgm/modules/diffusionmodules/openaimodel.py:67-114
TimestepEmbedSequential, 117-217 Upsample/Downsample, 220-364 ResBlock)."""
from __future__ import annotations

from typing import Iterable, Optional

import torch.nn as nn

from ..attention import _EngineOnly
from .util import conv_nd, linear, normalization, zero_module


class TimestepBlock(nn.Module):
    """Marker for modules whose forward takes the timestep embedding."""


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    """Ordered container of one U-Net stage; the engine dispatches on the member types exactly like the
    reference's isinstance chain (openaimodel.py:82-114)."""

    def forward(self, *args, **kwargs):
        raise RuntimeError("TimestepEmbedSequential is executed by its parent VideoUNet (v3d_amd.engine.unet)")


class Upsample(_EngineOnly):
    """nearest 2x then conv3x3 — fused into one implicit-GEMM launch (openaimodel.py:149-167)."""

    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1, third_up=False, kernel_size=3, scale_factor=2):
        super().__init__()
        if dims != 2 or not use_conv or kernel_size != 3 or scale_factor != 2 or padding != 1:
            raise NotImplementedError("only the 2-D nearest-2x + conv3x3 upsample of SVD/V3D is implemented")
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv, self.dims, self.third_up, self.scale_factor = use_conv, dims, third_up, scale_factor
        self.conv = conv_nd(dims, self.channels, self.out_channels, kernel_size, padding=padding)


class Downsample(_EngineOnly):
    """conv3x3 stride 2 pad 1 (openaimodel.py:202-217)."""

    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1, third_down=False):
        super().__init__()
        if dims != 2 or not use_conv or padding != 1:
            raise NotImplementedError("only the 2-D strided-conv downsample of SVD/V3D is implemented")
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv, self.dims = use_conv, dims
        self.op = conv_nd(dims, self.channels, self.out_channels, 3, stride=2, padding=padding)


class ResBlock(TimestepBlock, _EngineOnly):
    """GN+SiLU+conv -> (+emb) -> GN+SiLU+conv(zero-init) -> +skip.  dims=2 (3x3) or dims=3 with kernel (3,1,1)."""

    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False, use_scale_shift_norm=False,
                 dims=2, use_checkpoint=False, up=False, down=False, kernel_size=3, exchange_temb_dims=False,
                 skip_t_emb=False):
        super().__init__()
        if up or down or use_scale_shift_norm or use_conv:
            raise NotImplementedError("resblock_updown / use_scale_shift_norm / use_conv skip are not used by SVD/V3D")
        self.channels = channels
        self.emb_channels = emb_channels
        self.dropout = dropout
        self.out_channels = out_channels or channels
        self.use_checkpoint = use_checkpoint
        self.use_scale_shift_norm = use_scale_shift_norm
        self.exchange_temb_dims = exchange_temb_dims
        self.dims = dims
        if isinstance(kernel_size, Iterable):
            kernel_size = list(kernel_size)
            padding = [k // 2 for k in kernel_size]
        else:
            padding = kernel_size // 2
        if dims == 2 and kernel_size != 3:
            raise NotImplementedError("2-D ResBlock kernel must be 3x3")
        if dims == 3 and kernel_size != [3, 1, 1]:
            raise NotImplementedError("3-D ResBlock kernel must be (3,1,1)")
        self.kernel_size = kernel_size
        self.in_layers = nn.Sequential(normalization(channels), nn.SiLU(),
                                       conv_nd(dims, channels, self.out_channels, kernel_size, padding=padding))
        self.updown = False
        self.skip_t_emb = skip_t_emb
        self.emb_out_channels = self.out_channels
        if self.skip_t_emb:
            self.emb_layers = None
            self.exchange_temb_dims = False
        else:
            self.emb_layers = nn.Sequential(nn.SiLU(), linear(emb_channels, self.emb_out_channels))
        self.out_layers = nn.Sequential(
            normalization(self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
            zero_module(conv_nd(dims, self.out_channels, self.out_channels, kernel_size, padding=padding)))
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        else:
            self.skip_connection = conv_nd(dims, channels, self.out_channels, 1)
