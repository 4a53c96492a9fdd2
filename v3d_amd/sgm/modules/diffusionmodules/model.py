"""VAE building blocks — parameter owners with the reference's state-dict layout
(reference: sgm/modules/diffusionmodules/model.py:52-71 Normalize/Upsample, 74-91 Downsample, 94-151 ResnetBlock,
154-201 AttnBlock, 487-601 Encoder, 604-748 Decoder).  Executed by v3d_amd.engine.vae on the HIP kernels."""
from __future__ import annotations

from typing import Callable

import torch
import torch.nn as nn

from ..attention import _EngineOnly


def Normalize(in_channels: int, num_groups: int = 32) -> nn.GroupNorm:
    return nn.GroupNorm(num_groups=num_groups, num_channels=in_channels, eps=1e-6, affine=True)


class Upsample(_EngineOnly):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        if not with_conv:
            raise NotImplementedError("VAE Upsample without conv is not used by the SVD/V3D autoencoder")
        self.with_conv = with_conv
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)


class Downsample(_EngineOnly):
    """pad (0,1,0,1) + conv3x3 stride 2 pad 0 (model.py:84-88)."""

    def __init__(self, in_channels, with_conv):
        super().__init__()
        if not with_conv:
            raise NotImplementedError("VAE Downsample without conv is not used by the SVD/V3D autoencoder")
        self.with_conv = with_conv
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=2, padding=0)


class ResnetBlock(_EngineOnly):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout, temb_channels=512):
        super().__init__()
        if conv_shortcut or temb_channels > 0:
            raise NotImplementedError("ResnetBlock: conv_shortcut / temb are not used by the SVD/V3D autoencoder")
        self.in_channels = in_channels
        out_channels = in_channels if out_channels is None else out_channels
        self.out_channels = out_channels
        self.use_conv_shortcut = conv_shortcut
        self.norm1 = Normalize(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.norm2 = Normalize(out_channels)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if self.in_channels != self.out_channels:
            self.nin_shortcut = nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)


class AttnBlock(_EngineOnly):
    """Single-head attention over H*W tokens with head dim = channels (model.py:154-201)."""

    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.k = nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.v = nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.proj_out = nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)

    def forward(self, x, **kwargs):
        """The block on its own (model.py:180-201): x [n, C, H, W] -> x + proj_out(attention(norm(x)))."""
        from ....engine.standalone import vae_attn_block
        return vae_attn_block(self, x)


MemoryEfficientAttnBlock = AttnBlock  # same maths (model.py:204-274)


def make_attn(in_channels, attn_type="vanilla", attn_kwargs=None):
    assert attn_type in ["vanilla", "vanilla-xformers", "none"], f"attn_type {attn_type} unknown / not implemented"
    if attn_type == "none":
        return nn.Identity(in_channels)
    assert attn_kwargs is None
    return AttnBlock(in_channels)


class Encoder(nn.Module):
    """VAE encoder (model.py:487-601): parameter owner + forward through v3d_amd.engine.vae.run_encoder."""

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True, use_linear_attn=False,
                 attn_type="vanilla", **ignore_kwargs):
        super().__init__()
        if use_linear_attn:
            raise NotImplementedError("linear attention is not used by the SVD/V3D autoencoder")
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.resolution = resolution
        self.in_channels = in_channels
        self.conv_in = nn.Conv2d(in_channels, self.ch, kernel_size=3, stride=1, padding=1)
        curr_res = resolution
        in_ch_mult = (1,) + tuple(ch_mult)
        self.in_ch_mult = in_ch_mult
        self.down = nn.ModuleList()
        block_in = ch
        for i_level in range(self.num_resolutions):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_in = ch * in_ch_mult[i_level]
            block_out = ch * ch_mult[i_level]
            for _ in range(self.num_res_blocks):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=0, dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(make_attn(block_in, attn_type=attn_type))
            down = nn.Module()
            down.block, down.attn = block, attn
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in, resamp_with_conv)
                curr_res = curr_res // 2
            self.down.append(down)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type=attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, 2 * z_channels if double_z else z_channels, kernel_size=3, stride=1, padding=1)
        self._packed = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_packed())

    def invalidate_packed(self):
        self._packed = None

    def _apply(self, fn, *args, **kwargs):
        self._packed = None
        return super()._apply(fn, *args, **kwargs)

    def packed(self):
        if self._packed is None:
            from ....engine.packing import pack_vae_encoder
            self._packed = pack_vae_encoder(self)
        return self._packed

    @torch.no_grad()
    def forward(self, x):
        """x [n, in_channels, H, W] -> moments [n, 2 * z_channels, H/8, W/8] fp32 (model.py:575-601) on the HIP kernels."""
        from ....engine.vae import run_encoder
        return run_encoder(self.packed(), x)


class Decoder(nn.Module):
    """VAE decoder skeleton (model.py:604-748); the factories `_make_attn/_make_resblock/_make_conv` are the
    reference's own extension points, overridden by VideoDecoder."""

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, give_pre_end=False, tanh_out=False,
                 use_linear_attn=False, attn_type="vanilla", **ignorekwargs):
        super().__init__()
        if use_linear_attn or give_pre_end or tanh_out:
            raise NotImplementedError("Decoder: linear attention / give_pre_end / tanh_out are not used by SVD/V3D")
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.resolution = resolution
        self.in_channels = in_channels
        self.give_pre_end, self.tanh_out = give_pre_end, tanh_out
        block_in = ch * ch_mult[self.num_resolutions - 1]
        curr_res = resolution // 2 ** (self.num_resolutions - 1)
        self.z_shape = (1, z_channels, curr_res, curr_res)
        make_attn_cls = self._make_attn()
        make_resblock_cls = self._make_resblock()
        make_conv_cls = self._make_conv()
        self.conv_in = nn.Conv2d(z_channels, block_in, kernel_size=3, stride=1, padding=1)
        self.mid = nn.Module()
        self.mid.block_1 = make_resblock_cls(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = make_attn_cls(block_in, attn_type=attn_type)
        self.mid.block_2 = make_resblock_cls(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(self.num_res_blocks + 1):
                block.append(make_resblock_cls(in_channels=block_in, out_channels=block_out, temb_channels=0, dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    raise NotImplementedError("attn_resolutions inside the up path are not used by the SVD/V3D autoencoder")
            up = nn.Module()
            up.block, up.attn = block, attn
            if i_level != 0:
                up.upsample = Upsample(block_in, resamp_with_conv)
                curr_res = curr_res * 2
            self.up.insert(0, up)
        self.norm_out = Normalize(block_in)
        self.conv_out = make_conv_cls(block_in, out_ch, kernel_size=3, stride=1, padding=1)
        self._packed = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_packed())

    def _make_attn(self) -> Callable:
        return make_attn

    def _make_resblock(self) -> Callable:
        return ResnetBlock

    def _make_conv(self) -> Callable:
        return nn.Conv2d

    def get_last_layer(self, **kwargs):
        return self.conv_out.weight

    def invalidate_packed(self):
        self._packed = None

    def _apply(self, fn, *args, **kwargs):
        self._packed = None
        return super()._apply(fn, *args, **kwargs)
