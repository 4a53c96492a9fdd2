"""VideoUNet — the SVD-XT spatio-temporal U-Net behind the reference's plugin API
(reference: sgm/modules/diffusionmodules/video_model.py:12-81 VideoResBlock, 84-493 VideoUNet).

Same constructor params, forward signature and state-dict keys as the reference class, so
`target: sgm.modules.diffusionmodules.video_model.VideoUNet` -> `target: v3d_amd.sgm....VideoUNet` is the whole
switch.  The module tree only owns parameters; `forward` packs them once (bf16, kernel layouts) and runs the
block sequence on the gfx950 kernels through v3d_amd.engine.unet.
"""
from __future__ import annotations

from typing import List, Optional, Union

import torch
import torch.nn as nn

from ...util import default
from ..video_attention import SpatialVideoTransformer
from .openaimodel import Downsample, ResBlock, TimestepEmbedSequential, Upsample
from .util import AlphaBlender, conv_nd, linear, normalization, zero_module


class VideoResBlock(ResBlock):
    def __init__(self, channels: int, emb_channels: int, dropout: float, video_kernel_size: Union[int, List[int]] = 3,
                 merge_strategy: str = "fixed", merge_factor: float = 0.5, out_channels: Optional[int] = None,
                 use_conv: bool = False, use_scale_shift_norm: bool = False, dims: int = 2, use_checkpoint: bool = False,
                 up: bool = False, down: bool = False):
        super().__init__(channels, emb_channels, dropout, out_channels=out_channels, use_conv=use_conv,
                         use_scale_shift_norm=use_scale_shift_norm, dims=dims, use_checkpoint=use_checkpoint, up=up, down=down)
        oc = default(out_channels, channels)
        self.time_stack = ResBlock(oc, emb_channels, dropout=dropout, dims=3, out_channels=oc, use_scale_shift_norm=False,
                                   use_conv=False, up=False, down=False, kernel_size=video_kernel_size,
                                   use_checkpoint=use_checkpoint, exchange_temb_dims=True)
        self.time_mixer = AlphaBlender(alpha=merge_factor, merge_strategy=merge_strategy, rearrange_pattern="b t -> b 1 t 1 1")

    def forward(self, x: torch.Tensor, emb: torch.Tensor, num_video_frames: int, image_only_indicator: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The block on its own (video_model.py:68-101): x [(b t), C, H, W], emb [(b t), emb_channels] -> [(b t), out_channels, H, W].
        Inside VideoUNet the same executors run with every block's embedding projected in one GEMM (v3d_amd.engine.unet)."""
        from ....engine.standalone import unet_video_resblock
        return unet_video_resblock(self, x, emb, num_video_frames, image_only_indicator)


class VideoUNet(nn.Module):
    def __init__(self, in_channels: int, model_channels: int, out_channels: int, num_res_blocks: int,
                 attention_resolutions: int, dropout: float = 0.0, channel_mult: List[int] = (1, 2, 4, 8),
                 conv_resample: bool = True, dims: int = 2, num_classes: Optional[int] = None, use_checkpoint: bool = False,
                 num_heads: int = -1, num_head_channels: int = -1, num_heads_upsample: int = -1,
                 use_scale_shift_norm: bool = False, resblock_updown: bool = False,
                 transformer_depth: Union[List[int], int] = 1, transformer_depth_middle: Optional[int] = None,
                 context_dim: Optional[int] = None, time_downup: bool = False, time_context_dim: Optional[int] = None,
                 extra_ff_mix_layer: bool = False, use_spatial_context: bool = False, merge_strategy: str = "fixed",
                 merge_factor: float = 0.5, spatial_transformer_attn_type: str = "softmax",
                 video_kernel_size: Union[int, List[int]] = 3, use_linear_in_transformer: bool = False,
                 adm_in_channels: Optional[int] = None, disable_temporal_crossattention: bool = False,
                 max_ddpm_temb_period: int = 10000):
        super().__init__()
        assert context_dim is not None
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        if num_heads == -1:
            assert num_head_channels != -1
        if num_head_channels == -1:
            assert num_heads != -1
        if dims != 2 or resblock_updown or time_downup or not conv_resample:
            raise NotImplementedError("VideoUNet: only dims=2, conv resampling, no resblock_updown/time_downup (the SVD/V3D family)")
        if num_classes not in (None, "sequential"):
            raise NotImplementedError("VideoUNet: num_classes must be None or 'sequential' (the SVD/V3D family)")
        if not (use_spatial_context and extra_ff_mix_layer):
            raise NotImplementedError("VideoUNet: use_spatial_context and extra_ff_mix_layer are required (the SVD/V3D family)")

        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        if isinstance(transformer_depth, int):
            transformer_depth = len(channel_mult) * [transformer_depth]
        transformer_depth_middle = default(transformer_depth_middle, transformer_depth[-1])
        self.num_res_blocks = num_res_blocks
        self.attention_resolutions = attention_resolutions
        self.dropout = dropout
        self.channel_mult = channel_mult
        self.conv_resample = conv_resample
        self.num_classes = num_classes
        self.use_checkpoint = use_checkpoint
        self.num_heads, self.num_head_channels, self.num_heads_upsample = num_heads, num_head_channels, num_heads_upsample
        self.context_dim = context_dim
        self.merge_strategy = merge_strategy

        ted = model_channels * 4
        self.time_embed = nn.Sequential(linear(model_channels, ted), nn.SiLU(), linear(ted, ted))
        if self.num_classes == "sequential":
            assert adm_in_channels is not None
            self.label_emb = nn.Sequential(nn.Sequential(linear(adm_in_channels, ted), nn.SiLU(), linear(ted, ted)))

        def attn(ch, heads, dim_head, depth):
            return SpatialVideoTransformer(
                ch, heads, dim_head, depth=depth, context_dim=context_dim, time_context_dim=time_context_dim,
                dropout=dropout, ff_in=extra_ff_mix_layer, use_spatial_context=use_spatial_context,
                merge_strategy=merge_strategy, merge_factor=merge_factor, checkpoint=use_checkpoint,
                use_linear=use_linear_in_transformer, attn_mode=spatial_transformer_attn_type, disable_self_attn=False,
                disable_temporal_crossattention=disable_temporal_crossattention, max_time_embed_period=max_ddpm_temb_period)

        def res(ch, out_ch):
            return VideoResBlock(merge_factor=merge_factor, merge_strategy=merge_strategy, video_kernel_size=video_kernel_size,
                                 channels=ch, emb_channels=ted, dropout=dropout, out_channels=out_ch, dims=dims,
                                 use_checkpoint=use_checkpoint, use_scale_shift_norm=use_scale_shift_norm)

        def heads_of(ch):
            if num_head_channels == -1:
                return num_heads, ch // num_heads
            return ch // num_head_channels, num_head_channels

        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(conv_nd(dims, in_channels, model_channels, 3, padding=1))])
        chans = [model_channels]
        ch, ds = model_channels, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [res(ch, mult * model_channels)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers.append(attn(ch, *heads_of(ch), depth=transformer_depth[level]))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                chans.append(ch)
            if level != len(channel_mult) - 1:
                ds *= 2
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, conv_resample, dims=dims, out_channels=ch)))
                chans.append(ch)
        self.middle_block = TimestepEmbedSequential(res(ch, None), attn(ch, *heads_of(ch), depth=transformer_depth_middle), res(ch, None))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = chans.pop()
                rb = res(ch + ich, model_channels * mult)
                rb.concat_split = (ch, ich)  # channels of [h, skip] feeding this block: the concat is never materialised
                layers = [rb]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    layers.append(attn(ch, *heads_of(ch), depth=transformer_depth[level]))
                if level and i == num_res_blocks:
                    ds //= 2
                    layers.append(Upsample(ch, conv_resample, dims=dims, out_channels=ch))
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(normalization(ch), nn.SiLU(), zero_module(conv_nd(dims, model_channels, out_channels, 3, padding=1)))

        self._packed = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_packed())

    # ---- packed-weight cache ------------------------------------------------------------------------
    def invalidate_packed(self):
        self._packed = None

    def _apply(self, fn, *args, **kwargs):
        self._packed = None  # .to()/.cuda()/.half() change the owners -> repack lazily
        return super()._apply(fn, *args, **kwargs)

    def packed(self):
        if self._packed is None:
            from ....engine.packing import pack_unet
            self._packed = pack_unet(self)
        return self._packed

    # ---- forward --------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, x: torch.Tensor, timesteps: torch.Tensor, context: Optional[torch.Tensor] = None,
                y: Optional[torch.Tensor] = None, time_context: Optional[torch.Tensor] = None,
                num_video_frames: Optional[int] = None, image_only_indicator: Optional[torch.Tensor] = None):
        return self.forward_fused(x, None, None, timesteps, context, y, time_context, num_video_frames, image_only_indicator)

    @torch.no_grad()
    def forward_fused(self, x, scale, concat, timesteps, context=None, y=None, time_context=None, num_video_frames=None,
                      image_only_indicator=None):
        """forward(cat(x * scale[:, None, None, None], concat), ...) with the scaling / concat / NCHW->channels-last /
        bf16 cast done by one packing kernel.  Returns an [N, out_channels, H, W] fp32 view of the channels-last result."""
        assert (y is not None) == (self.num_classes is not None), "must specify y if and only if the model is class-conditional"
        if y is not None:
            assert y.shape[0] == x.shape[0]
        if time_context is not None:
            raise NotImplementedError("VideoUNet: explicit time_context is not used with use_spatial_context=True")
        from ....engine.unet import run_unet
        return run_unet(self.packed(), x, scale, concat, timesteps, context, y, num_video_frames, image_only_indicator)
