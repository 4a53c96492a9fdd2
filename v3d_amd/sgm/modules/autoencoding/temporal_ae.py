"""Temporal VAE decoder — the reference's VideoDecoder behind the same constructor / forward / state-dict keys
(reference: sgm/modules/autoencoding/temporal_ae.py:18-83 VideoResBlock, 86-107 AE3DConv, 293-349 VideoDecoder)."""
from __future__ import annotations

from functools import partial
from typing import Callable, Iterable, Union

import torch
import torch.nn as nn

from ..diffusionmodules.model import Decoder, ResnetBlock
from ..diffusionmodules.openaimodel import ResBlock


class VideoResBlock(ResnetBlock):
    """ResnetBlock + time_stack ResBlock(dims=3, skip_t_emb) + learned blend `mix_factor`
    (alpha * temporal + (1 - alpha) * spatial, temporal_ae.py:79-80)."""

    def __init__(self, out_channels, *args, dropout=0.0, video_kernel_size=3, alpha=0.0, merge_strategy="learned", **kwargs):
        super().__init__(out_channels=out_channels, dropout=dropout, *args, **kwargs)
        if video_kernel_size is None:
            video_kernel_size = [3, 1, 1]
        self.time_stack = ResBlock(channels=out_channels, emb_channels=0, dropout=dropout, dims=3, use_scale_shift_norm=False,
                                   use_conv=False, up=False, down=False, kernel_size=video_kernel_size, use_checkpoint=False,
                                   skip_t_emb=True)
        self.merge_strategy = merge_strategy
        if merge_strategy == "fixed":
            self.register_buffer("mix_factor", torch.tensor([float(alpha)], dtype=torch.float32))
        elif merge_strategy == "learned":
            self.register_parameter("mix_factor", nn.Parameter(torch.tensor([float(alpha)], dtype=torch.float32)))
        else:
            raise ValueError(f"unknown merge strategy {merge_strategy}")

    def forward(self, x, temb=None, skip_video=False, timesteps=None):
        """The block on its own (temporal_ae.py:59-82): x [(b t), C, H, W] -> [(b t), out_channels, H, W], timesteps = frames per sample."""
        from ....engine.standalone import vae_video_resblock
        return vae_video_resblock(self, x, temb, skip_video, timesteps)


class AE3DConv(nn.Conv2d):
    """Conv2d followed by a (3,1,1) Conv3d over frames on the output channels (temporal_ae.py:86-107)."""

    def __init__(self, in_channels, out_channels, video_kernel_size=3, *args, **kwargs):
        super().__init__(in_channels, out_channels, *args, **kwargs)
        if isinstance(video_kernel_size, Iterable):
            video_kernel_size = list(video_kernel_size)
            padding = [int(k // 2) for k in video_kernel_size]
        else:
            padding = int(video_kernel_size // 2)
        if video_kernel_size != [3, 1, 1]:
            raise NotImplementedError("AE3DConv: video_kernel_size must be (3,1,1)")
        self.time_mix_conv = nn.Conv3d(in_channels=out_channels, out_channels=out_channels, kernel_size=video_kernel_size, padding=padding)

    def forward(self, *a, **k):
        raise RuntimeError("AE3DConv is executed by its parent VideoDecoder (v3d_amd.engine.vae)")


class VideoDecoder(Decoder):
    available_time_modes = ["all", "conv-only", "attn-only"]

    def __init__(self, *args, video_kernel_size: Union[int, list] = 3, alpha: float = 0.0, merge_strategy: str = "learned",
                 time_mode: str = "conv-only", **kwargs):
        self.video_kernel_size = video_kernel_size
        self.alpha = alpha
        self.merge_strategy = merge_strategy
        self.time_mode = time_mode
        assert time_mode in self.available_time_modes, f"time_mode parameter has to be in {self.available_time_modes}"
        if time_mode != "conv-only":
            raise NotImplementedError("VideoDecoder: only time_mode='conv-only' (the SVD/V3D first stage) is implemented")
        super().__init__(*args, **kwargs)

    def get_last_layer(self, skip_time_mix=False, **kwargs):
        return self.conv_out.time_mix_conv.weight if not skip_time_mix else self.conv_out.weight

    def _make_conv(self) -> Callable:
        return partial(AE3DConv, video_kernel_size=self.video_kernel_size)

    def _make_resblock(self) -> Callable:
        return partial(VideoResBlock, video_kernel_size=self.video_kernel_size, alpha=self.alpha, merge_strategy=self.merge_strategy)

    def packed(self):
        if self._packed is None:
            from ....engine.packing import pack_vae_decoder
            self._packed = pack_vae_decoder(self)
        return self._packed

    @torch.no_grad()
    def forward(self, z: torch.Tensor, timesteps: int = None, skip_video: bool = False, **kwargs) -> torch.Tensor:
        """z [(b t), z_channels, h, w] fp32 -> [(b t), out_ch, 8h, 8w] fp32 (model.py:715-748 with the video factories)."""
        if skip_video:
            raise NotImplementedError("skip_video decode is not used on the V3D path")
        assert timesteps is not None, "VideoDecoder.forward needs timesteps (frames per sample)"
        from ....engine.vae import run_decoder
        return run_decoder(self.packed(), z, int(timesteps))
