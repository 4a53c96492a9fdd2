"""Synthetic weights and conditioning for benchmarks, smoke tests and parity fixtures.

The reference's checkpoints (ckpts/V3D_512.ckpt, ckpts/svd_xt.safetensors) are not available offline, and a freshly
constructed network is vacuous: zero-initialised out-convs / proj_out make VideoUNet return exactly 0
(SURVEY.md Appendix B-1).  `seeded_state_dict` therefore draws EVERY tensor — including the zero-initialised ones —
from a generator keyed by the tensor's state-dict name, so the reference modules and this package load bit-identical
weights from the same call, independent of module construction order.
"""
from __future__ import annotations

import hashlib
import math
from typing import Dict, Optional

import torch


def _key_seed(key: str, seed: int) -> int:
    return int.from_bytes(hashlib.sha256(f"{seed}:{key}".encode()).digest()[:8], "little") & ((1 << 62) - 1)


def seeded_tensor(key: str, shape, seed: int = 0, device="cpu") -> torch.Tensor:
    """Deterministic value for one state-dict entry (always drawn on the CPU generator for reproducibility)."""
    g = torch.Generator(device="cpu").manual_seed(_key_seed(key, seed))
    shape = tuple(shape)
    r = torch.randn(shape, generator=g, dtype=torch.float32)
    if key.endswith("mix_factor"):
        t = r * 1.0                                   # sigmoid -> blend weights spread over (0.1 .. 0.9)
    elif len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        t = r * (1.0 / math.sqrt(fan_in))             # variance preserving linear / conv weights
    elif key.endswith("weight"):
        t = 1.0 + 0.1 * r                             # norm gains
    else:
        t = 0.05 * r                                  # biases / norm shifts
    return t.to(device)


def seeded_state_dict(module_or_shapes, seed: int = 0, device="cpu") -> Dict[str, torch.Tensor]:
    """{key: tensor} for every entry of a module's state dict (or of a {key: shape} mapping)."""
    if hasattr(module_or_shapes, "state_dict"):
        shapes = {k: tuple(v.shape) for k, v in module_or_shapes.state_dict().items()}
    else:
        shapes = module_or_shapes
    return {k: seeded_tensor(k, s, seed, device) for k, s in shapes.items()}


@torch.no_grad()
def init_module_seeded(module: torch.nn.Module, seed: int = 0, chunk_device: Optional[str] = None):
    """In-place seeded re-initialisation of a (possibly GPU-resident) module, one tensor at a time."""
    sd = module.state_dict()
    for k, v in sd.items():
        v.copy_(seeded_tensor(k, v.shape, seed).to(device=v.device, dtype=v.dtype))
    if hasattr(module, "invalidate_packed"):
        module.invalidate_packed()
    return module


@torch.no_grad()
def init_module_fast(module: torch.nn.Module, seed: int = 0):
    """Same distribution as `seeded_tensor` but drawn with the module's own device generator (fast for the 1.5 B-param
    U-Net on a GPU; not bit-reproducible across devices — benchmarks only)."""
    for i, (k, v) in enumerate(module.state_dict().items()):
        g = torch.Generator(device=v.device).manual_seed(_key_seed(k, seed))
        r = torch.randn(v.shape, generator=g, dtype=torch.float32, device=v.device)
        if k.endswith("mix_factor"):
            t = r
        elif v.dim() >= 2:
            t = r * (1.0 / math.sqrt(v[0].numel()))
        elif k.endswith("weight"):
            t = 1.0 + 0.1 * r
        else:
            t = 0.05 * r
        v.copy_(t.to(v.dtype))
    if hasattr(module, "invalidate_packed"):
        module.invalidate_packed()
    return module


def sinusoidal(values: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = values.float()[:, None] * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def synthetic_conditioning(T: int, H: int, W: int, seed: int = 23, context_dim: int = 1024, latent_ch: int = 4,
                           fps_id: float = 1, motion_bucket_id: float = 300, cond_aug: float = 0.02, device="cpu",
                           batch: int = 1):
    """The c / uc dicts and the noise that scripts/pub/V3D_512.py builds before the sampler:
    crossattn [B*T,1,ctx] (one CLIP embedding repeated over frames; zeros in uc), concat [B*T,4,H,W] (one noisy latent
    repeated; zeros in uc), vector [B*T,768] = three 256-d sinusoidal embeddings of fps_id, motion_bucket_id, cond_aug
    (identical in c and uc), and noise [B*T,4,H,W]."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    noise = torch.randn(batch * T, latent_ch, H, W, generator=g)
    xatt = torch.randn(batch, 1, context_dim, generator=g).repeat_interleave(T, dim=0)
    concat = torch.randn(batch, latent_ch, H, W, generator=g).repeat_interleave(T, dim=0)
    vec = torch.cat([sinusoidal(torch.tensor([v], dtype=torch.float32), 256) for v in (fps_id, motion_bucket_id, cond_aug)], dim=-1)
    vec = vec.repeat(batch * T, 1)
    c = {"crossattn": xatt, "concat": concat, "vector": vec}
    uc = {"crossattn": torch.zeros_like(xatt), "concat": torch.zeros_like(concat), "vector": vec.clone()}
    mv = lambda d: {k: v.to(device) for k, v in d.items()}
    return noise.to(device), mv(c), mv(uc)


# ---- canonical configurations ---------------------------------------------------------------------------
def unet_config(model_channels: int = 320, attn_type: str = "softmax-xformers") -> dict:
    """network_config.params of scripts/pub/configs/V3D_512.yaml:31-58 (model_channels reducible for tests)."""
    return dict(adm_in_channels=768, num_classes="sequential", use_checkpoint=False, in_channels=8, out_channels=4,
                model_channels=model_channels, attention_resolutions=[4, 2, 1], num_res_blocks=2, channel_mult=[1, 2, 4, 4],
                num_head_channels=64, use_linear_in_transformer=True, transformer_depth=1, context_dim=1024,
                spatial_transformer_attn_type=attn_type, extra_ff_mix_layer=True, use_spatial_context=True,
                merge_strategy="learned_with_images", video_kernel_size=[3, 1, 1])


def encoder_config(ch: int = 128) -> dict:
    """first_stage_config.encoder_config.params of V3D_512.yaml:93-110 (ch reducible for tests)."""
    return dict(attn_type="vanilla", double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=ch,
                ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)


def decoder_config(ch: int = 128) -> dict:
    """first_stage_config.decoder_config.params of V3D_512.yaml:112-131 (ch reducible for tests)."""
    return dict(attn_type="vanilla", double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=ch,
                ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0, video_kernel_size=[3, 1, 1])
