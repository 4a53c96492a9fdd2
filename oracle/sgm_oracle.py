"""TEST INFRASTRUCTURE — fp32 restatement of the reference's algorithm for the V3D hot path (device-agnostic functional torch: it
runs where its input tensors live - on the CPU in the -m "not gpu" tests and in bench.py's cpu_baseline, on the GPU in fp32 as the
checker of the full-width -m gpu tests, tests/conftest.py::device_oracle).

A functional, state-dict-driven torch restatement (NCHW, fp32, no autocast) of
    EulerEDMSampler -> LinearPredictionGuider -> Denoiser(VScalingWithEDMcNoise) -> OpenAIWrapper -> VideoUNet
and AutoencodingEngine.decode -> VideoDecoder, each function citing the reference file:line it follows
(paths relative to the reference tree).  It is NOT the product and nothing in v3d_amd imports it.

Pinning: oracle/gen_golden.py imports the reference's own modules (oracle/ref_import.py) in the build container,
runs them on seeded inputs/weights and stores the outputs under tests/golden/; tests/test_oracle_pinned.py checks
this restatement against those fixtures (rtol 1e-4 / atol 1e-5 per SURVEY.md §8d).  The reference ships no golden
vectors or tests of its own for this path, so those fixtures are the pin.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# ------------------------------------------------------------------------------------------------
# leaf helpers
# ------------------------------------------------------------------------------------------------
def timestep_embedding(t: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    """[cos | sin], freqs exp(-ln(max_period) i / half)   (sgm/modules/diffusionmodules/util.py:207-231)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _lin(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _gn(sd: SD, p: str, x: torch.Tensor, eps: float) -> torch.Tensor:
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps)


def _ln(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _mlp(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """Linear -> SiLU -> Linear stored as Sequential indices 0 and 2."""
    return _lin(sd, p + ".2", F.silu(_lin(sd, p + ".0", x)))


# ------------------------------------------------------------------------------------------------
# U-Net blocks
# ------------------------------------------------------------------------------------------------
def resblock2d(sd: SD, p: str, x: torch.Tensor, emb: torch.Tensor) -> torch.Tensor:
    """ResBlock._forward, dims=2 (openaimodel.py:338-364): GN32(1e-5)+SiLU+conv3x3, + emb, GN+SiLU+conv3x3, + skip."""
    h = F.conv2d(F.silu(_gn(sd, p + ".in_layers.0", x, 1e-5)), sd[p + ".in_layers.2.weight"], sd[p + ".in_layers.2.bias"], padding=1)
    e = _lin(sd, p + ".emb_layers.1", F.silu(emb))
    h = h + e[:, :, None, None]
    h = F.conv2d(F.silu(_gn(sd, p + ".out_layers.0", h, 1e-5)), sd[p + ".out_layers.3.weight"], sd[p + ".out_layers.3.bias"], padding=1)
    if p + ".skip_connection.weight" in sd:
        x = F.conv2d(x, sd[p + ".skip_connection.weight"], sd[p + ".skip_connection.bias"])
    return x + h


def resblock3d(sd: SD, p: str, x: torch.Tensor, emb: Optional[torch.Tensor]) -> torch.Tensor:
    """ResBlock._forward, dims=3, kernel (3,1,1), exchange_temb_dims (openaimodel.py:338-364 via video_model.py:42-55).
    x: [b, c, t, h, w]; emb: [b, t, emb_ch] or None (skip_t_emb, temporal_ae.py:32-44)."""
    h = F.conv3d(F.silu(_gn(sd, p + ".in_layers.0", x, 1e-5)), sd[p + ".in_layers.2.weight"], sd[p + ".in_layers.2.bias"], padding=(1, 0, 0))
    if emb is not None:
        e = _lin(sd, p + ".emb_layers.1", F.silu(emb))             # [b, t, c]
        h = h + e.permute(0, 2, 1)[:, :, :, None, None]             # "b t c ... -> b c t ..."
    h = F.conv3d(F.silu(_gn(sd, p + ".out_layers.0", h, 1e-5)), sd[p + ".out_layers.3.weight"], sd[p + ".out_layers.3.bias"], padding=(1, 0, 0))
    return x + h                                                    # channels never change in time_stack


def _alpha(sd: SD, key: str, ioi: Optional[torch.Tensor]) -> torch.Tensor:
    """AlphaBlender.get_alpha, learned_with_images (util.py:352-363): where(ioi, 1, sigmoid(mix_factor)) -> [b, t]."""
    a = torch.sigmoid(sd[key])
    if ioi is None:
        return a
    return torch.where(ioi.bool(), torch.ones_like(ioi, dtype=a.dtype), a.expand_as(ioi))


def video_resblock(sd: SD, p: str, x: torch.Tensor, emb: torch.Tensor, T: int, ioi: torch.Tensor) -> torch.Tensor:
    """VideoResBlock.forward (video_model.py:62-81)."""
    xs = resblock2d(sd, p, x, emb)
    n, c, h, w = xs.shape
    b = n // T
    x5 = xs.reshape(b, T, c, h, w).permute(0, 2, 1, 3, 4)
    xt = resblock3d(sd, p + ".time_stack", x5, emb.reshape(b, T, -1))
    a = _alpha(sd, p + ".time_mixer.mix_factor", ioi)[:, None, :, None, None]       # "b t -> b 1 t 1 1"
    out = a * x5 + (1.0 - a) * xt
    return out.permute(0, 2, 1, 3, 4).reshape(n, c, h, w)


def attention(sd: SD, p: str, x: torch.Tensor, context: Optional[torch.Tensor], heads: int) -> torch.Tensor:
    """CrossAttention.forward (attention.py:286-349): to_q/k/v (no bias), softmax(q k^T d^-1/2) v, to_out.0."""
    ctx = x if context is None else context
    q, k, v = _lin(sd, p + ".to_q", x), _lin(sd, p + ".to_k", ctx), _lin(sd, p + ".to_v", ctx)
    B, N, _ = q.shape
    d = q.shape[-1] // heads

    def split(t):
        return t.reshape(B, -1, heads, d).permute(0, 2, 1, 3)

    o = F.scaled_dot_product_attention(split(q), split(k), split(v))
    return _lin(sd, p + ".to_out.0", o.permute(0, 2, 1, 3).reshape(B, N, heads * d))


def feedforward(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """FeedForward with GEGLU (attention.py:92-118): proj -> chunk(2) -> x * gelu_erf(gate) -> Linear."""
    a, gate = _lin(sd, p + ".net.0.proj", x).chunk(2, dim=-1)
    return _lin(sd, p + ".net.2", a * F.gelu(gate))


def basic_block(sd: SD, p: str, x: torch.Tensor, context: torch.Tensor, heads: int) -> torch.Tensor:
    """BasicTransformerBlock._forward (attention.py:556-577)."""
    x = attention(sd, p + ".attn1", _ln(sd, p + ".norm1", x), None, heads) + x
    x = attention(sd, p + ".attn2", _ln(sd, p + ".norm2", x), context, heads) + x
    return feedforward(sd, p + ".ff", _ln(sd, p + ".norm3", x)) + x


def video_block(sd: SD, p: str, x: torch.Tensor, context: torch.Tensor, T: int, heads: int) -> torch.Tensor:
    """VideoTransformerBlock._forward (video_attention.py:109-140), ff_in=True, is_res=True.  x: [(b t), s, c]."""
    n, S, C = x.shape
    b = n // T
    x = x.reshape(b, T, S, C).permute(0, 2, 1, 3).reshape(b * S, T, C)            # "(b t) s c -> (b s) t c"
    x = feedforward(sd, p + ".ff_in", _ln(sd, p + ".norm_in", x)) + x
    x = attention(sd, p + ".attn1", _ln(sd, p + ".norm1", x), None, heads) + x
    x = attention(sd, p + ".attn2", _ln(sd, p + ".norm2", x), context, heads) + x
    x = feedforward(sd, p + ".ff", _ln(sd, p + ".norm3", x)) + x
    return x.reshape(b, S, T, C).permute(0, 2, 1, 3).reshape(n, S, C)


def spatial_video_transformer(sd: SD, p: str, x: torch.Tensor, context: torch.Tensor, T: int, ioi: torch.Tensor,
                              heads: int, max_period: float = 10000.0) -> torch.Tensor:
    """SpatialVideoTransformer.forward (video_attention.py:230-301), use_spatial_context, use_linear, depth 1."""
    n, c, h, w = x.shape
    x_in = x
    time_ctx = context[::T].repeat_interleave(h * w, dim=0)                        # (b n) 1 c, frame-0 context per sample
    x = _gn(sd, p + ".norm", x, 1e-6).permute(0, 2, 3, 1).reshape(n, h * w, c)
    x = _lin(sd, p + ".proj_in", x)
    frames = torch.arange(T, device=x.device).repeat(n // T)
    emb = _mlp(sd, p + ".time_pos_embed", timestep_embedding(frames, c, max_period))[:, None, :]
    x = basic_block(sd, p + ".transformer_blocks.0", x, context, heads)
    x_mix = video_block(sd, p + ".time_stack.0", x + emb, time_ctx, T, heads)
    a = _alpha(sd, p + ".time_mixer.mix_factor", ioi).reshape(n, 1, 1)            # "b t -> (b t) 1 1"
    x = a * x + (1.0 - a) * x_mix
    x = _lin(sd, p + ".proj_out", x).reshape(n, h, w, c).permute(0, 3, 1, 2)
    return x + x_in


def unet_forward(sd: SD, cfg: dict, x: torch.Tensor, timesteps: torch.Tensor, context: torch.Tensor, y: torch.Tensor,
                 T: int, ioi: torch.Tensor) -> torch.Tensor:
    """VideoUNet.forward (video_model.py:442-493) with the block layout built by video_model.py:184-434."""
    mc, mult, nres = cfg["model_channels"], cfg["channel_mult"], cfg["num_res_blocks"]
    attn_res, hc = cfg["attention_resolutions"], cfg["num_head_channels"]
    emb = _mlp(sd, "time_embed", timestep_embedding(timesteps, mc))
    emb = emb + _mlp(sd, "label_emb.0", y)

    def stage(prefix: str, h: torch.Tensor, has_attn: bool, ch: int):
        h = video_resblock(sd, prefix + ".0", h, emb, T, ioi)
        if has_attn:
            h = spatial_video_transformer(sd, prefix + ".1", h, context, T, ioi, ch // hc)
        return h

    hs = []
    h = F.conv2d(x, sd["input_blocks.0.0.weight"], sd["input_blocks.0.0.bias"], padding=1)
    hs.append(h)
    idx, ds, ch = 1, 1, mc
    for level, m in enumerate(mult):
        for _ in range(nres):
            ch = m * mc
            h = stage(f"input_blocks.{idx}", h, ds in attn_res, ch)
            hs.append(h)
            idx += 1
        if level != len(mult) - 1:
            h = F.conv2d(h, sd[f"input_blocks.{idx}.0.op.weight"], sd[f"input_blocks.{idx}.0.op.bias"], stride=2, padding=1)
            hs.append(h)
            idx += 1
            ds *= 2
    h = video_resblock(sd, "middle_block.0", h, emb, T, ioi)
    h = spatial_video_transformer(sd, "middle_block.1", h, context, T, ioi, ch // hc)
    h = video_resblock(sd, "middle_block.2", h, emb, T, ioi)
    idx = 0
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nres + 1):
            h = torch.cat([h, hs.pop()], dim=1)
            ch = m * mc
            has_attn = ds in attn_res
            h = stage(f"output_blocks.{idx}", h, has_attn, ch)
            if level and i == nres:
                up = f"output_blocks.{idx}.{2 if has_attn else 1}.conv"
                h = F.interpolate(h, scale_factor=2, mode="nearest")                # openaimodel.py:164-166
                h = F.conv2d(h, sd[up + ".weight"], sd[up + ".bias"], padding=1)
                ds //= 2
            idx += 1
    h = F.silu(_gn(sd, "out.0", h, 1e-5))
    return F.conv2d(h, sd["out.2.weight"], sd["out.2.bias"], padding=1)


# ------------------------------------------------------------------------------------------------
# sampler stack
# ------------------------------------------------------------------------------------------------
def edm_sigmas(n: int, sigma_min: float = 0.002, sigma_max: float = 80.0, rho: float = 7.0) -> torch.Tensor:
    """EDMDiscretization.get_sigmas + append_zero (discretizer.py:17-39)."""
    ramp = torch.linspace(0, 1, n)
    lo, hi = sigma_min ** (1 / rho), sigma_max ** (1 / rho)
    return torch.cat([(hi + ramp * (lo - hi)) ** rho, torch.zeros(1)])


def denoise(net, x: torch.Tensor, sigma: torch.Tensor, cond: dict) -> torch.Tensor:
    """Denoiser.forward with VScalingWithEDMcNoise + OpenAIWrapper (denoiser.py:23-39; denoiser_scaling.py:51-59;
    wrappers.py:24-34).  `net(x8, c_noise, crossattn, vector)` is the U-Net callable."""
    s = sigma.reshape(-1, 1, 1, 1)
    c_skip, c_out, c_in = 1.0 / (s ** 2 + 1.0), -s / (s ** 2 + 1.0) ** 0.5, 1.0 / (s ** 2 + 1.0) ** 0.5
    c_noise = 0.25 * sigma.log()
    xin = torch.cat([x * c_in, cond["concat"]], dim=1)
    return net(xin, c_noise, cond["crossattn"], cond["vector"]) * c_out + x * c_skip


def guider_scale(kind: str, T: int, min_scale: float, max_scale: float) -> torch.Tensor:
    """Per-frame CFG scale: LinearPredictionGuider (guiders.py:61-76), CentralPredictionGuider (guiders.py:104-118),
    VanillaCFG (guiders.py:24-31: one scale = max_scale for every frame)."""
    if kind == "linear":
        return torch.linspace(min_scale, max_scale, T)
    if kind == "central":
        sc = torch.linspace(min_scale, 2 * max_scale, T)
        sc[T // 2:] = 2 * max_scale - sc[T // 2:]
        return sc
    if kind == "vanilla":
        return torch.full((T,), float(max_scale))
    raise ValueError(kind)


def sample_edm(net, x: torch.Tensor, c: dict, uc: dict, num_steps: int, T: int, min_scale: float, max_scale: float,
               sigma_max: float = 700.0, return_all: bool = False, guider: str = "linear", heun: bool = False):
    """EDMSampler.__call__ with s_churn = 0 (sampling.py:44-55,96-133), Euler (214-218) or Heun correction (221-237), and a
    per-frame-scale CFG guider (guiders.py:24-42,78-101,125-146; sampling_utils.py:34-35).
    `x` is NOT modified in place here (callers pass a copy)."""
    sigmas = edm_sigmas(num_steps, sigma_max=sigma_max).to(x.device)
    x = x * torch.sqrt(1.0 + sigmas[0] ** 2.0)
    s_in = x.new_ones([x.shape[0]])
    scale = guider_scale(guider, T, min_scale, max_scale).to(x.device)
    cond = {k: torch.cat([uc[k], c[k]], dim=0) for k in ("vector", "crossattn", "concat")}   # batch = [uc ; c]

    def guided(xx, sg):
        den = denoise(net, torch.cat([xx, xx]), torch.cat([sg, sg]), cond)
        x_u, x_c = den.chunk(2)
        n = x_u.shape[0]
        sc = scale.repeat(n // T).reshape(n, 1, 1, 1)
        return x_u + sc * (x_c - x_u)

    traj = []
    for i in range(num_steps):
        sig, nxt = s_in * sigmas[i], s_in * sigmas[i + 1]
        den = guided(x, sig)
        d = (x - den) / sig.reshape(-1, 1, 1, 1)
        dt = (nxt - sig).reshape(-1, 1, 1, 1)
        euler = x + dt * d
        if heun and float(nxt.sum()) >= 1e-14:
            d_new = (euler - guided(euler, nxt)) / nxt.reshape(-1, 1, 1, 1)
            x = torch.where(nxt.reshape(-1, 1, 1, 1) > 0.0, x + (d + d_new) / 2.0 * dt, euler)
        else:
            x = euler
        if return_all:
            traj.append(x.clone())
    return (x, traj) if return_all else x


def sample_euler_edm(net, x, c, uc, num_steps, T, min_scale, max_scale, sigma_max: float = 700.0, return_all: bool = False):
    """EulerEDMSampler x LinearPredictionGuider (the V3D_512 configuration)."""
    return sample_edm(net, x, c, uc, num_steps, T, min_scale, max_scale, sigma_max, return_all, "linear", False)


# ------------------------------------------------------------------------------------------------
# VAE decoder
# ------------------------------------------------------------------------------------------------
def vae_resblock(sd: SD, p: str, x: torch.Tensor, T: int) -> torch.Tensor:
    """temporal_ae.VideoResBlock.forward over ResnetBlock.forward (temporal_ae.py:64-83; model.py:131-151), temb=None."""
    h = F.conv2d(F.silu(_gn(sd, p + ".norm1", x, 1e-6)), sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    h = F.conv2d(F.silu(_gn(sd, p + ".norm2", h, 1e-6)), sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    if p + ".nin_shortcut.weight" in sd:
        x = F.conv2d(x, sd[p + ".nin_shortcut.weight"], sd[p + ".nin_shortcut.bias"])
    xs = x + h
    n, c, hh, ww = xs.shape
    x5 = xs.reshape(n // T, T, c, hh, ww).permute(0, 2, 1, 3, 4)
    xt = resblock3d(sd, p + ".time_stack", x5, None)
    a = torch.sigmoid(sd[p + ".mix_factor"])
    out = a * xt + (1.0 - a) * x5                                   # note: opposite convention to the U-Net blender
    return out.permute(0, 2, 1, 3, 4).reshape(n, c, hh, ww)


def vae_attn(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """AttnBlock.forward (model.py:180-201): GN(1e-6), 1x1 q/k/v, single-head SDPA (scale C^-1/2), 1x1 proj, + x."""
    n, c, h, w = x.shape
    hn = _gn(sd, p + ".norm", x, 1e-6)
    q, k, v = (F.conv2d(hn, sd[f"{p}.{t}.weight"], sd[f"{p}.{t}.bias"]).reshape(n, c, h * w).permute(0, 2, 1)[:, None]
               for t in ("q", "k", "v"))
    o = F.scaled_dot_product_attention(q, k, v)[:, 0].permute(0, 2, 1).reshape(n, c, h, w)
    return x + F.conv2d(o, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])


def decoder_forward(sd: SD, cfg: dict, z: torch.Tensor, T: int) -> torch.Tensor:
    """VideoDecoder.forward == Decoder.forward with time_mode="conv-only" (model.py:715-748; temporal_ae.py:293-349)."""
    nlev, nres = len(cfg["ch_mult"]), cfg["num_res_blocks"]
    h = F.conv2d(z, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    h = vae_resblock(sd, "mid.block_1", h, T)
    h = vae_attn(sd, "mid.attn_1", h)
    h = vae_resblock(sd, "mid.block_2", h, T)
    for lvl in reversed(range(nlev)):
        for i in range(nres + 1):
            h = vae_resblock(sd, f"up.{lvl}.block.{i}", h, T)
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, sd[f"up.{lvl}.upsample.conv.weight"], sd[f"up.{lvl}.upsample.conv.bias"], padding=1)
    h = F.silu(_gn(sd, "norm_out", h, 1e-6))
    h = F.conv2d(h, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)               # AE3DConv (temporal_ae.py:101-107)
    n, c, hh, ww = h.shape
    h5 = h.reshape(n // T, T, c, hh, ww).permute(0, 2, 1, 3, 4)
    h5 = F.conv3d(h5, sd["conv_out.time_mix_conv.weight"], sd["conv_out.time_mix_conv.bias"], padding=(1, 0, 0))
    return h5.permute(0, 2, 1, 3, 4).reshape(n, c, hh, ww)


def decode_first_stage(sd: SD, cfg: dict, z: torch.Tensor, scale_factor: float, decoding_t: int) -> torch.Tensor:
    """DiffusionEngine.decode_first_stage (video_diffusion.py:182-210): z / scale_factor, chunks of decoding_t frames."""
    z = z / scale_factor
    outs = [decoder_forward(sd, cfg, z[i:i + decoding_t], len(z[i:i + decoding_t])) for i in range(0, z.shape[0], decoding_t)]
    return torch.cat(outs, dim=0)


# ------------------------------------------------------------------------------------------------
# VAE encoder + regulariser
# ------------------------------------------------------------------------------------------------
def resnet2d(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """ResnetBlock.forward with temb = None (model.py:131-151)."""
    h = F.conv2d(F.silu(_gn(sd, p + ".norm1", x, 1e-6)), sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    h = F.conv2d(F.silu(_gn(sd, p + ".norm2", h, 1e-6)), sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    if p + ".nin_shortcut.weight" in sd:
        x = F.conv2d(x, sd[p + ".nin_shortcut.weight"], sd[p + ".nin_shortcut.bias"])
    return x + h


def encoder_forward(sd: SD, cfg: dict, x: torch.Tensor) -> torch.Tensor:
    """Encoder.forward (model.py:575-601): conv_in, per level num_res_blocks ResnetBlocks (+ AttnBlock where the resolution is
    in attn_resolutions), Downsample = F.pad(0,1,0,1) + conv stride 2 padding 0 (model.py:74-91), mid, GN + swish, conv_out."""
    nlev, nres = len(cfg["ch_mult"]), cfg["num_res_blocks"]
    res = cfg["resolution"]
    h = F.conv2d(x, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    for lvl in range(nlev):
        for i in range(nres):
            h = resnet2d(sd, f"down.{lvl}.block.{i}", h)
            if res in cfg["attn_resolutions"]:
                h = vae_attn(sd, f"down.{lvl}.attn.{i}", h)
        if lvl != nlev - 1:
            h = F.conv2d(F.pad(h, (0, 1, 0, 1)), sd[f"down.{lvl}.downsample.conv.weight"], sd[f"down.{lvl}.downsample.conv.bias"], stride=2)
            res //= 2
    h = resnet2d(sd, "mid.block_1", h)
    h = vae_attn(sd, "mid.attn_1", h)
    h = resnet2d(sd, "mid.block_2", h)
    h = F.silu(_gn(sd, "norm_out", h, 1e-6))
    return F.conv2d(h, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)


def gaussian_mode(moments: torch.Tensor) -> torch.Tensor:
    """DiagonalGaussianRegularizer(sample=False) -> posterior.mode() (regularizers/__init__.py:19-31; distributions.py:25-41)."""
    return torch.chunk(moments, 2, dim=1)[0]
