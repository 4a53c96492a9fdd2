"""VideoDecoder forward as a flat sequence of C-ABI ops (reference: SURVEY.md §3.4; model.py:715-748,
temporal_ae.py:64-107, model.py:180-201).

z -> conv_in -> mid (VideoResBlock, AttnBlock, VideoResBlock) -> 4 up levels x 3 VideoResBlocks with a fused
nearest-2x+conv3x3 between levels -> GroupNorm+swish -> AE3DConv.  Same channels-last bf16 layout and the same
kernels as the U-Net; the reference decodes in fp32 (disable_first_stage_autocast), here activations are bf16 with
fp32 accumulation and fp32 GroupNorm statistics (tolerance: tests/test_engine_gpu.py).

AttnBlock (single head, d = C = 512, 4096 tokens): q, k projections, V^T produced directly by a swapped GEMM, then ONE streamed-softmax
MFMA kernel (v3d_attn_vae_d512): scores and probabilities stay in registers, nothing [S, S]-shaped is written (the first version
materialised a 1.2 GB fp32 score tensor + 0.6 GB of probabilities per 18-frame decode; 8.1 + 4.1 GB at the 24 x 9216-token scene shape).
The value bias is added after P.V (softmax rows sum to 1, so P (V + 1 b^T) = P V + b^T).
"""
from __future__ import annotations

import torch

from ..ops import GemmCall, get_ops
from .blocks import Env, Geo, res_spatial, vae_resblock
from .packing import AttnPack, VAEEncPack, VAEPack

F32 = torch.float32


def run_vae_attn(env: Env, g: Geo, p: AttnPack, x: torch.Tensor) -> torch.Tensor:
    ops = env.ops
    n, S, C = g.n, g.S, p.C
    ga, be, eps = p.norm
    hn = ops.groupnorm(x, None, ga, be, n, S, eps=eps, silu=False)
    q = ops.linear(hn, *p.wq)
    k = ops.linear(hn, *p.wk)
    vT = ops.empty((n, C, S), None, x.device)
    ops.gemm(GemmCall(A=p.wv, W=hn.view(n, S, C), out=vT, M=C, N=S, K=C, batch=n))
    if C in getattr(ops, "ATTN_VAE_WIDTHS", ()) and S % 8 == 0:
        # streamed-softmax MFMA kernel: the [n, S, S] scores never exist (model.py:180-201 is SDPA in the reference too)
        o = ops.empty((n * S, C), None, x.device)
        ops.attn_vae(q, k, vT, p.bv, o, n, S, C, float(C) ** -0.5)
        return ops.linear(o, p.proj[0], p.proj[1], res1=x)
    # other widths (not used by V3D / SVD): batched GEMM -> fp32 scores -> row softmax -> batched GEMM
    scores = ops.empty((n, S, S), F32, x.device)
    ops.gemm(GemmCall(A=q.view(n, S, C), W=k.view(n, S, C), out=scores, M=S, N=S, K=C, batch=n, c_acc=float(C) ** -0.5))
    prob = ops.empty((n, S, S), None, x.device)
    ops.softmax_rows(scores, prob)
    o = ops.empty((n, S, C), None, x.device)
    ops.gemm(GemmCall(A=prob, W=vT, out=o, M=S, N=C, K=S, batch=n, bias=p.bv))
    return ops.linear(o.view(n * S, C), p.proj[0], p.proj[1], res1=x)


def run_decoder(pk: VAEPack, z: torch.Tensor, T: int, shard=None) -> torch.Tensor:
    """z [(b t), zc, h, w] fp32 -> [(b t), out_ch, 8h, 8w] fp32 (NCHW, like the reference)."""
    ops = get_ops()
    n, _, H, W = z.shape
    ops.begin_evaluation(z.device)
    if shard is None:
        from ..dist import active_shard
        shard = active_shard()
    if shard is not None:
        # frame-sharded decode: `z` holds this rank's frames of every sample and the time axis the kernels see is the LOCAL one, whatever
        # the caller passed (run_unet does the same).  A caller that chunks the decode (DiffusionEngine.decode_first_stage with
        # en_and_decode_n_samples_a_time < frames) would hand over chunks that are not whole local samples - every rank would then run a
        # different number of collectives: refuse instead of hanging or writing out of bounds.
        if n % shard.T_local != 0:
            raise ValueError(f"frame-sharded decode: {n} latent frames is not a multiple of this rank's {shard.T_local} frames "
                             "(chunked decode_first_stage is not defined under a FrameShard: decode all local frames in one call)")
        T = shard.T_local
    assert n % T == 0, f"{n} latent frames is not a multiple of timesteps={T}"
    env = Env(ops=ops, shard=shard)
    g = Geo(n=n, B=n // T, T=T, H=H, W=W)
    h = ops.nchw_to_nhwc_bf16(z.float().contiguous(), 1.0, pk.z_pad)
    h = ops.conv3x3(h, pk.conv_in[0], pk.conv_in[1], n, H, W)
    for kind, p in pk.mid:
        h = vae_resblock(env, g, p, h) if kind == "res" else run_vae_attn(env, g, p, h)
    for lvl in reversed(range(len(pk.up))):
        for p in pk.up[lvl]["blocks"]:
            h = vae_resblock(env, g, p, h)
        us = pk.up[lvl]["upsample"]
        if us is not None:
            h = ops.conv3x3(h, us[0], us[1], g.n, g.H, g.W, up=2)
            g = Geo(n=g.n, B=g.B, T=g.T, H=g.H * 2, W=g.W * 2)
    ga, be, eps = pk.norm_out
    h = ops.groupnorm(h, None, ga, be, g.n, g.S, eps=eps, silu=True)
    # AE3DConv: conv3x3 -> out_ch (fp32, padded to 4 columns) then the (3,1,1) frame mix on out_ch channels
    y = ops.empty((g.n * g.S, 4), F32, z.device)
    ops.conv3x3(h, pk.conv_out[0], pk.conv_out[1], g.n, g.H, g.W, out=y[:, :pk.out_ch])
    if shard is None:
        out = ops.tmix_small(y, pk.tmix_w, pk.tmix_b, g.B, g.T, g.S, pk.out_ch, 0, g.T - 1)
    else:
        out = shard.tmix_small(ops, y, pk.tmix_w, pk.tmix_b, g, pk.out_ch)
    return out.view(g.n, pk.out_ch, g.H, g.W)


def run_encoder(pk: VAEEncPack, x: torch.Tensor) -> torch.Tensor:
    """VAE Encoder (model.py:575-601): x [n, in_ch, H, W] fp32 -> moments [n, 2 * z_channels, H/8, W/8] fp32 (NCHW, like
    the reference).  conv_in -> per level {2-D ResnetBlocks (+ AttnBlocks), Downsample = zero pad right/bottom + 3x3 stride 2
    (model.py:74-91: `pad_mode = 1` of v3d_gemm)} -> mid (ResnetBlock, AttnBlock, ResnetBlock) -> GroupNorm + swish -> conv_out.
    Same kernels, layout and precision policy as the decoder."""
    ops = get_ops()
    n, _, H, W = x.shape
    ops.begin_evaluation(x.device)
    env = Env(ops=ops, shard=None)
    g = Geo(n=n, B=n, T=1, H=H, W=W)
    h = ops.nchw_to_nhwc_bf16(x.float().contiguous(), 1.0, pk.in_pad)
    h = ops.conv3x3(h, pk.conv_in[0], pk.conv_in[1], n, H, W)
    for lvl in pk.down:
        for i, p in enumerate(lvl["blocks"]):
            h = res_spatial(env, g, p, h, None)
            if lvl["attn"]:
                h = run_vae_attn(env, g, lvl["attn"][i], h)
        ds = lvl["downsample"]
        if ds is not None:
            assert g.H % 2 == 0 and g.W % 2 == 0, "VAE Downsample expects even feature maps"
            h = ops.conv3x3(h, ds[0], ds[1], g.n, g.H, g.W, stride=2, pad_mode=1)
            g = Geo(n=g.n, B=g.B, T=1, H=g.H // 2, W=g.W // 2)
    for kind, p in pk.mid:
        h = res_spatial(env, g, p, h, None) if kind == "res" else run_vae_attn(env, g, p, h)
    ga, be, eps = pk.norm_out
    h = ops.groupnorm(h, None, ga, be, g.n, g.S, eps=eps, silu=True)
    y = ops.conv3x3(h, pk.conv_out[0], pk.conv_out[1], g.n, g.H, g.W, out_dtype=F32)      # [n * S, out_ch] fp32
    return y.view(g.n, g.H, g.W, pk.out_ch).permute(0, 3, 1, 2).contiguous()
