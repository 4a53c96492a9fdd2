"""Block executors shared by the U-Net and the VAE decoder: the (Video)ResBlock as a sequence of C-ABI ops.

Activations are channels-last bf16 [(b t) * H*W, C]; the frame axis is the row-block stride H*W, so the
reference's "(b t) c h w -> b c t h w" rearranges (video_model.py:71-72,80) do not exist here: the 3-D GroupNorm
is a GroupNorm whose statistics group spans T consecutive images, the (3,1,1) conv a 3-tap GEMM along that stride.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional

import torch

from ..ops import OpsBase
from .packing import ResPack


@dataclass
class Geo:
    """Geometry of the current activation: n images = B samples x T local frames of H x W positions."""
    n: int
    B: int
    T: int
    H: int
    W: int

    @property
    def S(self) -> int:
        return self.H * self.W


@dataclass
class Env:
    """Per-evaluation context handed to the block executors."""
    ops: OpsBase
    emb_all: Optional[torch.Tensor] = None     # fp32 [n, emb_ld]: every ResBlock's emb projection (one GEMM)
    ctx_all: Optional[torch.Tensor] = None     # fp32 [n, ctx_ld]: every block's collapsed 1-token cross-attention
    coefs: Optional[torch.Tensor] = None       # fp32 [n_mixers, n, 3]: AlphaBlender epilogue coefficients
    shard: Optional[object] = None             # v3d_amd.dist.FrameShard when the frame axis is sharded over ranks


# GroupNorm statistics gathered by the GEMM that produces the tensor (GemmCall.gn_stats: the v3 <GN> epilogue, or the stand-alone pass inside
# v3d_gemm for launches on other kernels) instead of a separate read of it: three of the four norms of a VideoResBlock read a tensor this
# module has just produced.  Measured (tools/gn_epi_bench.py, same-box A/B of bench.py): behind a 3x3 convolution the epilogue costs 0-5 us
# against an 11-24 us stand-alone pass; behind the 3-tap temporal convolution (a third of the main loop to hide behind) it costs 14-16 us -
# a loss below the 64x64 level - so only the 3x3 producers use it: 44 launches per evaluation fewer, +0.3 % end to end.
# V3D_GN_EPILOGUE=0 restores the separate v3d_groupnorm_stats launches, =2 also uses the epilogue of the temporal convolutions (A/B knob).
_GN_EPILOGUE = int(os.environ.get("V3D_GN_EPILOGUE", "1") or 0)


def _gn_producer(ops, n_stat, rps, cout, device, groups=32, level=1):
    """(stats buffer, GemmCall keywords) for a GEMM whose [M, cout] output feeds a 32-group GroupNorm with `rps` rows per statistics group."""
    if _GN_EPILOGUE < level or cout % (2 * groups) or rps % 16:
        return None, {}
    st = ops.gn_stats_buffer(n_stat, device, groups)
    return st, dict(gn_stats=st, gn_rps=rps, gn_cpg=cout // groups)


def res_spatial(env: Env, g: Geo, p: ResPack, x1: torch.Tensor, x2: Optional[torch.Tensor], *, eps_override=None, out_stats_imgs=0):
    """2-D ResBlock (openaimodel.py:338-364 / model.py:131-151) on channels-last input (x1 [| x2] concatenated
    on channels but never materialised).  Returns xs [n*S, cout]; with out_stats_imgs = k > 0 also the GroupNorm partial sums of xs over
    groups of k images (the 3-D norm of the time_stack that follows), gathered by the last convolution: (xs, stats)."""
    ops = env.ops
    S = g.S
    ga, be, eps = p.gn1
    h = ops.groupnorm(x1, x2, ga, be, g.n, S, eps=eps, silu=True)
    epi = {}
    if p.emb_off >= 0:
        epi = dict(add=env.emb_all[:, p.emb_off:], add_rpg=S, add_ld=env.emb_all.stride(0))
    cout = p.w1.shape[-2]
    st2, gkw = _gn_producer(ops, g.n, S, cout, x1.device)
    h = ops.conv3x3(h, p.w1, p.b1, g.n, g.H, g.W, **epi, **gkw)
    ga, be, eps = p.gn2
    h = ops.groupnorm(h, None, ga, be, g.n, S, eps=eps, silu=True, stats=st2)
    if p.skip_w is None:
        assert x2 is None
        skip = x1
    else:
        if x2 is None:
            skip = ops.linear(x1, p.skip_w, p.skip_b)
        else:
            c1 = x1.shape[-1]
            skip = ops.linear(x1, p.skip_w[:, :c1], p.skip_b)
            skip = ops.linear(x2, p.skip_w[:, c1:], None, res1=skip)
    if not out_stats_imgs:
        return ops.conv3x3(h, p.w2, p.b2, g.n, g.H, g.W, res1=skip)
    st, gkw = _gn_producer(ops, g.n // out_stats_imgs, out_stats_imgs * S, p.w2.shape[-2], x1.device)
    return ops.conv3x3(h, p.w2, p.b2, g.n, g.H, g.W, res1=skip, **gkw), st


def res_temporal(env: Env, g: Geo, p: ResPack, xs: torch.Tensor, *, coef=None, c_acc=1.0, xs_stats=None):
    """time_stack ResBlock(dims=3, kernel (3,1,1)) + blend (video_model.py:74-79 / temporal_ae.py:73-80):
    out = xs + c * (conv_t(GN3d+SiLU(conv_t(GN3d+SiLU(xs)) + emb_t)) + bias), c from the blend table or scalar."""
    ops = env.ops
    S, T = g.S, g.T
    sh = env.shard
    gn_kw = dict(imgs_per_stat=T)
    C = xs.shape[-1]
    if sh is not None:
        gn_kw.update(stats_hook=sh.allreduce_stats, count_imgs=sh.T_global)
    # frame-sharded: GroupNorm writes the local frames straight into the middle of the split-halo buffer the 3-tap GEMM reads
    buf, mid = sh.halo_buffer(g.B, S, C, ops.act_dtype, xs.device) if sh is not None else (None, None)
    ga, be, eps = p.t_gn1
    h = ops.groupnorm(xs, None, ga, be, g.n, S, eps=eps, silu=True, out=mid, stats=xs_stats, **gn_kw)
    epi = {}
    if p.t_emb_off >= 0:
        epi = dict(add=env.emb_all[:, p.t_emb_off:], add_rpg=S, add_ld=env.emb_all.stride(0))
    st2, gkw = _gn_producer(ops, g.n // T, T * S, p.t_w1.shape[-2], xs.device, level=2)
    if sh is None:
        h = ops.convt3(h, p.t_w1, p.t_b1, T, S, **epi, **gkw)
    else:
        h = sh.convt3(ops, buf, p.t_w1, p.t_b1, g, **epi, **gkw)
    ga, be, eps = p.t_gn2
    h = ops.groupnorm(h, None, ga, be, g.n, S, eps=eps, silu=True, out=mid, stats=st2, **gn_kw)
    epi = dict(res1=xs)
    if coef is not None:
        epi.update(coef=coef, coef_rpg=S)
    else:
        epi.update(c_acc=c_acc, c_res1=1.0)
    if sh is None:
        return ops.convt3(h, p.t_w2, p.t_b2, T, S, **epi)
    return sh.convt3(ops, buf, p.t_w2, p.t_b2, g, **epi)


def unet_resblock(env: Env, g: Geo, p: ResPack, x1, x2=None):
    """VideoResBlock of the U-Net: alpha * spatial + (1 - alpha) * temporal  (util.py:341-369)."""
    xs, st = res_spatial(env, g, p, x1, x2, out_stats_imgs=g.T)
    return res_temporal(env, g, p, xs, coef=env.coefs[p.mixer], xs_stats=st)


def vae_resblock(env: Env, g: Geo, p: ResPack, x):
    """VideoResBlock of the VAE decoder: alpha * temporal + (1 - alpha) * spatial (temporal_ae.py:79-80 - the
    opposite convention), i.e. xs + alpha * (time_stack residual)."""
    xs, st = res_spatial(env, g, p, x, None, out_stats_imgs=g.T)
    return res_temporal(env, g, p, xs, c_acc=p.alpha, xs_stats=st)
