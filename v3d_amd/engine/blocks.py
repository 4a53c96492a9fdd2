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

from ..ops import GEMM_CONV3X3, GEMM_CONVT3, GemmCall, OpsBase
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
    # contexts of more than one token (general cross-attention; V3D / SVD condition on ONE token and use ctx_all instead):
    ctx_tok: Optional[torch.Tensor] = None     # bf16 [n * N, ctx_dim]: every image's N context tokens
    ctx0_tok: Optional[torch.Tensor] = None    # bf16 [B * N, ctx_dim]: frame-0 context of each sample (the temporal block's context)
    n_ctx: int = 1


# GroupNorm statistics gathered by the GEMM that produces the tensor (GemmCall.gn_stats: the v3 <GN> epilogue, or the stand-alone pass inside
# v3d_gemm for launches on other kernels) instead of a separate read of it: three of the four norms of a VideoResBlock read a tensor this
# module has just produced.  Measured (tools/gn_epi_bench.py, same-box A/B of bench.py): behind a 3x3 convolution the epilogue costs 0-5 us
# against an 11-24 us stand-alone pass; behind the 3-tap temporal convolution (a third of the main loop to hide behind) it costs 14-16 us -
# a loss below the 64x64 level - so only the 3x3 producers use it: 44 launches per evaluation fewer, +0.3 % end to end.
# V3D_GN_EPILOGUE=0 restores the separate v3d_groupnorm_stats launches, =2 also uses the epilogue of the temporal convolutions (A/B knob).
_GN_EPILOGUE = int(os.environ.get("V3D_GN_EPILOGUE", "1") or 0)


def _gn_producer(ops, n_stat, rps, cout, device, groups=32, level=1, imgs_per_stat=1, consumer_small=False):
    """(stats buffer, GemmCall keywords) for a GEMM whose [M, cout] output feeds a 32-group GroupNorm with `rps` rows per statistics group.
    consumer_small: that GroupNorm will run as ONE launch on its own (ops.groupnorm -> v3d_groupnorm_small: 8 x 8 level) - no sums wanted."""
    if consumer_small or _GN_EPILOGUE < level or cout % (2 * groups) or rps % 16:
        return None, {}
    st = ops.gn_stats_buffer(n_stat, device, groups, rps=rps, imgs_per_stat=imgs_per_stat)
    return st, dict(gn_stats=st, gn_rps=rps, gn_cpg=cout // groups)


# GroupNorm + SiLU in the operand path of the convolution that consumes it (v3d_gemm gn_in_table: the LDS-haloed kernels of conv.hip
# normalise their input tile on its way into LDS).  V3D_CONV_GN=0 restores "v3d_groupnorm_apply, then the convolution" everywhere (A/B knob).
_CONV_GN = os.environ.get("V3D_CONV_GN", "1") not in ("", "0")
_HALO8 = os.environ.get("V3D_CONV_HALO8", "0") not in ("", "0")


def _halo_level(g: "Geo") -> bool:
    """Levels the LDS-haloed 3x3 kernels run (conv.hip: W in {64, 32, 16}).  The kernel also exists for 8 x 8 images, three to a tile (round 6), and is
    correct there, but its 48 tiles need a stream-K hand-off over 5-6 blocks per tile: 133 us against ~115 us for the one-launch norm + split-K
    convolution, 12.49 vs 12.55 frames/s end to end (profiles/r06_conv_w8_ab.txt) - V3D_CONV_HALO8=1 switches it on."""
    return _CONV_GN and (g.W in (16, 32, 64) or (_HALO8 and g.W == 8 and g.H == 8 and g.n % 3 == 0))


def _small_norm(ops, g: "Geo", cout, imgs_per_stat=1) -> bool:
    """Will the GroupNorm of a [n*S, cout] tensor produced at this level run as one launch (v3d_groupnorm_small)?  2-D norms only below
    the haloed kernels' levels (there the norm rides the convolution's operand path); the 3-D norm wherever its statistics group fits."""
    if imgs_per_stat == 1 and _halo_level(g) and cout % 320 == 0:
        return False
    return ops._GN_SMALL and ops.groupnorm_small_supported(cout, 0, g.S, imgs_per_stat)


def conv3x3_gn(ops, x1, x2, norm, w, bias, g: "Geo", *, stats=None, out=None, **epi):
    """conv3x3(SiLU(GroupNorm(x1 | x2))): openaimodel.py:267-271 (in_layers) / 302-314 (out_layers), 2-D norm (one statistics group per image)."""
    ga, be, eps = norm
    n, S = g.n, g.S
    if stats is None and not _halo_level(g) and ops.groupnorm_small_fits(x1, x2, S, 1):
        h = ops.groupnorm(x1, x2, ga, be, n, S, eps=eps, silu=True)          # one launch (8 x 8 level)
        return ops.conv3x3(h, w, bias, n, g.H, g.W, out=out, **epi)
    table = ops.groupnorm_table(x1, x2, ga, be, n, S, eps=eps, stats=stats)
    N = w.shape[-2]
    K = x1.shape[-1] + (0 if x2 is None else x2.shape[-1])
    if _CONV_GN and (g.W != 8 or _halo_level(g)):          # (the library also takes 8 x 8 images since round 6: policy, see _halo_level)
        if out is None:
            out = ops.empty((n * S, N), ops.act_dtype, x1.device)
        call = GemmCall(A=x1, A2=x2, W=w, out=out, M=n * S, N=N, K=K, bias=bias, mode=GEMM_CONV3X3, Hin=g.H, Win=g.W, Hout=g.H, Wout=g.W,
                        gn_in=table, gn_in_rps=S, gn_in_silu=True, **epi)
        if ops.gemm_gn_in_supported(call):
            ops.gemm(call)
            return out
    h = ops.groupnorm(x1, x2, ga, be, n, S, eps=eps, silu=True, table=table)
    return ops.conv3x3(h, w, bias, n, g.H, g.W, out=out, **epi)


def res_spatial(env: Env, g: Geo, p: ResPack, x1: torch.Tensor, x2: Optional[torch.Tensor], *, eps_override=None, out_stats_imgs=0, out=None):
    """2-D ResBlock (openaimodel.py:338-364 / model.py:131-151) on channels-last input (x1 [| x2] concatenated
    on channels but never materialised).  Returns xs [n*S, cout]; with out_stats_imgs = k > 0 also the GroupNorm partial sums of xs over
    groups of k images (the 3-D norm of the time_stack that follows), gathered by the last convolution: (xs, stats)."""
    ops = env.ops
    S = g.S
    epi = {}
    if p.emb_off >= 0:
        epi = dict(add=env.emb_all[:, p.emb_off:], add_rpg=S, add_ld=env.emb_all.stride(0))
    cout = p.w1.shape[-2]
    st2, gkw = _gn_producer(ops, g.n, S, cout, x1.device, consumer_small=_small_norm(ops, g, cout))
    h = conv3x3_gn(ops, x1, x2, p.gn1, p.w1, p.b1, g, **epi, **gkw)
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
        return conv3x3_gn(ops, h, None, p.gn2, p.w2, p.b2, g, stats=st2, res1=skip, out=out)
    st, gkw = _gn_producer(ops, g.n // out_stats_imgs, out_stats_imgs * S, p.w2.shape[-2], x1.device, imgs_per_stat=out_stats_imgs,
                           consumer_small=env.shard is None and _small_norm(ops, g, p.w2.shape[-2], out_stats_imgs))
    return conv3x3_gn(ops, h, None, p.gn2, p.w2, p.b2, g, stats=st2, res1=skip, out=out, **gkw), st


def convt3_gn(ops, x, norm, w, bias, g: "Geo", *, stats=None, **epi):
    """conv_t(SiLU(GroupNorm3d(x))) of the time_stack (video_model.py:42-55): 3-D norm (one statistics group per sample), (3,1,1) convolution."""
    ga, be, eps = norm
    n, S, T = g.n, g.S, g.T
    N, K = w.shape[-2], x.shape[-1]
    if stats is None and ops.groupnorm_small_fits(x, None, S, T):
        h = ops.groupnorm(x, None, ga, be, n, S, eps=eps, silu=True, imgs_per_stat=T)      # one launch (8 x 8 level)
        return ops.convt3(h, w, bias, T, S, **epi)
    table = ops.groupnorm_table(x, None, ga, be, n, S, eps=eps, imgs_per_stat=T, stats=stats)
    # Policy (the library answers whether it CAN, this is whether it PAYS): with 3 taps per chunk the temporal kernel has two steps of MFMA
    # slots for its normalisation chain and 192 x 320 tiles only - measured (tools/conv_gn_bench.py, tools/op_times.py) it beats "apply, then
    # the v3 / v2 / split-K kernels" only where those tiles fill the CUs: the 64x64 level (768 tiles, x1.08-1.15); 384 tiles at 32x32 x0.85,
    # 192 at 16x16 x0.89, 48 at 8x8 x0.45.
    tiles = (n * S // 192) * (N // 320) if N % 320 == 0 else 0
    cus = getattr(ops, "cu_count", 256)
    pays = getattr(ops, "always_fuse", False) or (tiles > 0 and tiles * 10 >= -(-tiles // cus) * cus * 9)
    if _CONV_GN and pays:
        out = ops.empty((n * S, N), ops.act_dtype, x.device)
        call = GemmCall(A=x, W=w, out=out, M=n * S, N=N, K=K, bias=bias, mode=GEMM_CONVT3, T=T, S=S, tmin=0, tmax=T - 1,
                        gn_in=table, gn_in_rps=T * S, gn_in_silu=True, **epi)
        if ops.gemm_gn_in_supported(call):
            ops.gemm(call)
            return out
    h = ops.groupnorm(x, None, ga, be, n, S, eps=eps, silu=True, imgs_per_stat=T, table=table)
    return ops.convt3(h, w, bias, T, S, **epi)


def res_temporal(env: Env, g: Geo, p: ResPack, xs: torch.Tensor, *, coef=None, c_acc=1.0, xs_stats=None):
    """time_stack ResBlock(dims=3, kernel (3,1,1)) + blend (video_model.py:74-79 / temporal_ae.py:73-80):
    out = xs + c * (conv_t(GN3d+SiLU(conv_t(GN3d+SiLU(xs)) + emb_t)) + bias), c from the blend table or scalar."""
    ops = env.ops
    S, T = g.S, g.T
    sh = env.shard
    C = xs.shape[-1]
    epi1 = {}
    if p.t_emb_off >= 0:
        epi1 = dict(add=env.emb_all[:, p.t_emb_off:], add_rpg=S, add_ld=env.emb_all.stride(0))
    epi2 = dict(res1=xs)
    if coef is not None:
        epi2.update(coef=coef, coef_rpg=S)
    else:
        epi2.update(c_acc=c_acc, c_res1=1.0)
    st2, gkw = _gn_producer(ops, g.n // T, T * S, p.t_w1.shape[-2], xs.device, level=2, imgs_per_stat=T,
                            consumer_small=sh is None and _small_norm(ops, g, p.t_w1.shape[-2], T))
    if sh is None:
        h = convt3_gn(ops, xs, p.t_gn1, p.t_w1, p.t_b1, g, stats=xs_stats, **epi1, **gkw)
        return convt3_gn(ops, h, p.t_gn2, p.t_w2, p.t_b2, g, stats=st2, **epi2)
    # frame-sharded (dist.py): `xs` already stands in the middle of split-halo buffer 0 (res_spatial wrote it there).  Per norm +
    # convolution ONE grouped point-to-point call carries the RAW +-1 frame halos to the neighbours and this rank's fp64 (sum, sumsq)
    # table to every rank; the normalisation of the local frames and of the received halo frames then uses the all-rank statistics.
    B, Tl = g.B, sh.T_local
    buf0, mid0 = sh.halo_buffer(B, S, C, ops.act_dtype, xs.device, slot=0)
    buf1, mid1 = sh.halo_buffer(B, S, C, ops.act_dtype, xs.device, slot=1)
    buf2, mid2 = sh.halo_buffer(B, S, C, ops.act_dtype, xs.device, slot=2)
    if xs.data_ptr() != mid0.data_ptr():
        mid0.copy_(xs)                        # (callers that could not produce into the buffer)
        xs = mid0

    def normalise(raw_buf, raw_mid, norm, stats):
        ga, be, eps = norm
        table = ops.groupnorm_table(raw_mid, None, ga, be, g.n, S, eps=eps, imgs_per_stat=Tl, count_imgs=sh.T_global, stats=stats,
                                    stats_hook=lambda sums: sh.exchange_halo_and_sums(raw_buf, B, S, sums)[1])
        ops.groupnorm_apply(raw_mid, None, table, mid1, g.n, S, Tl, True)
        if not sh.first:                      # the neighbours' frames: one "image" per sample, normalised with that sample's table row
            ops.groupnorm_apply(raw_buf[:B * S], None, table, buf1[:B * S], B, S, 1, True)
        if not sh.last:
            ops.groupnorm_apply(raw_buf[(B + B * Tl) * S:], None, table, buf1[(B + B * Tl) * S:], B, S, 1, True)

    normalise(buf0, mid0, p.t_gn1, xs_stats)
    sh.convt3(ops, buf1, p.t_w1, p.t_b1, g, out=mid2, **epi1, **gkw)
    normalise(buf2, mid2, p.t_gn2, st2)
    return sh.convt3(ops, buf1, p.t_w2, p.t_b2, g, **epi2)


def _shard_out(env: Env, g: Geo, p: ResPack, dev):
    """Frame-sharded runs: the spatial half of a VideoResBlock produces straight into the middle of split-halo buffer 0."""
    if env.shard is None:
        return None
    return env.shard.halo_buffer(g.B, g.S, p.w2.shape[-2], env.ops.act_dtype, dev, slot=0)[1]


def unet_resblock(env: Env, g: Geo, p: ResPack, x1, x2=None):
    """VideoResBlock of the U-Net: alpha * spatial + (1 - alpha) * temporal  (util.py:341-369)."""
    xs, st = res_spatial(env, g, p, x1, x2, out_stats_imgs=g.T, out=_shard_out(env, g, p, x1.device))
    return res_temporal(env, g, p, xs, coef=env.coefs[p.mixer], xs_stats=st)


def vae_resblock(env: Env, g: Geo, p: ResPack, x):
    """VideoResBlock of the VAE decoder: alpha * temporal + (1 - alpha) * spatial (temporal_ae.py:79-80 - the
    opposite convention), i.e. xs + alpha * (time_stack residual)."""
    xs, st = res_spatial(env, g, p, x, None, out_stats_imgs=g.T, out=_shard_out(env, g, p, x.device))
    return res_temporal(env, g, p, xs, c_acc=p.alpha, xs_stats=st)
