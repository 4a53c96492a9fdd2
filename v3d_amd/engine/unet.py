"""VideoUNet forward as a flat sequence of C-ABI ops (reference call stack: SURVEY.md §3.3;
video_model.py:442-493, video_attention.py:230-301, attention.py:556-577).

One activation layout end to end — channels-last bf16 [(b t) * H*W, C] — so none of the reference's layout shuffles
exist (b c h w <-> b (h w) c, (b t) s c <-> (b s) t c, th.cat of the skip, F.interpolate).  Per evaluation:
  * 2 embedding MLPs + ONE GEMM for every ResBlock's emb projection + ONE GEMM for every block's (collapsed)
    1-token cross-attention + one kernel for every AlphaBlender's epilogue coefficients;
  * per VideoResBlock: 4 GroupNorm(+SiLU) pairs, 2 conv3x3 + 2 temporal 3-tap GEMMs (+1-2 skip GEMMs), with the
    emb add, residual and alpha-blend fused into GEMM epilogues;
  * per SpatialVideoTransformer: 1 GroupNorm, 6 LayerNorms, 13 GEMMs (bias / GEGLU / residual / cross-attn vector /
    alpha-blend in epilogues), one MFMA flash attention over H*W tokens, one temporal attention over T frames.
"""
from __future__ import annotations

import os

from typing import Optional

import torch

from ..ops import GemmCall, get_ops
from .blocks import Env, Geo, unet_resblock
from .packing import SVTPack, UNetPack, cross_attn_pack, round_up

BF, F32 = torch.bfloat16, torch.float32


def _cast_rows_bf16(ops, x2d: torch.Tensor, pad_to: int = 0) -> torch.Tensor:
    """fp32 [n, C] -> bf16 [n, Cpad] (zero padded) with the layout kernel (S = 1)."""
    n, C = x2d.shape
    return ops.nchw_to_nhwc_bf16(x2d.float().contiguous().reshape(n, C, 1), 1.0, max(pad_to, round_up(C, 8)))


def frame_pos_table(ops, p: SVTPack, B: int, frame_ids) -> torch.Tensor:
    """time_pos_embed(timestep_embedding(frame index)) per image, [B*T, C] fp32 (video_attention.py:266-276).
    Depends only on weights and the frame indices, so it is built once per (B, frames) and cached on the pack."""
    key = (B, tuple(frame_ids))
    tab = p.tables.get(key)
    if tab is None:
        dev = p.proj_in[0].device
        t = torch.tensor(list(frame_ids), dtype=F32, device=dev)
        te = ops.timestep_embedding(t, p.C, p.max_period)
        w0, b0, w2, b2 = p.tpe
        h = ops.silu_add(ops.linear(te, w0, b0, out_dtype=F32))
        tab = ops.linear(h, w2, b2, out_dtype=F32).repeat(B, 1).contiguous()
        p.tables[key] = tab
    return tab


def cross_attention(ops, x: torch.Tensor, pack, ctx_rows: torch.Tensor, n_ctx: int, heads: int) -> torch.Tensor:
    """x + attn2(norm2(x), context) for a context of 2 .. 32 tokens (BasicTransformerBlock / VideoTransformerBlock, attention.py:565-571,
    video_attention.py:127-133; CrossAttention.forward attention.py:286-349).  x [M, C] rows; ctx_rows [G * n_ctx, ctx_dim]: G contexts, each shared
    by M / G consecutive rows (one image of the spatial block; one whole sample - T frames - of the temporal block, whose context is the sample's
    frame-0 context).  The attention is v3d_attn_temporal's problem shape with other strides: up to 32 consecutive query rows x n_ctx keys x d_head 64 per
    (context, row block, head) - the matrix-core kernel of round 5 takes Tq != Tk and a zero key stride across row blocks as they come."""
    M, C = x.shape
    (ga, be, eps), wq, wkv, wo, bo = pack
    G = ctx_rows.shape[0] // n_ctx
    R = M // G
    assert G * R == M and 2 <= n_ctx <= 32, f"cross-attention: {M} rows / {G} contexts of {n_ctx} tokens (2..32 tokens)"
    QB = max(d for d in range(1, 33) if R % d == 0)                   # query rows per problem: the largest divisor of a context's row count <= 32
    n2 = ops.empty((M, C), ops.act_dtype, x.device)
    ops.layernorm(x, ga, be, n2, eps)
    q = ops.linear(n2, wq)
    kv = ops.linear(ctx_rows, wkv)                                    # [G * n_ctx, 2C]: k | v of every context token
    a = ops.empty((M, C), ops.act_dtype, x.device)
    blk = lambda t: t.view(G, R // QB, QB, C).permute(0, 2, 1, 3)     # [G, QB query rows, row blocks, C]
    tok = lambda t: t.view(G, n_ctx, 1, C).expand(G, n_ctx, R // QB, C)   # the same n_ctx keys for every row block: stride 0
    ops.attn_temporal(blk(q), tok(kv[:, :C]), tok(kv[:, C:]), blk(a), heads, 0.125)
    return ops.linear(a, wo, bo, res1=x)


def run_svt(env: Env, g: Geo, p: SVTPack, x_in: torch.Tensor) -> torch.Tensor:
    ops = env.ops
    n, S, C, T, B = g.n, g.S, p.C, g.T, g.B
    sh = env.shard
    ctx = env.ctx_all
    ctx_ld = ctx.stride(0) if ctx is not None else 0

    ga, be, eps = p.norm
    h = ops.groupnorm(x_in, None, ga, be, n, S, eps=eps, silu=False)
    x = ops.linear(h, *p.proj_in)

    # ---- spatial BasicTransformerBlock (attention.py:556-577) ----
    ga, be, eps = p.s_norm1
    # the 64x64 level: LayerNorm + q | k | v projection in ONE kernel (token rows normalised in registers, V^T written directly)
    ln_proj = (os.environ.get("V3D_LN_PROJ", "1") not in ("", "0") and p.s_wqkv_fused is not None and C in getattr(ops, "LN_PROJ_WIDTHS", ())
               and S % 128 == 0)
    n1 = None
    if ln_proj:
        qk, vT = ops.ln_proj(x, eps, p.s_wqkv_fused[0], p.s_wqkv_fused[1], 2 * C, S)
    else:
        n1 = ops.empty((n * S, C), ops.act_dtype, x.device)
        ops.layernorm(x, ga, be, n1, eps)
        qk = ops.linear(n1, p.s_wqk)
        # V^T[img] = Wv @ LN(x)[img]^T directly out of the projection: keys contiguous for the P.V MFMA, no transpose
        vT = ops.empty((n, C, S), ops.act_dtype, x.device)
        ops.gemm(GemmCall(A=p.s_wv, W=n1.view(n, S, C), out=vT, M=C, N=S, K=C, batch=n))
    a = ops.empty((n * S, C), ops.act_dtype, x.device)
    if os.environ.get("V3D_ATTN_FP8", "0") not in ("", "0") and hasattr(ops, "attn_spatial_fp8") and S % 16 == 0:
        # scene-config variant (BASELINE.json configs[4]): e4m3 q | k tiles and V^T slabs, QK^T and P.V on the K = 64 fp8 MFMA
        qk8, qk_scales = ops.quant_fp8_tiles(qk, n, S)
        v8, v_scale = ops.quant_fp8_slab(vT, p.heads)
        ops.attn_spatial_fp8(qk8, qk_scales, v8, v_scale, a, n, S, p.heads, 0.125)
    else:
        ops.attn_spatial(qk[:, :C], qk[:, C:], vT, a, n, S, p.heads, 0.125)
    multi = env.ctx_tok is not None                  # context of more than one token: the general cross-attention (no V3D / SVD config)
    if multi:
        x = ops.linear(a, p.s_wo[0], p.s_wo[1], res1=x)
        x = cross_attention(ops, x, cross_attn_pack(p, "s"), env.ctx_tok, env.n_ctx, p.heads)
    else:
        # x = attn1 + x, then attn2 (1 context token => a per-image vector, Appendix B-9) folded in the same epilogue
        x = ops.linear(a, p.s_wo[0], p.s_wo[1], res1=x, add=ctx[:, p.s_ctx_off:], add_rpg=S, add_ld=ctx_ld)
    # norm3 + ff: at the 64x64 level the LayerNorm runs inside the fused feed-forward (v3d_ln_ff_fused), else as its own launch
    x_s = feed_forward(ops, x, p.s_ff, res1=x, ln=p.s_norm3)

    # ---- temporal VideoTransformerBlock on x_mix = x + frame embedding (video_attention.py:109-140,286-289) ----
    frames = range(T) if sh is None else sh.local_frames
    table = frame_pos_table(ops, p, B, frames)
    ga, be, eps = p.t_norm_in
    x_mix = ops.empty((n * S, C), ops.act_dtype, x.device)
    nin = ops.empty((n * S, C), ops.act_dtype, x.device)
    ops.layernorm(x_s, ga, be, nin, eps, add=table, add_rpg=S, add_ld=C, xsum_out=x_mix)
    x_t = feed_forward(ops, nin, p.t_ff_in, res1=x_mix)
    ga, be, eps = p.t_norm1
    ta = ops.empty((B, T, S, C), ops.act_dtype, x.device)
    if sh is None and ln_proj:
        qkv = ops.ln_proj(x_t, eps, p.t_wqkv_fused[0], p.t_wqkv_fused[1], 3 * C, S)[0].view(B, T, S, 3 * C)
        ops.attn_temporal(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], ta, p.heads, 0.125)
    elif sh is None:
        n1 = ops.empty((n * S, C), ops.act_dtype, x.device) if n1 is None else n1
        ops.layernorm(x_t, ga, be, n1, eps)
        qkv = ops.linear(n1, p.t_wqkv).view(B, T, S, 3 * C)
        ops.attn_temporal(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], ta, p.heads, 0.125)
    else:
        n1 = ops.empty((n * S, C), ops.act_dtype, x.device) if n1 is None else n1
        ops.layernorm(x_t, ga, be, n1, eps)
        # frame-sharded: to_k | to_v first, their all-gather along frames goes out at once (asynchronously: RCCL's own stream on
        # the box) and the to_q GEMM runs while it is in flight
        kv_loc = ops.linear(n1, p.t_wqkv[C:]).view(B, T, S, 2 * C)
        kv, pending = sh.allgather_frames(kv_loc, async_op=True)              # [B, T_global, S, 2C]
        q = ops.linear(n1, p.t_wqkv[:C]).view(B, T, S, C)
        pending.wait()
        ops.attn_temporal(q, kv[..., :C], kv[..., C:], ta, p.heads, 0.125)
    # temporal attn2: context = frame-0 context of each sample (video_attention.py:249-253) -> per-sample vector
    # (rows n.. of ctx_all hold the frame-0 projections, one per sample)
    if multi:
        x_t = ops.linear(ta.view(n * S, C), p.t_wo[0], p.t_wo[1], res1=x_t)
        x_t = cross_attention(ops, x_t, cross_attn_pack(p, "t"), env.ctx0_tok, env.n_ctx, p.heads)
    else:
        x_t = ops.linear(ta.view(n * S, C), p.t_wo[0], p.t_wo[1], res1=x_t, add=ctx[n:, p.t_ctx_off:], add_rpg=S * T,
                         add_ld=ctx_ld)
    # AlphaBlender fused: alpha * x_s + (1 - alpha) * (ff(norm3(x_t)) + x_t)
    x = feed_forward(ops, x_t, p.t_ff, res1=x_t, res2=x_s, coef=env.coefs[p.mixer], coef_rpg=S, ln=p.t_norm3)
    return ops.linear(x, p.proj_out[0], p.proj_out[1], res1=x_in)


_FF_FUSED = os.environ.get("V3D_FF_FUSED", "1") not in ("", "0")
_LN_FF = os.environ.get("V3D_LN_FF", "1") not in ("", "0")      # A/B knob: 0 = LayerNorm as its own launch in front of v3d_ff_fused


def feed_forward(ops, xin, ff, *, res1, res2=None, coef=None, coef_rpg=0, ln=None):
    """FeedForward with GEGLU (attention.py:82-113): out = W2 (value * gelu(gate)) + b2 + residual(s).  At C = 320 (the 64x64 level)
    one fused kernel keeps the 4x-wide hidden tensor on the CU (v3d_ff_fused); elsewhere two v3d_gemm launches.
    ln = (gamma, beta, eps): `xin` is the input of the LayerNorm in front of the block (norm3) - normalised inside the fused kernel where it
    exists (v3d_ln_ff_fused, the rows are register-resident there anyway), by v3d_layernorm otherwise."""
    M, C = xin.shape
    kw = dict(res1=res1)
    if res2 is not None:
        kw.update(res2=res2, coef=coef, coef_rpg=coef_rpg)
    fused = _FF_FUSED and ff.w2_fused is not None and M % 128 == 0 and hasattr(ops, "ff_fused")
    if ln is not None:
        if fused and _LN_FF and ff.w1_ln_fused is not None and hasattr(ops, "ln_ff_fused"):
            out = ops.empty((M, C), ops.act_dtype, xin.device)
            return ops.ln_ff_fused(xin, ff.ln_eps, ff.w1_ln_fused, ff.b1_ln_fused, ff.w2_fused, ff.b2, out, **kw)
        ga, be, eps = ln
        normed = ops.empty((M, C), ops.act_dtype, xin.device)
        ops.layernorm(xin, ga, be, normed, eps)
        xin = normed
    if fused:
        out = ops.empty((M, C), ops.act_dtype, xin.device)
        return ops.ff_fused(xin, ff.w1_fused, ff.b1_fused, ff.w2_fused, ff.b2, out, **kw)
    f = ops.linear(xin, ff.w1, ff.b1, geglu=True)
    return ops.linear(f, ff.w2, ff.b2, **kw)


def run_unet(pk: UNetPack, x, scale, concat, timesteps, context, y, num_video_frames, image_only_indicator, shard=None,
             context_frame0=None):
    """x [n, C1, H, W] fp32 (n = cfg * B * T_local), optional per-image `scale` and channel-concat `concat`.
    `context_frame0` [B, ...]: context of each sample's global frame 0 (defaults to context[::T]; a frame-sharded
    caller passes it explicitly because frame 0 may live on another rank).
    Returns an [n, out_channels, H, W] fp32 view of the channels-last result."""
    ops = get_ops()
    n, _, H, W = x.shape
    ops.begin_evaluation(x.device)
    T = int(num_video_frames) if num_video_frames is not None else 1
    if shard is None:
        from ..dist import active_shard
        shard = active_shard()
    if shard is not None:
        T = shard.T_local
        if context_frame0 is None:
            context_frame0 = shard.context_frame0
        assert context_frame0 is not None, "frame-sharded evaluation needs the context of each sample's global frame 0 (FrameShard.activate)"
    assert n % T == 0, f"batch {n} is not a multiple of num_video_frames {T}"
    B = n // T
    dev = x.device
    mc = pk.model_channels

    # ---- timestep / label embeddings (video_model.py:455-461) ----
    te = ops.timestep_embedding(timesteps.reshape(-1).float().contiguous(), mc)
    w0, b0, w2, b2 = pk.time_embed
    e = ops.linear(ops.silu_add(ops.linear(te, w0, b0, out_dtype=F32)), w2, b2, out_dtype=F32)
    lab = None
    if pk.label_emb is not None:
        w0, b0, w2, b2 = pk.label_emb
        yb = _cast_rows_bf16(ops, y.reshape(n, -1), w0.shape[1])
        lab = ops.linear(ops.silu_add(ops.linear(yb, w0, b0, out_dtype=F32)), w2, b2, out_dtype=F32)
    semb = ops.silu_add(e, lab)                                             # SiLU(emb): input of every emb_layers
    emb_all = ops.linear(semb, pk.emb_w, pk.emb_b, out_dtype=F32)           # [n, sum Cout]
    assert context is not None and context.dim() == 3 and context.shape[0] == n and 1 <= context.shape[1] <= 32, \
        f"context must be [n={n}, N, context_dim] with 1 <= N <= 32 tokens per image (V3D / SVD condition on one), got {None if context is None else tuple(context.shape)}"
    n_ctx = context.shape[1]
    ctx_all = ctx_tok = ctx0_tok = None
    if n_ctx == 1:
        # the collapsed cross-attention (W_ov folded at pack time, engine/packing.py) is exact for ONE context token only
        c2 = context.reshape(n, -1)
        c0 = c2[::T] if context_frame0 is None else context_frame0.reshape(B, -1)   # time_context = context[::timesteps]
        cb = _cast_rows_bf16(ops, torch.cat([c2, c0.to(c2.dtype)], dim=0))
        ctx_all = ops.linear(cb, pk.ctx_w, pk.ctx_b, out_dtype=F32)            # [n + B, sum C]
    else:
        # general cross-attention: the tokens themselves go to every block (engine cross_attention); time_context = context[::timesteps]
        assert shard is None, "frame-sharded evaluation supports one context token per image"
        c0 = context[::T] if context_frame0 is None else context_frame0.reshape(B, n_ctx, -1)
        both = _cast_rows_bf16(ops, torch.cat([context.reshape(n * n_ctx, -1), c0.reshape(B * n_ctx, -1).to(context.dtype)], dim=0))
        ctx_tok, ctx0_tok = both[:n * n_ctx], both[n * n_ctx:]
    ioi = None
    # merge_strategy="learned_with_images" needs the indicator (AlphaBlender.get_alpha asserts it, util.py:352-354)
    assert not pk.uses_ioi or image_only_indicator is not None, "image_only_indicator is required by merge_strategy='learned_with_images'"
    if pk.uses_ioi and image_only_indicator is not None:
        ioi = image_only_indicator
        if shard is not None and ioi.numel() == B * shard.T_global:      # full-length indicator: keep this rank's frames
            ioi = ioi.reshape(B, shard.T_global)[:, shard.t0:shard.t0 + T]
        ioi = ioi.reshape(-1).float().contiguous()
        assert ioi.numel() == n, f"image_only_indicator has {ioi.numel()} entries for {n} images"
    coefs = ops.blend_coefs(pk.mix_alpha, pk.mix_kind, ioi, n)
    env = Env(ops=ops, emb_all=emb_all, ctx_all=ctx_all, coefs=coefs, shard=shard, ctx_tok=ctx_tok, ctx0_tok=ctx0_tok, n_ctx=n_ctx)

    # first convolution (8 input channels): the packed input leaves its assembly kernel already unfolded 3x3 and the convolution is ONE GEMM
    # with K = 96 (per tap K = 8 is below every MFMA kernel's granule: the implicit-GEMM launch ran on the generic kernel, 258 us)
    in_gemm = pk.conv_in_gemm is not None and x.dim() == 4 and os.environ.get("V3D_CONV_IO_GEMM", "1") != "0"
    if in_gemm:
        h = ops.pack_input_im2col3x3(x.float().contiguous(), scale, None if concat is None else concat.float().contiguous(), pk.conv_in_gemm[0].shape[-1])
    else:
        h = ops.pack_input(x.float().contiguous(), scale, None if concat is None else concat.float().contiguous(), pk.in_pad)
    g = Geo(n=n, B=B, T=T, H=H, W=W)

    def run_stage(items, h, g, skip=None):
        for kind, p in items:
            if kind == "conv_in":
                h = ops.linear(h, pk.conv_in_gemm[0], pk.conv_in_gemm[1]) if in_gemm else ops.conv3x3(h, p[0], p[1], g.n, g.H, g.W)
            elif kind == "res":
                h = unet_resblock(env, g, p, h, skip)
                skip = None
            elif kind == "svt":
                h = run_svt(env, g, p, h)
            elif kind == "down":
                h = ops.conv3x3(h, p[0], p[1], g.n, g.H, g.W, stride=2)
                g = Geo(n=g.n, B=g.B, T=g.T, H=(g.H + 1) // 2, W=(g.W + 1) // 2)
            elif kind == "up":
                h = ops.conv3x3(h, p[0], p[1], g.n, g.H, g.W, up=2)
                g = Geo(n=g.n, B=g.B, T=g.T, H=g.H * 2, W=g.W * 2)
            else:
                raise ValueError(kind)
        return h, g

    hs = []
    for items in pk.input_stages:
        h, g = run_stage(items, h, g)
        hs.append(h)
    h, g = run_stage(pk.middle, h, g)
    for items in pk.output_stages:
        h, g = run_stage(items, h, g, skip=hs.pop())      # th.cat([h, hs.pop()], 1) is consumed split, never built
    ga, be, eps = pk.out_norm
    h = ops.groupnorm(h, None, ga, be, g.n, g.S, eps=eps, silu=True)
    if pk.out_conv_taps is not None and os.environ.get("V3D_CONV_IO_GEMM", "1") != "0":
        # last convolution (4 output channels): ONE GEMM with the nine taps' 36 weight rows on the unshifted pixels, then the taps' products are
        # gathered per pixel (N = 4 is below every MFMA kernel's tile: the implicit-GEMM launch ran on the generic kernel, 162 us)
        y = ops.linear(h, pk.out_conv_taps[0], None, out_dtype=F32)
        out = ops.tapsum3x3(y, pk.out_conv_taps[1], g.n, g.H, g.W, pk.out_channels)         # [n*S, out_ch] fp32
    else:
        out = ops.conv3x3(h, pk.out_conv[0], pk.out_conv[1], g.n, g.H, g.W, out_dtype=F32)     # [n*S, out_ch] fp32
    return out.view(n, g.H, g.W, pk.out_channels).permute(0, 3, 1, 2)
