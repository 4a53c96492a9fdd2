"""Forwards of the inner blocks as stand-alone calls (reference signatures, NCHW tensors in and out).

The networks (VideoUNet / VideoDecoder) run a flat op sequence over ALL their blocks - embeddings of every block in one GEMM, skip
concatenations never built, statistics handed from producer to consumer.  A caller who holds ONE block (a user of the reference's module
tree: `unet.input_blocks[4][0]`, `decoder.mid.attn_1`) gets the same block executors here, fed by a pack of that block alone:

  VideoResBlock (U-Net)        video_model.py:68-101     forward(x, emb, num_video_frames, image_only_indicator=None)
  SpatialVideoTransformer      video_attention.py:230-301 forward(x, context=None, time_context=None, timesteps=None, image_only_indicator=None)
  VideoResBlock (VAE)          temporal_ae.py:59-82      forward(x, temb, skip_video=False, timesteps=None)
  AttnBlock (VAE)              model.py:180-201          forward(x)

Packs are cached on the module and rebuilt when a parameter changes (data pointer / version).  Same kernels, same arithmetic as inside the
networks; what differs is only what the network amortises over blocks (one launch for all embeddings, producer-side statistics)."""
from __future__ import annotations

import torch

from ..ops import get_ops
from .blocks import Env, Geo, unet_resblock, vae_resblock
from .packing import _Collector, _bf, _f, pack_resblock, pack_svt, pack_vae_resblock, _pack_attnblock

F32 = torch.float32


def _signature(mod):
    return tuple((p.data_ptr(), p._version, str(p.device)) for p in mod.parameters())


def _cached(mod, build):
    sig = _signature(mod)
    hit = mod.__dict__.get("_v3d_standalone")
    if hit is None or hit[0] != sig:
        hit = (sig, build())
        object.__setattr__(mod, "_v3d_standalone", hit)
    return hit[1]


def _rows(ops, x):
    """NCHW -> channels-last bf16 rows [n * H * W, C]."""
    return ops.nchw_to_nhwc_bf16(x.float().contiguous(), 1.0, x.shape[1])


def _nchw(rows, n, H, W, like):
    return rows.view(n, H, W, -1).permute(0, 3, 1, 2).to(like.dtype)


def _mixer_tables(col, dev):
    return torch.tensor(col.alphas, dtype=F32, device=dev), torch.tensor(col.kinds, dtype=torch.int32, device=dev)


def _ioi(image_only_indicator, n, B, T, dev):
    if image_only_indicator is None:
        return torch.zeros(n, dtype=F32, device=dev)
    ioi = image_only_indicator.reshape(-1).float().contiguous().to(dev)
    assert ioi.numel() == n, f"image_only_indicator has {ioi.numel()} entries for {n} images ([B={B}, T={T}])"
    return ioi


def unet_video_resblock(rb, x, emb, num_video_frames, image_only_indicator=None):
    ops = get_ops()
    n, C, H, W = x.shape
    T = int(num_video_frames)
    assert n % T == 0, f"batch {n} is not a multiple of num_video_frames {T}"
    dev = x.device

    def build():
        col = _Collector()
        p = pack_resblock(rb, col)
        alpha, kind = _mixer_tables(col, dev)
        return p, _bf(torch.cat(col.emb_w, 0)), _f(torch.cat(col.emb_b, 0)), alpha, kind

    p, emb_w, emb_b, alpha, kind = _cached(rb, build)
    ops.begin_evaluation(dev)
    semb = ops.silu_add(emb.reshape(n, -1).float().contiguous())                 # emb_layers = SiLU -> Linear (openaimodel.py:294-300)
    env = Env(ops=ops, emb_all=ops.linear(semb, emb_w, emb_b, out_dtype=F32),
              coefs=ops.blend_coefs(alpha, kind, _ioi(image_only_indicator, n, n // T, T, dev), n))
    out = unet_resblock(env, Geo(n=n, B=n // T, T=T, H=H, W=W), p, _rows(ops, x))
    return _nchw(out, n, H, W, x)


def spatial_video_transformer(st, x, context=None, time_context=None, timesteps=None, image_only_indicator=None):
    from .unet import _cast_rows_bf16, run_svt
    ops = get_ops()
    n, C, H, W = x.shape
    T = int(timesteps) if timesteps is not None else 1
    assert n % T == 0, f"batch {n} is not a multiple of timesteps {T}"
    B = n // T
    dev = x.device
    assert context is not None and context.dim() == 3 and context.shape[0] == n and context.shape[1] == 1, \
        "context must be [n, 1, context_dim]: the collapsed cross-attention is exact for one context token per image (V3D / SVD)"
    assert time_context is None, "explicit time_context is not used with use_spatial_context=True (video_attention.py:262-270)"

    def build():
        col = _Collector()
        p = pack_svt(st, col)
        alpha, kind = _mixer_tables(col, dev)
        return p, _bf(torch.cat(col.ctx_w, 0)), _f(torch.cat(col.ctx_b, 0)), alpha, kind

    p, ctx_w, ctx_b, alpha, kind = _cached(st, build)
    ops.begin_evaluation(dev)
    c2 = context.reshape(n, -1)
    cb = _cast_rows_bf16(ops, torch.cat([c2, c2[::T]], dim=0))                   # time_context = context[::timesteps]
    env = Env(ops=ops, ctx_all=ops.linear(cb, ctx_w, ctx_b, out_dtype=F32),
              coefs=ops.blend_coefs(alpha, kind, _ioi(image_only_indicator, n, B, T, dev), n))
    out = run_svt(env, Geo(n=n, B=B, T=T, H=H, W=W), p, _rows(ops, x))
    return _nchw(out, n, H, W, x)


def vae_video_resblock(rb, x, temb=None, skip_video=False, timesteps=None):
    assert temb is None, "the SVD / V3D autoencoder has no timestep embedding (temb_channels = 0)"
    assert not skip_video, "skip_video decode is not used on the V3D path"
    ops = get_ops()
    n, C, H, W = x.shape
    T = int(timesteps) if timesteps is not None else getattr(rb, "timesteps", None)
    assert T, "timesteps (frames per sample) is required"
    assert n % T == 0
    p = _cached(rb, lambda: pack_vae_resblock(rb))
    ops.begin_evaluation(x.device)
    out = vae_resblock(Env(ops=ops), Geo(n=n, B=n // T, T=T, H=H, W=W), p, _rows(ops, x))
    return _nchw(out, n, H, W, x)


def vae_attn_block(ab, x):
    from .vae import run_vae_attn
    ops = get_ops()
    n, C, H, W = x.shape
    p = _cached(ab, lambda: _pack_attnblock(ab))
    ops.begin_evaluation(x.device)
    out = run_vae_attn(Env(ops=ops), Geo(n=n, B=n, T=1, H=H, W=W), p, _rows(ops, x))
    return _nchw(out, n, H, W, x)


# ---- the smaller owners of the spatial transformer, as plain op sequences (inside the networks their arithmetic is fused across module
#      boundaries - LayerNorm into projections, the feed-forward in one kernel, the 1-token cross-attention folded into a vector) ------------
def _tokens(ops, x):
    """[..., C] -> bf16 rows [M, C] (zero-copy when x already is a contiguous bf16 row matrix)."""
    from .unet import _cast_rows_bf16
    return _cast_rows_bf16(ops, x.reshape(-1, x.shape[-1]))


def feed_forward_module(ff, x):
    """FeedForward.forward (attention.py:82-113): Linear(value | gate) -> value * gelu(gate) -> Linear, any leading dims."""
    from .packing import _pack_ff
    ops = get_ops()
    p = _cached(ff, lambda: _pack_ff(ff))
    rows = _tokens(ops, x)
    h = ops.linear(rows, p.w1, p.b1, geglu=True)
    return ops.linear(h, p.w2, p.b2).view(*x.shape[:-1], -1).to(x.dtype)


def cross_attention_module(at, x, context=None, mask=None):
    """CrossAttention.forward (attention.py:286-349) for the two cases V3D / SVD use: self-attention over x [B, N, C] (heads of 64), and
    cross-attention to ONE context token per batch element (softmax over a single key is 1: out = to_out(to_v(context)), exact)."""
    from ..ops import GemmCall
    from .packing import pack_linear
    assert mask is None, "attention masks are not used by V3D / SVD"
    ops = get_ops()
    B, N, C = x.shape
    dev = x.device

    def build():
        wq, wk, wv = (_bf(m.weight) for m in (at.to_q, at.to_k, at.to_v))
        return wq, wk, wv, pack_linear(at.to_out[0])

    wq, wk, wv, wo = _cached(at, build)
    inner = wq.shape[0]
    if context is not None:
        if context.dim() != 3 or context.shape[0] != B or context.shape[1] != 1:
            raise NotImplementedError("CrossAttention with more than one context token per batch element is not used by V3D / SVD")
        v = ops.linear(_tokens(ops, context), wv)                                   # [B, inner]
        o = ops.linear(v, wo[0], wo[1], out_dtype=F32)                              # [B, C]
        return o[:, None, :].expand(B, N, o.shape[-1]).to(x.dtype).contiguous()
    assert at.dim_head == 64, "attention kernels are specialised for dim_head = 64"
    rows = _tokens(ops, x)
    q, k = ops.linear(rows, wq), ops.linear(rows, wk)
    vT = ops.empty((B, inner, N), ops.act_dtype, dev)
    ops.gemm(GemmCall(A=wv, W=rows.view(B, N, C), out=vT, M=inner, N=N, K=C, batch=B))
    a = ops.empty((B * N, inner), ops.act_dtype, dev)
    ops.attn_spatial(q, k, vT, a, B, N, at.heads, float(at.scale))
    return ops.linear(a, wo[0], wo[1]).view(B, N, -1).to(x.dtype)


def basic_transformer_block(blk, x, context=None):
    """BasicTransformerBlock._forward (attention.py:556-577): x + attn1(norm1 x); + attn2(norm2 x, context); + ff(norm3 x).  x [B, N, C]."""
    from .packing import pack_norm
    ops = get_ops()
    B, N, C = x.shape
    cur = _tokens(ops, x)

    def ln(norm, rows):
        ga, be, eps = pack_norm(norm)
        out = ops.empty(tuple(rows.shape), ops.act_dtype, rows.device)
        ops.layernorm(rows, ga, be, out, eps)
        return out

    cur = (cross_attention_module(blk.attn1, ln(blk.norm1, cur).view(B, N, C)).reshape(B * N, C).float() + cur.float())
    cur = _tokens(ops, cur)
    cur = (cross_attention_module(blk.attn2, ln(blk.norm2, cur).view(B, N, C), context).reshape(B * N, C).float() + cur.float())
    cur = _tokens(ops, cur)
    cur = feed_forward_module(blk.ff, ln(blk.norm3, cur)).float() + cur.float()
    return cur.view(B, N, C).to(x.dtype)
