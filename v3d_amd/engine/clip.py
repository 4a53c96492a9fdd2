"""OpenCLIP ViT image tower as a flat sequence of C-ABI ops (the conditioning front-end).

Reference: sgm/modules/encoders/modules.py:594-752 (FrozenOpenCLIPImageEmbedder: preprocess + `model.visual(img)`); the tower
itself is open_clip's `VisionTransformer` (third party, `open-clip-torch` in requirements.txt, not in the reference tree):
conv1 (patch embed, no bias) -> [class token ; patches] + positional embedding -> ln_pre -> L x {x + MHA(ln_1 x); x + MLP(ln_2 x)}
-> ln_post on the class token -> @ proj.  Runs once per sample, outside the timed hot path, on the hot path's kernels:
`v3d_clip_preprocess` (resize + normalise + patch unfold in one kernel), `v3d_gemm`, `v3d_layernorm`, `v3d_softmax_rows`,
`v3d_gelu_bf16`, `v3d_copy2d_bf16`.

Layout: one row per token, L = tokens rounded up to a multiple of 8 rows per image (the padding rows stay finite and are masked
out of every softmax through the GEMM's per-column `add` vector).  Head dim 80 (ViT-H) does not fit the d = 64 streamed-softmax
kernels of the U-Net, and 257 tokens x 16 heads is tiny: attention is the unfused form the VAE AttnBlock uses - batched
q k^T into fp32 (batch = heads), row softmax, batched P V with V^T produced directly by a swapped GEMM.  The 1/sqrt(d) scale is
folded into the packed q projection, the value bias into the out-projection bias (softmax rows sum to 1).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List

import torch

from ..ops import GemmCall, get_ops
from .packing import _bf, _f, round_up

F32 = torch.float32


@dataclass
class ClipBlockPack:
    ln1: tuple
    wqk: torch.Tensor      # [2 W, W], q rows pre-scaled by d^-1/2
    bqk: torch.Tensor
    wv: torch.Tensor       # [W, W] (A operand of the swapped GEMM that produces V^T)
    wo: torch.Tensor
    bo: torch.Tensor       # out_proj.bias + out_proj.weight @ v_bias
    ln2: tuple
    wfc: torch.Tensor
    bfc: torch.Tensor
    wproj: torch.Tensor
    bproj: torch.Tensor


@dataclass
class ClipPack:
    width: int
    heads: int
    patch: int
    image_size: int
    tokens: int            # 1 + (image_size / patch)^2
    L: int                 # token rows per image (tokens rounded up to 8)
    kpad: int
    embed_dim: int
    wpatch: torch.Tensor   # [width, kpad]
    cls_row: torch.Tensor  # [1, width] = class_embedding + positional_embedding[0]
    pos_rows: torch.Tensor  # [tokens - 1, width]
    ln_pre: tuple
    ln_post: tuple
    wout: torch.Tensor     # proj^T [embed_dim, width]
    mask: torch.Tensor     # [L] fp32: 0 on real tokens, -1e30 on padding columns
    blocks: List[ClipBlockPack] = field(default_factory=list)


def _ln(n):
    return _f(n.weight), _f(n.bias), float(n.eps)


def pack_clip_visual(vis) -> ClipPack:
    """vis: v3d_amd.sgm.modules.encoders.open_clip_vit.VisionTransformer (open_clip parameter names)."""
    W, heads = vis.width, vis.heads
    d = W // heads
    assert d % 8 == 0, "head dim must be a multiple of 8 (16-byte rows)"
    P = vis.patch_size
    g = vis.image_size // P
    tokens = 1 + g * g
    L = round_up(tokens, 8)
    kpad = round_up(3 * P * P, 32)
    dev = vis.conv1.weight.device
    wp = torch.zeros(W, kpad, dtype=F32, device=dev)
    wp[:, :3 * P * P] = vis.conv1.weight.detach().float().reshape(W, 3 * P * P)
    pos = vis.positional_embedding.detach().float()
    mask = torch.zeros(L, dtype=F32, device=dev)
    mask[tokens:] = -1e30
    pk = ClipPack(width=W, heads=heads, patch=P, image_size=vis.image_size, tokens=tokens, L=L, kpad=kpad, embed_dim=vis.proj.shape[1],
                  wpatch=_bf(wp), cls_row=_bf((vis.class_embedding.detach().float() + pos[0])[None]), pos_rows=_bf(pos[1:]),
                  ln_pre=_ln(vis.ln_pre), ln_post=_ln(vis.ln_post), wout=_bf(vis.proj.detach().float().t()), mask=mask)
    sc = float(d) ** -0.5
    for blk in vis.transformer.resblocks:
        w_in, b_in = blk.attn.in_proj_weight.detach().float(), blk.attn.in_proj_bias.detach().float()
        wq, wk, wv = w_in[:W], w_in[W:2 * W], w_in[2 * W:]
        bq, bk, bv = b_in[:W], b_in[W:2 * W], b_in[2 * W:]
        wo, bo = blk.attn.out_proj.weight.detach().float(), blk.attn.out_proj.bias.detach().float()
        pk.blocks.append(ClipBlockPack(
            ln1=_ln(blk.ln_1), wqk=_bf(torch.cat([wq * sc, wk], 0)), bqk=_f(torch.cat([bq * sc, bk], 0)), wv=_bf(wv), wo=_bf(wo),
            bo=_f(bo + wo @ bv), ln2=_ln(blk.ln_2), wfc=_bf(blk.mlp.c_fc.weight), bfc=_f(blk.mlp.c_fc.bias),
            wproj=_bf(blk.mlp.c_proj.weight), bproj=_f(blk.mlp.c_proj.bias)))
    return pk


def _attention(ops, pk: ClipPack, bp: ClipBlockPack, xn: torch.Tensor, x: torch.Tensor, B: int) -> torch.Tensor:
    """x + out_proj(MHA(xn)) for B images of L token rows."""
    W, H, L = pk.width, pk.heads, pk.L
    d = W // H
    dev = xn.device
    qk = ops.linear(xn, bp.wqk, bp.bqk)                                   # [B L, 2 W]
    o = ops.empty((B * L, W), None, dev)
    for b in range(B):
        rows = slice(b * L, (b + 1) * L)
        vT = ops.empty((W, L), None, dev)                                 # V^T (without bias): row = channel (head-major), col = token
        ops.gemm(GemmCall(A=bp.wv, W=xn[rows], out=vT, M=W, N=L, K=W))
        q = qk[rows, :W].view(L, H, d).permute(1, 0, 2)                   # [H, L, d] views of the fused projection
        k = qk[rows, W:].view(L, H, d).permute(1, 0, 2)
        scores = ops.empty((H, L, L), F32, dev)
        ops.gemm(GemmCall(A=q, W=k, out=scores, M=L, N=L, K=d, batch=H, add=pk.mask, add_rpg=L, add_ld=0))
        prob = ops.empty((H, L, L), None, dev)
        ops.softmax_rows(scores, prob)
        ops.gemm(GemmCall(A=prob, W=vT.view(H, d, L), out=o[rows].view(L, H, d).permute(1, 0, 2), M=L, N=d, K=L, batch=H))
    return ops.linear(o, bp.wo, bp.bo, res1=x)


@torch.no_grad()
def run_clip_visual(pk: ClipPack, img: torch.Tensor, antialias: bool, mean, std) -> torch.Tensor:
    """img [B, 3, H, W] in [-1, 1] -> image embedding [B, embed_dim] fp32."""
    ops = get_ops()
    B = img.shape[0]
    W, L, T = pk.width, pk.L, pk.tokens
    dev = img.device
    patches = ops.clip_preprocess(img.float().contiguous(), pk.image_size, pk.patch, antialias, mean, std, pk.kpad)   # [B (T-1), kpad]
    x = torch.zeros(B * L, W, dtype=ops.act_dtype, device=dev)           # padding rows stay zero until the first LayerNorm
    for b in range(B):
        ops.copy2d_bf16(pk.cls_row, x[b * L:b * L + 1])
        ops.gemm(GemmCall(A=patches[b * (T - 1):(b + 1) * (T - 1)], W=pk.wpatch, out=x[b * L + 1:b * L + T], M=T - 1, N=W, K=pk.kpad,
                          res1=pk.pos_rows))
    h = ops.empty((B * L, W), None, dev)
    ops.layernorm(x, pk.ln_pre[0], pk.ln_pre[1], h, pk.ln_pre[2])
    x = h
    for bp in pk.blocks:
        xn = ops.empty((B * L, W), None, dev)
        ops.layernorm(x, bp.ln1[0], bp.ln1[1], xn, bp.ln1[2])
        x = _attention(ops, pk, bp, xn, x, B)
        xn = ops.empty((B * L, W), None, dev)
        ops.layernorm(x, bp.ln2[0], bp.ln2[1], xn, bp.ln2[2])
        f = ops.gelu(ops.linear(xn, bp.wfc, bp.bfc))
        x = ops.linear(f, bp.wproj, bp.bproj, res1=x)
    pooled = ops.empty((B * L, W), None, dev)
    ops.layernorm(x, pk.ln_post[0], pk.ln_post[1], pooled, pk.ln_post[2])
    out = ops.empty((B, pk.embed_dim), F32, dev)
    ops.gemm(GemmCall(A=pooled.view(B, L, W)[:, 0], W=pk.wout, out=out, M=B, N=pk.embed_dim, K=W))
    return out
