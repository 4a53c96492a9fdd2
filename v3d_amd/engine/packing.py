"""Weight packing: nn.Module parameter owners (reference state-dict layout, fp32) -> kernel-ready device tensors.

Layouts (include/v3d_hip.h): linear W [N][K] bf16; conv3x3 W [9][Cout][Cin] bf16 (tap = ky*3+kx); temporal conv
W [3][Cout][Cin]; GEGLU projections with value/gate rows interleaved in groups of 16 so both halves of a pair land
in the same lane of the MFMA epilogue; norm affine / biases fp32.  Weight-only algebra done here once:
  * cross-attention to a single context token has softmax == 1, so attn2(x) = to_out(to_v(ctx)) for every query
    (reference: attention.py:517-524,570-575): W_ov = to_out.W @ to_v.W is folded at pack
    time and ALL blocks' W_ov are concatenated into one [sum C, context_dim] matrix (one GEMM per U-Net evaluation);
  * every ResBlock's emb_layers Linear (spatial and temporal) is concatenated the same way (one GEMM per evaluation).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch
import torch.nn as nn

BF, F32 = torch.bfloat16, torch.float32


def _bf(t: torch.Tensor) -> torch.Tensor:
    """Packed-weight storage: bf16 for the HIP kernels (the test emulator may ask for exact fp32)."""
    from ..ops import get_ops
    return t.detach().to(get_ops().act_dtype).contiguous()


def _f(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(F32).contiguous()


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def pack_linear(lin: nn.Linear, k_pad: int = 0):
    w = lin.weight.detach()
    if k_pad and w.shape[1] < k_pad:
        w = torch.cat([w, w.new_zeros(w.shape[0], k_pad - w.shape[1])], dim=1)
    return _bf(w), (None if lin.bias is None else _f(lin.bias))


def pack_conv1x1(conv: nn.Module):
    w = conv.weight.detach()
    return _bf(w.reshape(w.shape[0], w.shape[1])), (None if conv.bias is None else _f(conv.bias))


def pack_conv3x3(conv: nn.Conv2d):
    w = conv.weight.detach()                      # [O, I, 3, 3]
    O, I = w.shape[0], w.shape[1]
    Ip = round_up(I, 8)
    if Ip != I:
        w = torch.cat([w, w.new_zeros(O, Ip - I, 3, 3)], dim=1)
    return _bf(w.permute(2, 3, 0, 1).reshape(9, O, Ip)), (None if conv.bias is None else _f(conv.bias))


def pack_conv3x3_im2col(conv: nn.Conv2d, Kpad=96):
    """First convolution of the U-Net (8 input channels) as ONE GEMM over the unfolded input (v3d_pack_input_im2col3x3):
    W[o][tap * 8 + c], zero-padded to Kpad columns."""
    w = conv.weight.detach()                      # [O, I, 3, 3]
    O, I = w.shape[0], w.shape[1]
    assert I <= 8 and Kpad >= 72
    wp = w.new_zeros(O, Kpad)
    wp.view(O, Kpad)[:, :72].view(O, 9, 8)[:, :, :I] = w.permute(0, 2, 3, 1).reshape(O, 9, I)
    return _bf(wp), (None if conv.bias is None else _f(conv.bias))


def pack_conv3x3_taps(conv: nn.Conv2d, Npad=64):
    """Last convolution of the U-Net (4 output channels) as GEMM + gather (v3d_tapsum3x3): rows tap * O + o of ONE [Npad, I] weight."""
    w = conv.weight.detach()                      # [O, I, 3, 3]
    O, I = w.shape[0], w.shape[1]
    assert 9 * O <= Npad
    wp = w.new_zeros(Npad, I)
    wp[:9 * O] = w.permute(2, 3, 0, 1).reshape(9 * O, I)
    return _bf(wp), (None if conv.bias is None else _f(conv.bias))


def pack_convt3(conv: nn.Conv3d):
    w = conv.weight.detach()                      # [O, I, 3, 1, 1]
    O, I = w.shape[0], w.shape[1]
    return _bf(w.reshape(O, I, 3).permute(2, 0, 1)), (None if conv.bias is None else _f(conv.bias))


def pack_geglu(lin: nn.Linear):
    """[2*inner, dim] (value rows then gate rows) -> rows interleaved as 16 value / 16 gate / 16 value / ..."""
    w, b = lin.weight.detach(), lin.bias.detach()
    inner = w.shape[0] // 2
    assert inner % 16 == 0
    wv, wg = w[:inner].reshape(inner // 16, 16, -1), w[inner:].reshape(inner // 16, 16, -1)
    wp = torch.stack([wv, wg], dim=1).reshape(2 * inner, -1)
    bp = torch.stack([b[:inner].reshape(-1, 16), b[inner:].reshape(-1, 16)], dim=1).reshape(2 * inner)
    return _bf(wp), _f(bp)


def pack_norm(n: nn.Module):
    return _f(n.weight), _f(n.bias), float(n.eps)


@dataclass
class ResPack:
    cin: int
    cout: int
    split: Optional[tuple]          # (C_h, C_skip) when the input is the never-materialised skip concat
    gn1: tuple = None
    w1: torch.Tensor = None
    b1: torch.Tensor = None
    gn2: tuple = None
    w2: torch.Tensor = None
    b2: torch.Tensor = None
    skip_w: Optional[torch.Tensor] = None
    skip_b: Optional[torch.Tensor] = None
    emb_off: int = -1                # column of this block's spatial emb projection in emb_all (-1: none)
    t_gn1: tuple = None
    t_w1: torch.Tensor = None
    t_b1: torch.Tensor = None
    t_gn2: tuple = None
    t_w2: torch.Tensor = None
    t_b2: torch.Tensor = None
    t_emb_off: int = -1
    mixer: int = -1                  # row in the blend-coefficient table (U-Net)
    alpha: float = 0.0               # host alpha (VAE blocks: scalar epilogue coefficient)


@dataclass
class FFPack:
    w1: torch.Tensor
    b1: torch.Tensor
    w2: torch.Tensor
    b2: torch.Tensor
    # v3d_ff_fused operands (C = 320 only), see ff_fused_pack
    w1_fused: Optional[torch.Tensor] = None
    b1_fused: Optional[torch.Tensor] = None
    w2_fused: Optional[torch.Tensor] = None
    # v3d_ln_ff_fused: the LayerNorm in front of the block folded into the first Linear (W1 diag(gamma), b1 + W1 beta), then the same packing
    w1_ln_fused: Optional[torch.Tensor] = None
    b1_ln_fused: Optional[torch.Tensor] = None
    ln_eps: float = 0.0


def ff_fused_row_order(hidden: int) -> torch.Tensor:
    """Row order of W1p / b1 inside v3d_ff_fused (include/v3d_hip.h), as indices into the Linear's [value rows ; gate rows]:
    row 64 s + 32 a + 8 g + 4 h + c holds the (g odd ? gate : value) row of hidden channel 32 s + 16 a + 8 (g >> 1) + 4 h + c -
    the order in which the rows of a 32x32 MFMA tile land in a lane's accumulator registers."""
    s_, a, g, h, c = torch.meshgrid(torch.arange(hidden // 32), torch.arange(2), torch.arange(4), torch.arange(2), torch.arange(4),
                                    indexing="ij")
    return ((g & 1) * hidden + 32 * s_ + 16 * a + 8 * (g >> 1) + 4 * h + c).reshape(-1)


def ff_fused_k_perm(hidden: int) -> torch.Tensor:
    """Column order of W2p inside v3d_ff_fused: position 32 s + 16 a + 8 h + 4 t + c holds hidden channel 32 s + 16 a + 8 t + 4 h + c
    - the order in which a lane holds its 8 GEGLU outputs after the first MFMA stage."""
    s_, a, h, t, c = torch.meshgrid(torch.arange(hidden // 32), torch.arange(2), torch.arange(2), torch.arange(2), torch.arange(4),
                                    indexing="ij")
    return (32 * s_ + 16 * a + 8 * t + 4 * h + c).reshape(-1)


def ff_dma_tile_index(rows: int, cols: int, slab_rows: int, slab_cols: int) -> torch.Tensor:
    """Flat indices (into a row-major [rows, cols] matrix) in the order v3d_ff_fused's weight stream reads them: the matrix is cut
    into slabs of slab_rows x slab_cols, a slab into 1-KiB LDS-DMA pieces of 16 rows x 32 columns (64 B per row) ordered
    (column block, row block), and inside a piece lane l = 4 * row + p carries the 8 columns of chunk p ^ swz(row) - the XOR swizzle
    of the LDS image (ff.hip ff_swz) applied at pack time, so that a piece is 1 KiB of CONTIGUOUS memory (one buffer_load ... lds of
    64 lanes x 16 B).  Exactly one of slab_rows / slab_cols differs from the full extent."""
    assert rows % slab_rows == 0 and cols % slab_cols == 0 and slab_rows % 16 == 0 and slab_cols % 32 == 0
    idx = []
    swz = torch.tensor([0, 2, 3, 1])
    r16, pp, e = torch.meshgrid(torch.arange(16), torch.arange(4), torch.arange(8), indexing="ij")
    chunk = pp ^ swz[(r16 >> 2) & 3]
    piece = r16 * cols + chunk * 8 + e                                           # [16, 4, 8] offsets inside a piece's 16 x 32 window
    for sr in range(rows // slab_rows):
        for sc in range(cols // slab_cols):
            for cb in range(slab_cols // 32):
                for rb in range(slab_rows // 16):
                    idx.append(((sr * slab_rows + rb * 16) * cols + sc * slab_cols + cb * 32) + piece.reshape(-1))
    return torch.cat(idx)


def ff_fused_pack(w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, ln=None):
    """Linear(C, 2 hidden).weight / .bias and Linear(hidden, C).weight -> (W1p, b1p, W2p) of v3d_ff_fused: rows / columns in the
    MFMA hand-over order (ff_fused_row_order / ff_fused_k_perm), then tiled into the DMA pieces of the kernel's weight stream.
    ln = (gamma, beta) of a LayerNorm in front of the block (v3d_ln_ff_fused): LN(x) W1^T + b1 = xh (W1 diag(gamma))^T + (b1 + W1 beta)."""
    hidden = w2.shape[-1]
    C = w1.shape[1]
    if ln is not None:
        w1f = w1.detach().float()
        b1 = b1.detach().float() + w1f @ ln[1].detach().float()
        w1 = w1f * ln[0].detach().float()[None, :]
    ro = ff_fused_row_order(hidden).to(w1.device)
    w1p = w1[ro]                                                                  # [2 hidden, C]: slabs of 64 rows
    w2p = w2.reshape(-1, hidden)[:, ff_fused_k_perm(hidden).to(w2.device)]        # [C, hidden]:   slabs of 32 columns
    t1 = ff_dma_tile_index(2 * hidden, C, 64, C).to(w1.device)
    t2 = ff_dma_tile_index(C, hidden, C, 32).to(w2.device)
    return (_bf(w1p.reshape(-1)[t1].reshape(2 * hidden, C)), _f(b1[ro]), _bf(w2p.reshape(-1)[t2].reshape(C, hidden)))


def ln_proj_pack(w: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor):
    """[N, C] concatenated q ; k ; v weight + the LayerNorm affine in front of it -> (Wp, bias) of v3d_ln_proj: LN(x) W^T = xh (W diag(gamma))^T + W beta
    with xh = (x - mean) * rstd, Wp in the kernel's LDS-DMA piece order (slabs of 64 rows, see ff_dma_tile_index), bias fp32."""
    N, C = w.shape
    wf = w.detach().float()
    wg = wf * gamma.detach().float()[None, :]
    return (_bf(wg.reshape(-1)[ff_dma_tile_index(N, C, 64, C).to(w.device)].reshape(N, C)), _f(wf @ beta.detach().float()))


@dataclass
class SVTPack:
    C: int
    heads: int
    norm: tuple
    proj_in: tuple
    proj_out: tuple
    s_norm1: tuple = None
    s_wqk: torch.Tensor = None
    s_wv: torch.Tensor = None
    s_wo: tuple = None
    s_ctx_off: int = 0
    s_norm3: tuple = None
    s_ff: FFPack = None
    t_norm_in: tuple = None
    t_ff_in: FFPack = None
    t_norm1: tuple = None
    t_wqkv: torch.Tensor = None
    s_wqkv_fused: Optional[tuple] = None     # C = 320 only: (DMA-tiled [3C, C] weight with norm1's gamma folded in, bias W beta) for v3d_ln_proj
    t_wqkv_fused: Optional[tuple] = None
    t_wo: tuple = None
    t_ctx_off: int = 0
    # general cross-attention (context of 2 .. 32 tokens: attention.py:286-349 without the one-token collapse): module references, packed on
    # first use by cross_attn_pack (every V3D / SVD configuration conditions on ONE token and never builds these)
    s_attn2: object = None
    s_norm2: object = None
    t_attn2: object = None
    t_norm2: object = None
    x2: Dict = field(default_factory=dict)       # "s" / "t" -> (norm2, wq, wkv, wo, bo)
    t_norm3: tuple = None
    t_ff: FFPack = None
    tpe: tuple = None                # time_pos_embed (w0, b0, w2, b2)
    max_period: float = 10000.0
    mixer: int = -1
    tables: Dict = field(default_factory=dict)   # (B, tuple(frame ids)) -> [B*T, C] fp32 frame-position table


@dataclass
class UNetPack:
    device: torch.device
    in_channels: int
    in_pad: int
    model_channels: int
    out_channels: int
    context_dim: int
    adm_in: Optional[int]
    time_embed: tuple
    label_emb: Optional[tuple]
    conv_in: tuple
    input_stages: List[list]
    middle: list
    output_stages: List[list]
    out_norm: tuple
    out_conv: tuple
    emb_w: torch.Tensor
    emb_b: torch.Tensor
    ctx_w: torch.Tensor
    ctx_b: torch.Tensor
    mix_alpha: torch.Tensor
    mix_kind: torch.Tensor
    uses_ioi: bool
    conv_in_gemm: Optional[tuple] = None      # (W [model_channels, 96] over the unfolded 8-channel input, bias): v3d_pack_input_im2col3x3 + ONE GEMM
    out_conv_taps: Optional[tuple] = None     # (W [64, model_channels] rows tap * out_ch + o, bias): ONE GEMM + v3d_tapsum3x3


def _pack_ff(ff, norm=None) -> FFPack:
    """norm: the plain LayerNorm module whose output is this block's only input (norm3 of a transformer block) - folded for v3d_ln_ff_fused."""
    w1, b1 = pack_geglu(ff.net[0].proj)
    w2, b2 = pack_linear(ff.net[2])
    p = FFPack(w1, b1, w2, b2)
    C, hidden = w2.shape[-2], w2.shape[-1]
    if C == 320 and hidden % 64 == 0 and hidden >= 128:
        lin1, lin2 = ff.net[0].proj, ff.net[2]
        p.w1_fused, p.b1_fused, p.w2_fused = ff_fused_pack(lin1.weight.detach(), lin1.bias.detach(), lin2.weight.detach())
        if norm is not None:
            p.w1_ln_fused, p.b1_ln_fused, _ = ff_fused_pack(lin1.weight.detach(), lin1.bias.detach(), lin2.weight.detach(),
                                                             ln=(norm.weight, norm.bias))
            p.ln_eps = float(norm.eps)
    return p


class _Collector:
    """Accumulates the concatenated emb / ctx projection matrices and the mixer table while walking the net."""

    def __init__(self):
        self.emb_w, self.emb_b, self.emb_cols = [], [], 0
        self.ctx_w, self.ctx_b, self.ctx_cols = [], [], 0
        self.alphas, self.kinds = [], []

    def add_emb(self, lin: nn.Linear) -> int:
        off = self.emb_cols
        self.emb_w.append(lin.weight.detach().float())
        self.emb_b.append(lin.bias.detach().float())
        self.emb_cols += lin.weight.shape[0]
        return off

    def add_ctx(self, attn) -> int:
        off = self.ctx_cols
        wo, wv = attn.to_out[0].weight.detach().float(), attn.to_v.weight.detach().float()
        self.ctx_w.append(wo @ wv)
        self.ctx_b.append(attn.to_out[0].bias.detach().float())
        self.ctx_cols += wo.shape[0]
        return off

    def add_mixer(self, blender, kind: int) -> int:
        self.alphas.append(blender.alpha_value())
        self.kinds.append(kind)
        return len(self.alphas) - 1


def pack_resblock(rb, col: Optional[_Collector]) -> ResPack:
    """U-Net VideoResBlock (2-D ResBlock + (3,1,1) time_stack + AlphaBlender)."""
    p = ResPack(cin=rb.channels, cout=rb.out_channels, split=getattr(rb, "concat_split", None))
    p.gn1 = pack_norm(rb.in_layers[0])
    p.w1, p.b1 = pack_conv3x3(rb.in_layers[2])
    p.gn2 = pack_norm(rb.out_layers[0])
    p.w2, p.b2 = pack_conv3x3(rb.out_layers[3])
    if not isinstance(rb.skip_connection, nn.Identity):
        p.skip_w, p.skip_b = pack_conv1x1(rb.skip_connection)
    if rb.emb_layers is not None:
        p.emb_off = col.add_emb(rb.emb_layers[1])
    ts = rb.time_stack
    p.t_gn1 = pack_norm(ts.in_layers[0])
    p.t_w1, p.t_b1 = pack_convt3(ts.in_layers[2])
    p.t_gn2 = pack_norm(ts.out_layers[0])
    p.t_w2, p.t_b2 = pack_convt3(ts.out_layers[3])
    if ts.emb_layers is not None:
        p.t_emb_off = col.add_emb(ts.emb_layers[1])
    p.mixer = col.add_mixer(rb.time_mixer, 0)
    return p


def pack_svt(st, col: _Collector) -> SVTPack:
    assert len(st.transformer_blocks) == 1 and len(st.time_stack) == 1, "transformer_depth > 1 is not used by V3D/SVD"
    blk, tb = st.transformer_blocks[0], st.time_stack[0]
    C = st.proj_in.weight.shape[0]
    p = SVTPack(C=C, heads=st.n_heads, norm=pack_norm(st.norm), proj_in=pack_linear(st.proj_in), proj_out=pack_linear(st.proj_out))
    assert st.d_head == 64, "attention kernels are specialised for d_head = 64"
    p.s_norm1 = pack_norm(blk.norm1)
    p.s_wqk = _bf(torch.cat([blk.attn1.to_q.weight.detach(), blk.attn1.to_k.weight.detach()], dim=0))
    p.s_wv = _bf(blk.attn1.to_v.weight)
    if C == 320:     # the 64x64 level: LayerNorm + q | k | v projection in one kernel (v3d_ln_proj)
        p.s_wqkv_fused = ln_proj_pack(torch.cat([blk.attn1.to_q.weight, blk.attn1.to_k.weight, blk.attn1.to_v.weight], dim=0), blk.norm1.weight, blk.norm1.bias)
        p.t_wqkv_fused = ln_proj_pack(torch.cat([tb.attn1.to_q.weight, tb.attn1.to_k.weight, tb.attn1.to_v.weight], dim=0), tb.norm1.weight, tb.norm1.bias)
    p.s_wo = pack_linear(blk.attn1.to_out[0])
    p.s_ctx_off = col.add_ctx(blk.attn2)
    p.s_attn2, p.s_norm2 = blk.attn2, blk.norm2
    p.s_norm3 = pack_norm(blk.norm3)
    p.s_ff = _pack_ff(blk.ff, blk.norm3)
    assert tb.ff_in is not False and tb.is_res
    p.t_norm_in = pack_norm(tb.norm_in)
    p.t_ff_in = _pack_ff(tb.ff_in)
    p.t_norm1 = pack_norm(tb.norm1)
    p.t_wqkv = _bf(torch.cat([tb.attn1.to_q.weight.detach(), tb.attn1.to_k.weight.detach(), tb.attn1.to_v.weight.detach()], dim=0))
    p.t_wo = pack_linear(tb.attn1.to_out[0])
    assert tb.attn2 is not None, "disable_temporal_crossattention is not used by V3D/SVD"
    p.t_ctx_off = col.add_ctx(tb.attn2)
    p.t_attn2, p.t_norm2 = tb.attn2, tb.norm2
    p.t_norm3 = pack_norm(tb.norm3)
    p.t_ff = _pack_ff(tb.ff, tb.norm3)
    p.tpe = pack_linear(st.time_pos_embed[0]) + pack_linear(st.time_pos_embed[2])
    p.max_period = float(st.max_time_embed_period)
    p.mixer = col.add_mixer(st.time_mixer, 1)
    return p


def cross_attn_pack(p: SVTPack, which: str):
    """(norm2, W_q [C, C], W_k | W_v [2C, ctx_dim], W_o [C, C], b_o) of the spatial ("s") / temporal ("t") attn2 of a transformer block, for contexts of
    more than one token (CrossAttention.forward, attention.py:286-349: to_q / to_k / to_v without bias, to_out[0] with bias).  Packed on first use."""
    if which not in p.x2:
        # first use must not happen inside a HIP-graph capture: the packed weights would be allocated in the graph's private pool (and their
        # conversion kernels recorded into the graph) and then cached for later eager calls - run one eager evaluation first
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("cross_attn_pack: first multi-token cross-attention evaluation inside a HIP-graph capture; run one eager evaluation before capturing")
        attn, norm = (p.s_attn2, p.s_norm2) if which == "s" else (p.t_attn2, p.t_norm2)
        wq = _bf(attn.to_q.weight)
        wkv = _bf(torch.cat([attn.to_k.weight.detach(), attn.to_v.weight.detach()], dim=0))
        p.x2[which] = (pack_norm(norm), wq, wkv) + pack_linear(attn.to_out[0])
    return p.x2[which]


def pack_unet(net) -> UNetPack:
    from ..sgm.modules.diffusionmodules.openaimodel import Downsample, Upsample
    from ..sgm.modules.diffusionmodules.video_model import VideoResBlock
    from ..sgm.modules.video_attention import SpatialVideoTransformer

    col = _Collector()
    dev = net.out[2].weight.device

    def pack_stage(seq) -> list:
        items = []
        for m in seq:
            if isinstance(m, VideoResBlock):
                items.append(("res", pack_resblock(m, col)))
            elif isinstance(m, SpatialVideoTransformer):
                items.append(("svt", pack_svt(m, col)))
            elif isinstance(m, Downsample):
                items.append(("down", pack_conv3x3(m.op)))
            elif isinstance(m, Upsample):
                items.append(("up", pack_conv3x3(m.conv)))
            elif isinstance(m, nn.Conv2d):
                items.append(("conv_in", pack_conv3x3(m)))
            else:
                raise TypeError(f"unsupported U-Net stage member {type(m).__name__}")
        return items

    input_stages = [pack_stage(seq) for seq in net.input_blocks]
    middle = pack_stage(net.middle_block)
    output_stages = [pack_stage(seq) for seq in net.output_blocks]
    conv_in = input_stages[0][0][1]
    te = net.time_embed
    time_embed = pack_linear(te[0]) + pack_linear(te[2])
    label = None
    adm_in = None
    if net.num_classes == "sequential":
        le = net.label_emb[0]
        adm_in = le[0].weight.shape[1]
        label = pack_linear(le[0], k_pad=round_up(adm_in, 8)) + pack_linear(le[2])
    uses_ioi = net.merge_strategy == "learned_with_images"
    return UNetPack(
        device=dev, in_channels=net.in_channels, in_pad=round_up(net.in_channels, 8), model_channels=net.model_channels,
        out_channels=net.out_channels, context_dim=net.context_dim, adm_in=adm_in, time_embed=time_embed, label_emb=label,
        conv_in=conv_in, input_stages=input_stages, middle=middle, output_stages=output_stages,
        out_norm=pack_norm(net.out[0]), out_conv=pack_conv3x3(net.out[2]),
        conv_in_gemm=pack_conv3x3_im2col(net.input_blocks[0][0]) if net.in_channels <= 8 else None,
        out_conv_taps=pack_conv3x3_taps(net.out[2]) if 9 * net.out_channels <= 64 else None,
        emb_w=_bf(torch.cat(col.emb_w, 0)), emb_b=_f(torch.cat(col.emb_b, 0)),
        ctx_w=_bf(torch.cat(col.ctx_w, 0)), ctx_b=_f(torch.cat(col.ctx_b, 0)),
        mix_alpha=torch.tensor(col.alphas, dtype=F32, device=dev), mix_kind=torch.tensor(col.kinds, dtype=torch.int32, device=dev),
        uses_ioi=uses_ioi)


# ---------------------------------------------------------------------------------------------------------
# VAE decoder (reference: sgm/modules/diffusionmodules/model.py:604-748 Decoder; autoencoding/temporal_ae.py)
# ---------------------------------------------------------------------------------------------------------
@dataclass
class AttnPack:
    C: int
    norm: tuple
    wq: tuple
    wk: tuple
    wv: torch.Tensor
    bv: torch.Tensor
    proj: tuple


@dataclass
class VAEPack:
    device: torch.device
    z_channels: int
    z_pad: int
    conv_in: tuple
    mid: list
    up: List[dict]                   # index = i_level: {"blocks": [ResPack], "upsample": (w, b) | None}
    norm_out: tuple
    conv_out: tuple                  # conv3x3 128 -> out_ch (fp32 output, ld 4)
    tmix_w: torch.Tensor             # [out_ch, out_ch, 3] fp32
    tmix_b: torch.Tensor
    out_ch: int


def pack_vae_resblock(rb) -> ResPack:
    p = ResPack(cin=rb.in_channels, cout=rb.out_channels, split=None)
    p.gn1 = pack_norm(rb.norm1)
    p.w1, p.b1 = pack_conv3x3(rb.conv1)
    p.gn2 = pack_norm(rb.norm2)
    p.w2, p.b2 = pack_conv3x3(rb.conv2)
    if rb.in_channels != rb.out_channels:
        p.skip_w, p.skip_b = pack_conv1x1(rb.nin_shortcut)
    ts = rb.time_stack
    p.t_gn1 = pack_norm(ts.in_layers[0])
    p.t_w1, p.t_b1 = pack_convt3(ts.in_layers[2])
    p.t_gn2 = pack_norm(ts.out_layers[0])
    p.t_w2, p.t_b2 = pack_convt3(ts.out_layers[3])
    m = float(rb.mix_factor.detach().float().cpu())
    p.alpha = m if rb.merge_strategy == "fixed" else float(torch.sigmoid(torch.tensor(m)))
    return p


def pack_vae_decoder(dec) -> VAEPack:
    dev = dec.conv_in.weight.device
    a = dec.mid.attn_1
    wv, bv = pack_conv1x1(a.v)
    attn = AttnPack(C=a.in_channels, norm=pack_norm(a.norm), wq=pack_conv1x1(a.q), wk=pack_conv1x1(a.k), wv=wv, bv=bv,
                    proj=pack_conv1x1(a.proj_out))
    mid = [("res", pack_vae_resblock(dec.mid.block_1)), ("attn", attn), ("res", pack_vae_resblock(dec.mid.block_2))]
    up = []
    for lvl in dec.up:
        d = {"blocks": [pack_vae_resblock(b) for b in lvl.block], "upsample": None}
        if hasattr(lvl, "upsample"):
            d["upsample"] = pack_conv3x3(lvl.upsample.conv)
        up.append(d)
    co = dec.conv_out
    tw = co.time_mix_conv.weight.detach()
    return VAEPack(device=dev, z_channels=dec.conv_in.weight.shape[1], z_pad=round_up(dec.conv_in.weight.shape[1], 8),
                   conv_in=pack_conv3x3(dec.conv_in), mid=mid, up=up, norm_out=pack_norm(dec.norm_out),
                   conv_out=pack_conv3x3(co), tmix_w=_f(tw.reshape(tw.shape[0], tw.shape[1], 3)), tmix_b=_f(co.time_mix_conv.bias),
                   out_ch=co.weight.shape[0])


# ---- VAE encoder ---------------------------------------------------------------------------------------------------

@dataclass
class VAEEncPack:
    device: torch.device
    in_ch: int
    in_pad: int
    conv_in: tuple
    down: List[dict]                 # per level: {"blocks": [ResPack], "attn": [AttnPack | None], "downsample": (w, b) | None}
    mid: list
    norm_out: tuple
    conv_out: tuple                  # conv3x3 block_in -> 2 * z_channels (fp32 output)
    out_ch: int


def pack_resnet2d(rb) -> ResPack:
    """Plain 2-D ResnetBlock of the VAE encoder (model.py:94-151, temb_channels = 0)."""
    p = ResPack(cin=rb.in_channels, cout=rb.out_channels, split=None)
    p.gn1 = pack_norm(rb.norm1)
    p.w1, p.b1 = pack_conv3x3(rb.conv1)
    p.gn2 = pack_norm(rb.norm2)
    p.w2, p.b2 = pack_conv3x3(rb.conv2)
    if rb.in_channels != rb.out_channels:
        p.skip_w, p.skip_b = pack_conv1x1(rb.nin_shortcut)
    return p


def _pack_attnblock(a) -> AttnPack:
    wv, bv = pack_conv1x1(a.v)
    return AttnPack(C=a.in_channels, norm=pack_norm(a.norm), wq=pack_conv1x1(a.q), wk=pack_conv1x1(a.k), wv=wv, bv=bv,
                    proj=pack_conv1x1(a.proj_out))


def pack_vae_encoder(enc) -> VAEEncPack:
    down = []
    for lvl in enc.down:
        d = {"blocks": [pack_resnet2d(b) for b in lvl.block], "attn": [_pack_attnblock(a) for a in lvl.attn], "downsample": None}
        if hasattr(lvl, "downsample"):
            d["downsample"] = pack_conv3x3(lvl.downsample.conv)
        down.append(d)
    mid = [("res", pack_resnet2d(enc.mid.block_1)), ("attn", _pack_attnblock(enc.mid.attn_1)), ("res", pack_resnet2d(enc.mid.block_2))]
    cin = enc.conv_in.weight.shape[1]
    return VAEEncPack(device=enc.conv_in.weight.device, in_ch=cin, in_pad=round_up(cin, 8), conv_in=pack_conv3x3(enc.conv_in), down=down,
                      mid=mid, norm_out=pack_norm(enc.norm_out), conv_out=pack_conv3x3(enc.conv_out), out_ch=enc.conv_out.weight.shape[0])
