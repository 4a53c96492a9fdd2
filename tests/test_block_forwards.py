"""Inner blocks called on their own (reference signatures, NCHW in / out) against the oracle's functional restatement of the same block
(oracle/sgm_oracle.py, pinned to the reference by the whole-network fixtures): the modules a user of the reference's tree may pick out -
`unet.input_blocks[i][0]` (VideoResBlock), `unet.input_blocks[i][1]` (SpatialVideoTransformer), `decoder.mid.block_1` (VAE VideoResBlock),
`decoder.mid.attn_1` (AttnBlock).  CPU: the product's block executors run on the torch restatement of the C-ABI ops (exact=True keeps fp32
storage: any residual is a wiring error; exact=False reproduces the kernels' bf16 rounding points)."""
import pytest
import torch

from conftest import rel_cos
from oracle import sgm_oracle as O
from oracle.ops_emul import EmulOps
from tiny import TINY, build_decoder, build_unet
from v3d_amd.ops import use_backend
from v3d_amd.sgm.modules.autoencoding.temporal_ae import VideoResBlock as VaeVideoResBlock
from v3d_amd.sgm.modules.diffusionmodules.model import AttnBlock
from v3d_amd.sgm.modules.diffusionmodules.video_model import VideoResBlock
from v3d_amd.sgm.modules.video_attention import SpatialVideoTransformer

torch.set_grad_enabled(False)
MODES = [(True, 5e-5, 0.999999), (False, 4e-2, 0.999)]


def _named(net, cls):
    return [(name, m) for name, m in net.named_modules() if type(m) is cls]


def _sd(net):
    return {k: v.float() for k, v in net.state_dict().items()}


@pytest.mark.parametrize("exact,tol,cosmin", MODES)
def test_unet_video_resblock_and_transformer(exact, tol, cosmin):
    T, H, W = TINY["T"], TINY["H"], TINY["W"]
    n = 2 * T
    g = torch.Generator().manual_seed(5)
    with use_backend(EmulOps("cpu", exact=exact)):
        net = build_unet()
        sd = _sd(net)
        ioi = torch.zeros(2, T)
        ioi[1, T // 2] = 1.0
        checked = 0
        for name, rb in _named(net, VideoResBlock)[:3]:
            x = torch.randn(n, rb.channels, H, W, generator=g)
            emb = torch.randn(n, rb.emb_channels, generator=g)
            # (zero-initialised output convolutions would hide the second half of the block: the tiny net's seeded weights are non-zero)
            got = rb(x, emb, T, ioi)
            want = O.video_resblock(sd, name, x, emb, T, ioi)
            assert got.shape == want.shape == (n, rb.out_channels, H, W)
            rel, cos = rel_cos(got, want)
            assert rel <= tol and cos >= cosmin, (name, rel, cos)
            again = rb(x, emb, T, ioi)                       # second call: the cached pack
            assert torch.equal(again, got)
            checked += 1
        for name, st in _named(net, SpatialVideoTransformer)[:2]:
            x = torch.randn(n, st.in_channels, H, W, generator=g)
            ctx = torch.randn(n, 1, TINY.get("context_dim", 1024), generator=g)
            got = st(x, context=ctx, timesteps=T, image_only_indicator=ioi)
            want = O.spatial_video_transformer(sd, name, x, ctx, T, ioi, st.n_heads, float(st.max_time_embed_period))
            rel, cos = rel_cos(got, want)
            assert rel <= tol and cos >= cosmin, (name, rel, cos)
            checked += 1
        assert checked >= 4


@pytest.mark.parametrize("exact,tol,cosmin", MODES)
def test_vae_blocks(exact, tol, cosmin):
    T = TINY["T"]
    g = torch.Generator().manual_seed(6)
    with use_backend(EmulOps("cpu", exact=exact)):
        dec = build_decoder()
        sd = _sd(dec)
        name, rb = _named(dec, VaeVideoResBlock)[0]
        x = torch.randn(T, rb.in_channels, 8, 8, generator=g)
        rel, cos = rel_cos(rb(x, None, False, timesteps=T), O.vae_resblock(sd, name, x, T))
        assert rel <= tol and cos >= cosmin, (name, rel, cos)
        name, ab = _named(dec, AttnBlock)[0]
        x = torch.randn(T, ab.in_channels, 8, 8, generator=g)
        rel, cos = rel_cos(ab(x), O.vae_attn(sd, name, x))
        assert rel <= tol and cos >= cosmin, (name, rel, cos)


def test_pack_is_rebuilt_when_a_parameter_changes():
    T, H, W = TINY["T"], TINY["H"], TINY["W"]
    with use_backend(EmulOps("cpu", exact=True)):
        net = build_unet()
        name, rb = _named(net, VideoResBlock)[0]
        g = torch.Generator().manual_seed(7)
        x, emb = torch.randn(T, rb.channels, H, W, generator=g), torch.randn(T, rb.emb_channels, generator=g)
        ioi = torch.zeros(1, T)
        a = rb(x, emb, T, ioi)
        rb.in_layers[2].bias.add_(1.0)                       # in-place update bumps the parameter's version
        b = rb(x, emb, T, ioi)
        assert not torch.equal(a, b)
        rel, cos = rel_cos(b, O.video_resblock(_sd(net), name, x, emb, T, ioi))
        assert rel <= 5e-5, (rel, cos)


@pytest.mark.parametrize("exact,tol,cosmin", MODES)
def test_transformer_owners(exact, tol, cosmin):
    """FeedForward, CrossAttention (self-attention; one context token) and BasicTransformerBlock of a U-Net transformer, on their own."""
    from v3d_amd.sgm.modules.attention import BasicTransformerBlock
    g = torch.Generator().manual_seed(8)
    with use_backend(EmulOps("cpu", exact=exact)):
        net = build_unet()
        sd = _sd(net)
        name, blk = _named(net, BasicTransformerBlock)[0]
        C = blk.norm1.weight.shape[0]
        B, N = 3, 64
        x = torch.randn(B, N, C, generator=g)
        ctx = torch.randn(B, 1, blk.attn2.to_k.weight.shape[1], generator=g)
        heads = blk.attn1.heads
        for got, want in ((blk.ff(x), O.feedforward(sd, name + ".ff", x)),
                          (blk.attn1(x), O.attention(sd, name + ".attn1", x, None, heads)),
                          (blk.attn2(x, context=ctx), O.attention(sd, name + ".attn2", x, ctx, heads)),
                          (blk(x, context=ctx), O.basic_block(sd, name, x, ctx, heads))):
            assert got.shape == want.shape
            rel, cos = rel_cos(got, want)
            assert rel <= tol and cos >= cosmin, (rel, cos)
        with pytest.raises(NotImplementedError):
            blk.attn2(x, context=torch.randn(B, 2, ctx.shape[-1], generator=g))
