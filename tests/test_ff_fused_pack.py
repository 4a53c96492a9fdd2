"""Host-side packing of v3d_ff_fused (CPU): the fused operand order (engine/packing.py ff_fused_pack, include/v3d_hip.h) must describe
the same FeedForward as the two-GEMM path and as the reference formula (sgm/modules/attention.py:82-113)."""
import torch
import torch.nn.functional as F

from oracle.ops_emul import EmulOps
from v3d_amd.engine import unet as unet_engine
from v3d_amd.engine.packing import _pack_ff
from v3d_amd.ops import use_backend
from v3d_amd.sgm.modules.attention import FeedForward


def _ff_reference(ff, x):
    proj, lin2 = ff.net[0].proj, ff.net[2]
    v, g = F.linear(x, proj.weight, proj.bias).chunk(2, dim=-1)
    return F.linear(v * F.gelu(g), lin2.weight, lin2.bias)


def test_fused_pack_matches_two_gemm_path_and_reference(monkeypatch):
    torch.manual_seed(0)
    C, M = 320, 256
    ff = FeedForward(C, mult=2, glu=True)
    for prm in ff.parameters():
        prm.data.normal_(0, 0.05)
    ops = EmulOps("cpu", exact=True)
    with use_backend(ops):
        pk = _pack_ff(ff)
    assert pk.w1_fused is not None and pk.w2_fused is not None and pk.b1_fused is not None
    x = torch.randn(M, C)
    res1, res2 = torch.randn(M, C), torch.randn(M, C)
    coef = torch.randn(M // 64, 3)
    want = _ff_reference(ff, x)
    for kw in (dict(res1=res1), dict(res1=res1, res2=res2, coef=coef, coef_rpg=64)):
        monkeypatch.setattr(unet_engine, "_FF_FUSED", True)
        fused = unet_engine.feed_forward(ops, x, pk, **kw)
        monkeypatch.setattr(unet_engine, "_FF_FUSED", False)
        plain = unet_engine.feed_forward(ops, x, pk, **kw)
        torch.testing.assert_close(fused, plain, rtol=1e-4, atol=1e-4)
        if "res2" not in kw:
            torch.testing.assert_close(fused, want + res1, rtol=1e-4, atol=1e-4)


def test_fused_path_not_taken_off_spec():
    """C != 320 or a hidden size the kernel does not take -> no fused operands, the two-GEMM path runs."""
    with use_backend(EmulOps("cpu", exact=True)):
        assert _pack_ff(FeedForward(640, mult=4, glu=True)).w2_fused is None
        assert _pack_ff(FeedForward(320, mult=0.1, glu=True)).w2_fused is None      # hidden 32


def test_dma_tile_order_two_independent_statements_agree():
    """packing.ff_dma_tile_index (index list, product side) and EmulOps._ff_untile (reshape / gather, test side) state the weight
    stream order of include/v3d_hip.h independently: tiling with one and untiling with the other is the identity."""
    from v3d_amd.engine.packing import ff_dma_tile_index
    for rows, cols, sr, sc in ((2560, 320, 64, 320), (320, 1280, 320, 32), (512, 320, 64, 320), (320, 256, 320, 32)):
        w = torch.randn(rows, cols)
        tiled = w.reshape(-1)[ff_dma_tile_index(rows, cols, sr, sc)].reshape(rows, cols)
        assert not torch.equal(tiled, w)
        assert torch.equal(EmulOps._ff_untile(tiled, rows, cols, sr, sc), w)
    # a piece is 1 KiB of consecutive elements: the first 512 bf16 of the stream are rows 0..15 x columns 0..31 of the matrix
    idx = ff_dma_tile_index(128, 320, 64, 320)[:512]
    assert set((idx // 320).tolist()) == set(range(16)) and set((idx % 320).tolist()) == set(range(32))
