"""Per-operator parity cases: HIP kernel (v3d_amd.hip.HipOps, through the C ABI) vs the torch restatement of
the same op (oracle/ops_emul.py) on identical seeded inputs.  Shared by tests/test_ops_gpu.py (pytest, -m gpu)
and tools/gpu_check.py (prints the whole table without stopping at the first failure).

Tolerance: bf16 kernels vs fp32 restatement: max|err| / max|ref| <= 2e-2 and cosine >= 0.999;
fp32 elementwise ops: 1e-5.
"""
from __future__ import annotations

import math
import os

import torch

from v3d_amd.ops import GEMM_CONV3X3, GEMM_CONVT3, GEMM_LINEAR, GemmCall

BF, F32 = torch.bfloat16, torch.float32
TOL_BF16 = 2e-2
TOL_F32 = 1e-5


def _rand(gen, shape, dtype=BF, scale=1.0, device="cuda"):
    return (torch.randn(shape, generator=gen, device="cpu", dtype=torch.float32) * scale).to(device=device, dtype=dtype)


def compare(a: torch.Tensor, b: torch.Tensor):
    a, b = a.float().flatten(), b.float().flatten()
    if not torch.isfinite(a).all():
        return float("inf"), 0.0
    denom = b.abs().max().clamp_min(1e-12)
    rel = ((a - b).abs().max() / denom).item()
    cos = torch.nn.functional.cosine_similarity(a, b, dim=0).item() if b.abs().max() > 0 else 1.0
    return rel, cos


# ------------------------------------------------------------------------------------------------
def case_groupnorm_small(hip, emu, dev, *, n_img, S, C1, C2=0, imgs_per_stat=1, silu=True, seed=0, supported=True, mean=0.5):
    """v3d_groupnorm_small (one launch: statistics + normalisation of small statistics groups) against fp64 F.group_norm, against the
    three-step form on the same input (same result contract: both round scale / shift alike -> at most a bf16 ulp apart), and bit-equal
    to itself on a second launch."""
    g = torch.Generator().manual_seed(seed)
    C = C1 + C2
    assert hip.groupnorm_small_supported(C1, C2, S, imgs_per_stat) == supported == emu.groupnorm_small_supported(C1, C2, S, imgs_per_stat)
    if not supported:
        return 0.0, 1.0
    off = (torch.rand((C,), generator=g) * 2 - 1) * mean
    x1 = (_rand(g, (n_img * S, C1), F32, 1.0, dev) + off[:C1].to(dev)).to(BF)
    x2 = (_rand(g, (n_img * S, C2), F32, 1.3, dev) + off[C1:].to(dev)).to(BF) if C2 else None
    gamma, beta = _rand(g, (C,), F32, 0.3, dev) + 1.0, _rand(g, (C,), F32, 0.3, dev)
    out = torch.zeros((n_img * S, C), dtype=BF, device=dev)
    hip.groupnorm_small(x1, x2, gamma, beta, out, n_img, S, eps=1e-5, silu=silu, imgs_per_stat=imgs_per_stat)
    again = torch.zeros_like(out)
    hip.groupnorm_small(x1, x2, gamma, beta, again, n_img, S, eps=1e-5, silu=silu, imgs_per_stat=imgs_per_stat)
    assert torch.equal(out, again), "two identical launches differ"
    x = x1.double() if x2 is None else torch.cat([x1.double(), x2.double()], dim=-1)
    n_stat, rows = n_img // imgs_per_stat, imgs_per_stat * S
    ref = torch.nn.functional.group_norm(x.reshape(n_stat, rows, C).permute(0, 2, 1), 32, gamma.double(), beta.double(), 1e-5).permute(0, 2, 1).reshape(n_img * S, C)
    if silu:
        ref = ref * torch.sigmoid(ref)
    table = hip.groupnorm_table(x1, x2, gamma, beta, n_img, S, eps=1e-5, imgs_per_stat=imgs_per_stat)
    three = torch.zeros_like(out)
    hip.groupnorm_apply(x1, x2, table, three, n_img, S, imgs_per_stat, silu)
    r3, _ = compare(out, three)
    assert r3 <= 1e-2, f"one-launch form vs statistics -> finalize -> apply: {r3:.3e}"
    emu_out = torch.zeros_like(out)
    emu.groupnorm_small(x1, x2, gamma, beta, emu_out, n_img, S, eps=1e-5, silu=silu, imgs_per_stat=imgs_per_stat)
    re_, _ = compare(out, emu_out)
    assert re_ <= 1e-2, f"one-launch form vs its emulation: {re_:.3e}"
    return compare(out, ref.float())


def case_gemm(hip, emu, dev, *, M, N, K, mode=GEMM_LINEAR, geglu=False, bias=True, add=False, res=0, coef=False,
              out_fp32=False, conv=None, convt=None, batch=1, lda_pad=0, seed=0, shared_w=True, pad_mode=0, gn_rps=0, expect_streamk=None, expect_family=None):
    """gn_rps > 0: the launch also gathers the GroupNorm partial sums of its output (GemmCall.gn_stats, 32 groups, gn_rps rows per statistics
    group); they are checked against sums over the rows the kernel itself stored (fp32 partials, one slot per writer: 2e-4 of the largest sum)
    and a second launch must reproduce output AND statistics bit for bit (no atomics anywhere)."""
    g = torch.Generator().manual_seed(seed)
    kw = {}
    n_out = N // 2 if geglu else N
    taps = {GEMM_LINEAR: 1, GEMM_CONV3X3: 9, GEMM_CONVT3: 3}[mode]
    if mode == GEMM_CONV3X3:
        n_img, Hin, Win, stride, up = conv
        ptot = 1 if pad_mode else 2      # pad_mode 1: zero pixels on the right / bottom only (VAE encoder Downsample)
        Hout, Wout = (Hin * up + ptot - 3) // stride + 1, (Win * up + ptot - 3) // stride + 1
        a_rows = n_img * Hin * Win
        M = n_img * Hout * Wout
        kw.update(Hin=Hin, Win=Win, Hout=Hout, Wout=Wout, stride=stride, up=up, pad_mode=pad_mode)
    elif mode == GEMM_CONVT3:
        B, T, S, halo, tmin, tmax = convt
        M = B * T * S
        a_rows = M + 2 * halo * S
        kw.update(T=T, S=S, tmin=tmin, tmax=tmax, a_row0=halo * S)
    else:
        a_rows = M
    lda = K + lda_pad
    if batch > 1:
        A = _rand(g, (batch, a_rows, lda), device=dev)[:, :, :K]
        W = _rand(g, (N, K), scale=1 / math.sqrt(K), device=dev) if shared_w else _rand(g, (batch, N, K), scale=1 / math.sqrt(K), device=dev)
    else:
        A = _rand(g, (a_rows, lda), device=dev)[:, :K]
        W = _rand(g, (taps, N, K), scale=1 / math.sqrt(K * taps), device=dev)
    odt = F32 if out_fp32 else BF
    oshape = (batch, M, n_out) if batch > 1 else (M, n_out)
    out_h = torch.zeros(oshape, dtype=odt, device=dev)
    out_e = torch.zeros(oshape, dtype=odt, device=dev)
    if bias:
        kw["bias"] = _rand(g, (N,), F32, 0.5, dev)
    if add:
        rpg = max(1, M // 6)
        ngroups = (M + rpg - 1) // rpg
        kw.update(add=_rand(g, (ngroups, N + 24), F32, 0.5, dev), add_rpg=rpg, add_ld=N + 24)
    if res >= 1:
        kw.update(res1=_rand(g, (M, n_out), device=dev), c_res1=0.75)
    if res >= 2:
        kw.update(res2=_rand(g, (M, n_out), device=dev), c_res2=-0.5)
    kw["c_acc"] = 0.6 if res else 1.0
    if coef:
        rpg = max(1, M // 5)
        kw.update(coef=_rand(g, ((M + rpg - 1) // rpg, 3), F32, 1.0, dev), coef_rpg=rpg)
    base = dict(A=A, W=W, M=M, N=N, K=K, mode=mode, geglu=geglu, batch=batch, **kw)
    if gn_rps:
        from v3d_amd.ops import OpsBase
        st_h = torch.zeros((M // gn_rps, OpsBase.gn_nslots(gn_rps), 32, 2), dtype=F32, device=dev)
        st_e = torch.zeros_like(st_h)
        st_2, out_2 = torch.zeros_like(st_h), torch.zeros_like(out_h)
        hip.gemm(GemmCall(out=out_2, gn_stats=st_2, gn_rps=gn_rps, gn_cpg=N // 32, **base))
        hip.gemm(GemmCall(out=out_h, gn_stats=st_h, gn_rps=gn_rps, gn_cpg=N // 32, **base))
        assert torch.equal(out_2, out_h) and torch.equal(st_2, st_h), "two identical launches differ: the statistics epilogue is not deterministic"
        emu.gemm(GemmCall(out=out_e, gn_stats=st_e, gn_rps=gn_rps, gn_cpg=N // 32, **base))
        v = out_h.float().reshape(M // gn_rps, gn_rps, 32, N // 32)
        want = torch.stack([v.sum(dim=(1, 3)), (v * v).sum(dim=(1, 3))], dim=-1)
        got = st_h.sum(dim=1)
        err = ((got - want).abs().max() / want.abs().max()).item()
        assert err <= 2e-4, f"gn_stats epilogue: partial sums off by {err:.3e} of the largest sum"
        r_e, _ = compare(st_e.sum(dim=1), want)
        assert r_e <= 2e-2, f"emulated gn_stats far from the kernel's ({r_e:.3e})"
        if expect_streamk is not None:
            _assert_no_sk_timeouts(hip)
        return compare(out_h, out_e)
    sk0 = _sk_counter(hip, "v3d_debug_sk_launches") if expect_streamk is not None else 0
    hip.gemm(GemmCall(out=out_h, **base))
    if expect_family is not None and getattr(hip, "name", "") == "hip" and not any(os.environ.get(k) for k in ("V3D_GEMM_IMPL", "V3D_GEMM_CFG", "V3D_GEMM_V6")):
        # the dispatcher's choice under the default policy (gemm.hip dispatch): 6 = two persistent 4-wave blocks per CU on 192 x 160 tiles
        fam = hip.last_gemm_launch()["family"]
        assert fam == expect_family, f"expected kernel family {expect_family}, the library launched {hip.last_gemm_launch()}"
        out_2 = torch.zeros_like(out_h)
        hip.gemm(GemmCall(out=out_2, **base))
        assert torch.equal(out_2, out_h), "two identical launches differ"
    if expect_streamk is not None:
        # stream-K tail of the persistent v3 kernels (the last round's tiles shared out over all CUs): taken / not taken as the plan says,
        # and a second launch reproduces the first bit for bit (partials are added in block order)
        if os.environ.get("V3D_STREAMK", "1") != "0" and not os.environ.get("V3D_GEMM_IMPL"):
            assert (_sk_counter(hip, "v3d_debug_sk_launches") > sk0) == expect_streamk, "stream-K plan: the launch did not take the expected path"
        out_2 = torch.zeros_like(out_h)
        hip.gemm(GemmCall(out=out_2, **base))
        assert torch.equal(out_2, out_h), "two identical launches differ"
        _assert_no_sk_timeouts(hip)
    emu.gemm(GemmCall(out=out_e, **base))
    return compare(out_h, out_e)


def case_ff_fused(hip, emu, dev, *, M, C=320, hidden=1280, res=1, coef=False, seed=0, ln=False):
    """v3d_ff_fused (GEGLU feed-forward, hidden tensor on chip) against its emulation (= two emulated GEMMs with the same packing).
    ln: v3d_ln_ff_fused - LayerNorm of the (non-zero-mean) rows inside the kernel, against layer_norm -> bf16 -> the same emulation."""
    g = torch.Generator().manual_seed(seed)
    x = _rand(g, (M, C), device=dev)
    if ln:
        x = (x.float() * 1.5 + 0.3).to(BF)
    w1 = _rand(g, (2 * hidden, C), scale=1 / math.sqrt(C), device=dev)
    b1 = _rand(g, (2 * hidden,), F32, 0.5, dev)
    w2 = _rand(g, (C, hidden), scale=1 / math.sqrt(hidden), device=dev)
    b2 = _rand(g, (C,), F32, 0.5, dev)
    kw = {}
    if res >= 1:
        kw.update(res1=_rand(g, (M, C), device=dev), c_res1=0.75)
    if res >= 2:
        kw.update(res2=_rand(g, (M, C), device=dev), c_res2=-0.5)
    kw["c_acc"] = 0.6 if res else 1.0
    if coef:
        rpg = 128
        kw.update(coef=_rand(g, ((M + rpg - 1) // rpg, 3), F32, 1.0, dev), coef_rpg=rpg)
    o_h = torch.zeros((M, C), dtype=BF, device=dev)
    o_e = torch.zeros((M, C), dtype=BF, device=dev)
    if ln:
        hip.ln_ff_fused(x, 1e-5, w1, b1, w2, b2, o_h, **kw)
        emu.ln_ff_fused(x, 1e-5, w1, b1, w2, b2, o_e, **kw)
        return compare(o_h, o_e)
    hip.ff_fused(x, w1, b1, w2, b2, o_h, **kw)
    o_2 = torch.zeros_like(o_h)
    hip.ff_fused(x, w1, b1, w2, b2, o_2, **kw)
    assert torch.equal(o_h, o_2), "two identical v3d_ff_fused launches differ"
    emu.ff_fused(x, w1, b1, w2, b2, o_e, **kw)
    return compare(o_h, o_e)


def case_groupnorm(hip, emu, dev, *, n_img, S, C1, C2=0, imgs_per_stat=1, eps=1e-5, silu=True, seed=0, mean=0.5, std=1.0):
    """GroupNorm (stats -> finalize -> apply) against the emulation, and - independent of it - against torch.nn.functional.group_norm in
    fp64 on the same bf16 input.  mean / std: per-channel offsets up to `mean` x the spread (real SVD checkpoints have channels with
    |mean| >> std: E[x^2] - E[x]^2 in fp32 would cancel there; the finalize step works in fp64).  Two identical calls must agree bit for bit."""
    g = torch.Generator().manual_seed(seed)
    C = C1 + C2
    off = (torch.rand((C,), generator=g) * 2 - 1) * mean
    x1 = (_rand(g, (n_img * S, C1), F32, std, dev) + off[:C1].to(dev)).to(BF)
    x2 = (_rand(g, (n_img * S, C2), F32, 2.0 * std, dev) - off[C1:].to(dev)).to(BF) if C2 else None
    gamma, beta = _rand(g, (C,), F32, 0.3, dev) + 1.0, _rand(g, (C,), F32, 0.3, dev)
    o_h = hip.groupnorm(x1, x2, gamma, beta, n_img, S, eps=eps, silu=silu, imgs_per_stat=imgs_per_stat)
    o_2 = hip.groupnorm(x1, x2, gamma, beta, n_img, S, eps=eps, silu=silu, imgs_per_stat=imgs_per_stat)
    assert torch.equal(o_h, o_2), "two identical GroupNorm calls differ: statistics are not deterministic"
    o_e = emu.groupnorm(x1, x2, gamma, beta, n_img, S, eps=eps, silu=silu, imgs_per_stat=imgs_per_stat)
    if getattr(hip, "name", "") == "hip":
        # ABI 5: the one-launch statistics + table (last block of a statistics group folds its slots) against the stats -> finalize launches.
        # Both add the same fp32 slot sums in fp64, in different fixed orders: equal to fp64 rounding, i.e. to an ulp of the fp32 table; the
        # fused launch itself is bit-reproducible (whichever block arrives last computes the same thing).
        t_f = hip.groupnorm_table(x1, x2, gamma, beta, n_img, S, eps=eps, imgs_per_stat=imgs_per_stat)
        t_f2 = hip.groupnorm_table(x1, x2, gamma, beta, n_img, S, eps=eps, imgs_per_stat=imgs_per_stat)
        assert torch.equal(t_f, t_f2), "two identical fused statistics + table launches differ"
        hip._GN_FUSED_TABLE = False
        try:
            t_3 = hip.groupnorm_table(x1, x2, gamma, beta, n_img, S, eps=eps, imgs_per_stat=imgs_per_stat)
        finally:
            del hip._GN_FUSED_TABLE
        assert torch.allclose(t_f, t_3, rtol=2e-6, atol=1e-6 * float(t_3.abs().max())), f"fused table differs from stats -> finalize by {(t_f - t_3).abs().max().item():.3e}"
    # fp64 reference: [n_stat, C, ips * S] in the layout F.group_norm wants
    x = (x1 if x2 is None else torch.cat([x1, x2], dim=-1)).double().reshape(n_img // imgs_per_stat, imgs_per_stat * S, C).permute(0, 2, 1)
    y = torch.nn.functional.group_norm(x, 32, gamma.double(), beta.double(), eps).permute(0, 2, 1).reshape(n_img * S, C)
    if silu:
        y = y * torch.sigmoid(y)
    r64, c64 = compare(o_h, y)
    rel, cos = compare(o_h, o_e)
    return max(rel, r64), min(cos, c64)


def case_layernorm(hip, emu, dev, *, M, C, add=False, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = (_rand(g, (M, C), F32, 1.5, dev) + 0.2).to(BF)
    gamma, beta = _rand(g, (C,), F32, 0.3, dev) + 1.0, _rand(g, (C,), F32, 0.3, dev)
    kw = {}
    xs_h = xs_e = None
    if add:
        rpg = max(1, M // 4)
        kw = dict(add=_rand(g, ((M + rpg - 1) // rpg, C), F32, 0.5, dev), add_rpg=rpg, add_ld=C)
        xs_h, xs_e = torch.zeros_like(x), torch.zeros_like(x)
    o_h, o_e = torch.zeros_like(x), torch.zeros_like(x)
    hip.layernorm(x, gamma, beta, o_h, 1e-5, xsum_out=xs_h, **kw)
    emu.layernorm(x, gamma, beta, o_e, 1e-5, xsum_out=xs_e, **kw)
    rel, cos = compare(o_h, o_e)
    if add:
        r2, c2 = compare(xs_h, xs_e)
        rel, cos = max(rel, r2), min(cos, c2)
    return rel, cos


def case_attn_spatial(hip, emu, dev, *, n_img, S, heads, seed=0, spike=False):
    g = torch.Generator().manual_seed(seed)
    C = heads * 64
    qk = _rand(g, (n_img * S, 2 * C), device=dev)
    if spike:  # force large running-max jumps late in the key sequence (online-softmax rescale path)
        qk[S // 2 + 3, C:] *= 6.0
        qk[S - 5, C:] *= 9.0
    vT = _rand(g, (n_img, C, S), device=dev)
    o_h = torch.zeros((n_img * S, C), dtype=BF, device=dev)
    o_e = torch.zeros_like(o_h)
    hip.attn_spatial(qk[:, :C], qk[:, C:], vT, o_h, n_img, S, heads, 0.125)
    emu.attn_spatial(qk[:, :C], qk[:, C:], vT, o_e, n_img, S, heads, 0.125)
    return compare(o_h, o_e)


def case_attn_temporal(hip, emu, dev, *, B, Tq, Tk, S, heads, seed=0):
    g = torch.Generator().manual_seed(seed)
    C = heads * 64
    q = _rand(g, (B, Tq, S, C), device=dev)
    kv = _rand(g, (B, Tk, S, 2 * C), device=dev)
    o_h = torch.zeros((B, Tq, S, C), dtype=BF, device=dev)
    o_e = torch.zeros_like(o_h)
    hip.attn_temporal(q, kv[..., :C], kv[..., C:], o_h, heads, 0.125)
    emu.attn_temporal(q, kv[..., :C], kv[..., C:], o_e, heads, 0.125)
    return compare(o_h, o_e)


def case_attn_vae(hip, emu, dev, *, n_img, S, C=512, seed=0, spike=False, bias=True):
    """v3d_attn_vae_d512 (single head of width C, streamed softmax) vs SDPA; `spike` forces running-max jumps beyond the deferred-rescale
    threshold late in the key sequence (cdna guide rule 26: the rare branch needs an input that takes it)."""
    g = torch.Generator().manual_seed(seed)
    q = _rand(g, (n_img * S, C), device=dev)
    k = _rand(g, (n_img * S, C + 64), device=dev)[:, :C]        # strided key rows (ldk != C)
    if spike:
        q[7] *= 3.0
        k[S // 2 + 3] = (q[7].float() * 4.0).to(BF)             # raw score ~ 12 |q|^2: far above every other key of query 7
        k[S - 5] *= 9.0
    vT = _rand(g, (n_img, C, S), device=dev)
    b = _rand(g, (C,), F32, 0.5, dev) if bias else None
    o_h = torch.zeros((n_img * S, C), dtype=BF, device=dev)
    o_e = torch.zeros_like(o_h)
    hip.attn_vae(q, k, vT, b, o_h, n_img, S, C, float(C) ** -0.5)
    emu.attn_vae(q, k, vT, b, o_e, n_img, S, C, float(C) ** -0.5)
    return compare(o_h, o_e)


def case_ln_proj(hip, emu, dev, *, M, N, n_rm, S, seed=0):
    """v3d_ln_proj (LayerNorm + q | k | v projection, row-major and transposed outputs) against layer_norm -> bf16 -> matmul."""
    from v3d_amd.engine.packing import ln_proj_pack
    g = torch.Generator().manual_seed(seed)
    C = 320
    x = (_rand(g, (M, C + 8), F32, 1.5, dev) + 0.3).to(BF)[:, :C]          # strided rows, non-zero mean
    w = _rand(g, (N, C), scale=1 / math.sqrt(C), device=dev)
    gamma, beta = _rand(g, (C,), F32, 0.3, dev) + 1.0, _rand(g, (C,), F32, 0.3, dev)
    wp, bias = ln_proj_pack(w, gamma, beta)
    o_h, t_h = hip.ln_proj(x, 1e-5, wp, bias, n_rm, S)
    o_e, t_e = emu.ln_proj(x, 1e-5, wp, bias, n_rm, S)
    rel, cos = 0.0, 1.0
    for a, b in ((o_h, o_e), (t_h, t_e)):
        if a is not None:
            r, c = compare(a, b)
            rel, cos = max(rel, r), min(cos, c)
    return rel, cos


def case_attn_fp8(hip, emu, dev, *, n_img, S, heads, seed=0, what="attn"):
    """fp8 attention chain.  what="quant": the two quantisation kernels against torch.float8_e4m3fn (bytes may differ by one code where a value
    sits on a rounding boundary after the scale division: compared after dequantisation); what="attn": v3d_attn_spatial_fp8 against exact
    attention on the SAME dequantised operands (remaining difference: P rounded to e4m3, <= 6 % per probability, averaged over the keys);
    what="vs_bf16": the whole fp8 chain against the bf16 kernel on the unquantised operands (the tolerance the scene config states)."""
    g = torch.Generator().manual_seed(seed)
    C = heads * 64
    qk = _rand(g, (n_img * S, 2 * C), device=dev)
    vT = _rand(g, (n_img, C, S), device=dev)
    qk8_h, sc_h = hip.quant_fp8_tiles(qk, n_img, S)
    v8_h, vs_h = hip.quant_fp8_slab(vT, heads)
    qk8_e, sc_e = emu.quant_fp8_tiles(qk, n_img, S)
    v8_e, vs_e = emu.quant_fp8_slab(vT, heads)
    if what == "quant":
        rows = torch.arange(S, device=dev) // 64
        dq = lambda x8, sc: x8.view(torch.float8_e4m3fn).float().reshape(n_img, S, 2 * heads, 64) * sc[:, rows][..., None]
        dv = lambda x8, sc: x8.view(torch.float8_e4m3fn).float().reshape(n_img, heads, 64, S) * sc[:, :, None, None]
        r1, c1 = compare(dq(qk8_h, sc_h), dq(qk8_e, sc_e))
        r2, c2 = compare(dv(v8_h, vs_h), dv(v8_e, vs_e))
        r3, _ = compare(sc_h, sc_e)
        r4, _ = compare(vs_h, vs_e)
        return max(r1, r2, r3, r4), min(c1, c2)
    o_h = torch.zeros((n_img * S, C), dtype=BF, device=dev)
    o_e = torch.zeros_like(o_h)
    hip.attn_spatial_fp8(qk8_h, sc_h, v8_h, vs_h, o_h, n_img, S, heads, 0.125)
    if what == "attn":
        emu.attn_spatial_fp8(qk8_h, sc_h, v8_h, vs_h, o_e, n_img, S, heads, 0.125)
    else:
        hip.attn_spatial(qk[:, :C], qk[:, C:], vT, o_e, n_img, S, heads, 0.125)
    return compare(o_h, o_e)


def case_conv_gn(hip, emu, dev, *, N, C1, C2=0, conv=None, convt=None, add=False, res=0, coef=False, gn_out=False, silu=True, seed=0,
                 expect_fused=True, mean=0.5, expect_streamk=None):
    """GroupNorm (+SiLU) in the operand path of the convolution (GemmCall.gn_in / A2): the LDS-haloed kernels normalise the raw input
    tile on its way into LDS.  Reference = the emulation (normalise -> round to bf16 -> convolution); the statistics table comes from the
    product's own stats / finalize kernels.  conv = (n_img, H, W) 3x3 stride 1; convt = (B, T, S) the (3,1,1) conv with the 3-D norm."""
    from v3d_amd.ops import OpsBase
    g = torch.Generator().manual_seed(seed)
    K = C1 + C2
    if conv is not None:
        n_img, H, W = conv
        S, ips, mode = H * W, 1, GEMM_CONV3X3
        kw = dict(mode=mode, Hin=H, Win=W, Hout=H, Wout=W)
        taps = 9
    else:
        B, T, S = convt
        n_img, ips, mode = B * T, T, GEMM_CONVT3
        kw = dict(mode=mode, T=T, S=S, tmin=0, tmax=T - 1)
        taps = 3
    M = n_img * S
    off = (torch.rand((K,), generator=g) * 2 - 1) * mean
    x1 = (_rand(g, (M, C1), F32, 1.0, dev) + off[:C1].to(dev)).to(BF)
    x2 = (_rand(g, (M, C2), F32, 1.5, dev) + off[C1:].to(dev)).to(BF) if C2 else None
    gamma, beta = _rand(g, (K,), F32, 0.3, dev) + 1.0, _rand(g, (K,), F32, 0.3, dev)
    table = hip.groupnorm_table(x1, x2, gamma, beta, n_img, S, eps=1e-5, imgs_per_stat=ips)
    Wt = _rand(g, (taps, N, K), scale=1 / math.sqrt(K * taps), device=dev)
    kw.update(bias=_rand(g, (N,), F32, 0.5, dev))
    if add:
        kw.update(add=_rand(g, (n_img, N + 24), F32, 0.5, dev), add_rpg=S, add_ld=N + 24)
    if res >= 1:
        kw.update(res1=_rand(g, (M, N), device=dev), c_res1=0.75, c_acc=0.6)
    if coef:
        kw.update(coef=_rand(g, (n_img, 3), F32, 1.0, dev), coef_rpg=S)
    base = dict(A=x1, A2=x2, W=Wt, M=M, N=N, K=K, gn_in=table, gn_in_rps=ips * S, gn_in_silu=silu, **kw)
    o_h = torch.zeros((M, N), dtype=BF, device=dev)
    o_e = torch.zeros_like(o_h)
    skw_h, skw_e = {}, {}
    if gn_out:
        rps = ips * S
        st_h = torch.zeros((M // rps, OpsBase.gn_nslots(rps, ips), 32, 2), dtype=F32, device=dev)
        skw_h = dict(gn_stats=st_h, gn_rps=rps, gn_cpg=N // 32)
        skw_e = dict(gn_stats=torch.zeros_like(st_h), gn_rps=rps, gn_cpg=N // 32)
    call = GemmCall(out=o_h, **base, **skw_h)
    fused = hip.gemm_gn_in_supported(call)
    assert fused == expect_fused, f"gemm_gn_in_supported = {fused}, the case expects {expect_fused}"
    if not fused:
        return 0.0, 1.0
    sk0 = _sk_counter(hip, "v3d_debug_sk_launches")
    hip.gemm(call)
    if expect_streamk is not None and os.environ.get("V3D_STREAMK", "1") != "0":
        assert (_sk_counter(hip, "v3d_debug_sk_launches") > sk0) == expect_streamk, "stream-K plan: the launch did not take the expected path"
    emu.gemm(GemmCall(out=o_e, **base, **skw_e))
    o_2 = torch.zeros_like(o_h)
    if gn_out:
        st_2 = torch.zeros_like(st_h)
        hip.gemm(GemmCall(out=o_2, **base, **dict(skw_h, gn_stats=st_2)))
        assert torch.equal(st_2, st_h), "statistics epilogue of the haloed kernel is not deterministic"
        v = o_h.float().reshape(M // rps, rps, 32, N // 32)
        want = torch.stack([v.sum(dim=(1, 3)), (v * v).sum(dim=(1, 3))], dim=-1)
        err = ((st_h.sum(dim=1) - want).abs().max() / want.abs().max()).item()
        assert err <= 2e-4, f"gn_stats epilogue of the haloed kernel: partial sums off by {err:.3e} of the largest sum"
    else:
        hip.gemm(GemmCall(out=o_2, **base))
    assert torch.equal(o_2, o_h), "two identical launches of the haloed kernel differ"
    _assert_no_sk_timeouts(hip)
    return compare(o_h, o_e)


def _sk_counter(hip, name):
    import ctypes
    fn = getattr(hip.lib, name)
    fn.restype = ctypes.c_longlong
    fn.argtypes = []
    return int(fn())


def _assert_no_sk_timeouts(hip):
    """stream-K tail of the persistent kernels: an owner piece that gave up waiting for a donor's partial (bounded spin) would still return -
    with a wrong tile.  The library counts those; the count must stay 0."""
    assert _sk_counter(hip, "v3d_debug_sk_timeouts") == 0, "a stream-K hand-off timed out"


def case_convt3_split_halo(hip, emu, dev, *, B, T, S, N, K, first=False, last=False, seed=0):
    """CONVT3 in the split-halo layout of frame sharding (ABI 3): [B*S halo | B*T*S local | B*S halo] rows, all B samples in one launch.
    The halo slabs of a global end are filled with NaN: they must never be read."""
    g = torch.Generator().manual_seed(seed)
    M = B * T * S
    A = _rand(g, ((B + B * T + B) * S, K), device=dev)
    if first:
        A[:B * S] = float("nan")
    if last:
        A[(B + B * T) * S:] = float("nan")
    W = _rand(g, (3, N, K), scale=1 / math.sqrt(3 * K), device=dev)
    kw = dict(A=A, W=W, M=M, N=N, K=K, mode=GEMM_CONVT3, T=T, S=S, tmin=0 if first else -1, tmax=T - 1 if last else T, a_row0=B * S,
              halo_rows=B * S, bias=_rand(g, (N,), F32, 0.5, dev), res1=_rand(g, (M, N), device=dev))
    o_h = torch.zeros((M, N), dtype=BF, device=dev)
    o_e = torch.zeros_like(o_h)
    hip.gemm(GemmCall(out=o_h, **kw))
    emu.gemm(GemmCall(out=o_e, **kw))
    return compare(o_h, o_e)


def case_softmax(hip, emu, dev, *, rows, L, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = _rand(g, (rows, L), F32, 3.0, dev)
    o_h = torch.zeros((rows, L), dtype=BF, device=dev)
    o_e = torch.zeros_like(o_h)
    hip.softmax_rows(x, o_h)
    emu.softmax_rows(x, o_e)
    return compare(o_h, o_e)


def case_elementwise(hip, emu, dev, seed=0):
    """All small sampler / boundary kernels; returns the worst (rel, cos) over them, plus per-op detail."""
    g = torch.Generator().manual_seed(seed)
    res = {}
    n, T, H, W = 12, 6, 8, 8
    t = _rand(g, (n,), F32, 2.0, dev).abs() + 0.1
    res["timestep_embedding"] = compare(hip.timestep_embedding(t, 64), emu.timestep_embedding(t, 64))
    res["timestep_embedding_odd"] = compare(hip.timestep_embedding(t, 65, 100.0), emu.timestep_embedding(t, 65, 100.0))
    a, b = _rand(g, (n, 96), F32, 1.0, dev), _rand(g, (n, 96), F32, 1.0, dev)
    res["silu_add"] = compare(hip.silu_add(a, b), emu.silu_add(a, b))
    res["silu"] = compare(hip.silu_add(a), emu.silu_add(a))
    sig = t * 10
    for i, (h_, e_) in enumerate(zip(hip.edm_scalings(sig), emu.edm_scalings(sig))):
        res[f"edm_scalings{i}"] = compare(h_, e_)
    x = _rand(g, (n, 4, H, W), F32, 1.0, dev)
    cond = _rand(g, (n, 4, H, W), F32, 1.0, dev)
    sc = _rand(g, (n,), F32, 1.0, dev)
    res["pack_input"] = compare(hip.pack_input(x, sc, cond, 8), emu.pack_input(x, sc, cond, 8))
    res["pack_input_pad"] = compare(hip.pack_input(x, None, None, 8), emu.pack_input(x, None, None, 8))
    # the U-Net's first / last convolution as GEMMs: unfolded packed input, gather-sum of the nine taps' products; and, end to end, the two
    # forms of each convolution against the implicit-GEMM launch they replace
    res["pack_input_im2col"] = compare(hip.pack_input_im2col3x3(x, sc, cond, 96), emu.pack_input_im2col3x3(x, sc, cond, 96))
    res["pack_input_im2col_nocond"] = compare(hip.pack_input_im2col3x3(x, None, None, 96), emu.pack_input_im2col3x3(x, None, None, 96))
    yt = _rand(g, (n * H * W, 64), F32, 1.0, dev)
    bt = _rand(g, (4,), F32, 1.0, dev)
    res["tapsum3x3"] = compare(hip.tapsum3x3(yt, bt, n, H, W, 4), emu.tapsum3x3(yt, bt, n, H, W, 4))
    res["tapsum3x3_nobias_c3"] = compare(hip.tapsum3x3(yt[:, :27], None, n, H, W, 3), emu.tapsum3x3(yt[:, :27], None, n, H, W, 3))
    w_in = _rand(g, (9, 32, 8), scale=0.3, device=dev)                      # [tap][O][I]
    b_in = _rand(g, (32,), F32, 0.5, dev)
    w96 = torch.zeros(32, 96, dtype=BF, device=dev)
    w96[:, :72] = w_in.permute(1, 0, 2).reshape(32, 72)
    direct = hip.conv3x3(hip.pack_input(x, sc, cond, 8), w_in, b_in, n, H, W)
    res["conv_in_as_gemm"] = compare(hip.linear(hip.pack_input_im2col3x3(x, sc, cond, 96), w96, b_in), direct)
    xo = _rand(g, (n * H * W, 64), device=dev)
    w_out = _rand(g, (9, 4, 64), scale=0.1, device=dev)
    w36 = torch.zeros(64, 64, dtype=BF, device=dev)
    w36[:36] = w_out.reshape(36, 64)
    direct = hip.conv3x3(xo, w_out, bt, n, H, W, out_dtype=F32)
    res["conv_out_as_gemm"] = compare(hip.tapsum3x3(hip.linear(xo, w36, None, out_dtype=F32), bt, n, H, W, 4), direct)
    net = _rand(g, (n * H * W, 4), F32, 1.0, dev)
    res["denoise_combine"] = compare(hip.denoise_combine(net, x, sc, t), emu.denoise_combine(net, x, sc, t))
    scale = _rand(g, (T,), F32, 1.0, dev) + 3
    res["cfg_combine"] = compare(hip.cfg_combine(x, scale, T), emu.cfg_combine(x, scale, T))
    res["euler_step"] = compare(hip.euler_step(x, cond, sig, sig * 0.5), emu.euler_step(x, cond, sig, sig * 0.5))
    nxt = sig * 0.5
    nxt[::2] = 0.0                      # exercises the "noise level 0 -> keep the Euler step" branch per sample
    eul = emu.euler_step(x, cond, sig, nxt)
    den2 = (x * 0.3 + cond * 0.1).contiguous()
    res["heun_step"] = compare(hip.heun_step(x, cond, eul, den2, sig, nxt), emu.heun_step(x, cond, eul, den2, sig, nxt))
    res["axpb"] = compare(hip.axpb_f32(x, 1.5, -0.25), emu.axpb_f32(x, 1.5, -0.25))
    alpha = torch.sigmoid(_rand(g, (5,), F32, 1.0, dev))
    kind = torch.tensor([0, 1, 0, 1, 1], dtype=torch.int32, device=dev)
    ioi = torch.zeros(n, dtype=F32, device=dev)
    ioi[3] = 1.0
    res["blend_coefs"] = compare(hip.blend_coefs(alpha, kind, ioi, n), emu.blend_coefs(alpha, kind, ioi, n))
    res["blend_coefs_noioi"] = compare(hip.blend_coefs(alpha, kind, None, n), emu.blend_coefs(alpha, kind, None, n))
    res["nchw_to_nhwc"] = compare(hip.nchw_to_nhwc_bf16(x, 0.5, 8), emu.nchw_to_nhwc_bf16(x, 0.5, 8))
    xt = _rand(g, (2 * T * H * W, 4), F32, 1.0, dev)
    w3, b3 = _rand(g, (3, 3, 3), F32, 0.5, dev), _rand(g, (3,), F32, 0.5, dev)
    res["tmix_small"] = compare(hip.tmix_small(xt, w3, b3, 2, T, H * W, 3, 0, T - 1), emu.tmix_small(xt, w3, b3, 2, T, H * W, 3, 0, T - 1))
    src = _rand(g, (40, 48), device=dev)
    d_h = torch.zeros((40, 64), dtype=BF, device=dev)
    d_e = torch.zeros_like(d_h)
    hip.copy2d_bf16(src[:, 8:40], d_h[:, 16:48])
    emu.copy2d_bf16(src[:, 8:40], d_e[:, 16:48])
    res["copy2d"] = compare(d_h, d_e)
    # output stage and the ViT activation: uint8 frames must be BIT-EXACT with the reference's clamp / * 255 / astype(uint8) chain
    fr = (_rand(g, (3, 3, 24, 20), F32, 0.8, dev) * 1.2).contiguous()       # includes values outside [-1, 1]
    u_h, u_e = hip.frames_to_uint8(fr), emu.frames_to_uint8(fr)
    exact = bool((u_h == u_e).all()) and u_h.shape == (3, 24, 20, 3) and u_h.dtype == torch.uint8
    res["frames_to_uint8_bitexact"] = (0.0, 1.0) if exact else (1.0, 0.0)
    ge = _rand(g, (33, 40), device=dev) * 3
    res["gelu"] = compare(hip.gelu(ge), emu.gelu(ge))
    return res


# ------------------------------------------------------------------------------------------------
# The case table.  (name, fn, kwargs, tol).  Shapes marked [V3D] are the exact shapes of the V3D_512 census
# (SURVEY.md Appendix A.2); the others are ragged / edge shapes (M, N, K tails, odd T, uneven frame shards).
def all_cases(full: bool = True):
    C = []
    L, C3, CT = GEMM_LINEAR, GEMM_CONV3X3, GEMM_CONVT3
    C += [
        ("gemm_linear_tiny", case_gemm, dict(M=64, N=64, K=64), TOL_BF16),
        ("gemm_linear_ragged", case_gemm, dict(M=200, N=72, K=40, add=True, res=2), TOL_BF16),
        ("gemm_linear_n4_fp32", case_gemm, dict(M=300, N=4, K=320, out_fp32=True), TOL_BF16),
        ("gemm_linear_n3_ldo3", case_gemm, dict(M=130, N=3, K=128, out_fp32=True, bias=True), TOL_BF16),
        ("gemm_linear_coef", case_gemm, dict(M=384, N=320, K=320, res=2, coef=True), TOL_BF16),
        ("gemm_linear_lda_strided", case_gemm, dict(M=256, N=128, K=64, lda_pad=128), TOL_BF16),
        ("gemm_geglu", case_gemm, dict(M=192, N=256, K=320, geglu=True, res=1), TOL_BF16),
        ("gemm_geglu_n64tile", case_gemm, dict(M=192, N=320, K=64, geglu=True), TOL_BF16),
        ("gemm_batched_sharedW", case_gemm, dict(M=96, N=192, K=128, batch=3, bias=False), TOL_BF16),
        ("ff_fused_res1", case_ff_fused, dict(M=384, res=1), TOL_BF16),
        ("ff_fused_blend", case_ff_fused, dict(M=128 * 5, res=2, coef=True), TOL_BF16),
        ("ff_fused_plain_hidden256", case_ff_fused, dict(M=256, hidden=256, res=0), TOL_BF16),
        ("ff_fused_two_blocks_per_cu", case_ff_fused, dict(M=128 * 300, res=1, seed=3), TOL_BF16),
        ("ln_ff_fused_res1", case_ff_fused, dict(M=384, res=1, ln=True, seed=4), TOL_BF16),
        ("ln_ff_fused_blend", case_ff_fused, dict(M=128 * 5, res=2, coef=True, ln=True, seed=5), TOL_BF16),
        ("ln_ff_fused_many_blocks", case_ff_fused, dict(M=128 * 600, res=1, ln=True, seed=6), TOL_BF16),
        # (round 6: 1.25 rounds and 0.4 rounds of row blocks on 256 CUs - sizes a tail split along the hidden dimension was tried on; see tools/ff_rounds_probe.py)
        ("ff_fused_partial_round", case_ff_fused, dict(M=128 * 320, res=2, coef=True, seed=7), TOL_BF16),
        ("ln_ff_fused_partial_round", case_ff_fused, dict(M=128 * 320, res=1, ln=True, seed=8), TOL_BF16),
        ("ff_fused_under_one_round", case_ff_fused, dict(M=128 * 100, res=1, seed=9), TOL_BF16),
        ("gemm_batched_perbatchW", case_gemm, dict(M=128, N=128, K=512, batch=2, bias=False, shared_w=False, out_fp32=True), TOL_BF16),
        ("conv3x3_small", case_gemm, dict(M=0, N=64, K=32, mode=C3, conv=(2, 8, 8, 1, 1)), TOL_BF16),
        ("conv3x3_odd_hw", case_gemm, dict(M=0, N=40, K=24, mode=C3, conv=(3, 7, 5, 1, 1), add=True, res=1), TOL_BF16),
        ("conv3x3_stride2", case_gemm, dict(M=0, N=64, K=64, mode=C3, conv=(2, 16, 16, 2, 1)), TOL_BF16),
        ("conv3x3_up2", case_gemm, dict(M=0, N=64, K=64, mode=C3, conv=(2, 8, 8, 1, 2)), TOL_BF16),
        ("conv3x3_stride2_asym_pad", case_gemm, dict(M=0, N=64, K=64, mode=C3, conv=(2, 16, 12, 2, 1), pad_mode=1), TOL_BF16),
        ("conv3x3_stride2_asym_pad_big", case_gemm, dict(M=0, N=256, K=128, mode=C3, conv=(4, 64, 64, 2, 1), pad_mode=1, res=1), TOL_BF16),
        ("conv3x3_k8", case_gemm, dict(M=0, N=320, K=8, mode=C3, conv=(2, 16, 16, 1, 1)), TOL_BF16),
        ("convt3_small", case_gemm, dict(M=0, N=64, K=64, mode=CT, convt=(2, 5, 16, 0, 0, 4), res=1, coef=True), TOL_BF16),
        ("convt3_T1", case_gemm, dict(M=0, N=64, K=64, mode=CT, convt=(3, 1, 16, 0, 0, 0)), TOL_BF16),
        ("convt3_halo_shard", case_gemm, dict(M=0, N=64, K=64, mode=CT, convt=(1, 3, 16, 1, -1, 3)), TOL_BF16),
        ("convt3_halo_leftedge", case_gemm, dict(M=0, N=64, K=64, mode=CT, convt=(1, 2, 16, 1, 0, 2)), TOL_BF16),
        ("gn2d_320", case_groupnorm, dict(n_img=4, S=256, C1=320), TOL_BF16),
        ("gn2d_concat_1920", case_groupnorm, dict(n_img=2, S=64, C1=1280, C2=640), TOL_BF16),
        ("gn2d_concat_2560", case_groupnorm, dict(n_img=2, S=64, C1=1280, C2=1280, silu=False, eps=1e-6), TOL_BF16),
        ("gn3d_T3", case_groupnorm, dict(n_img=6, S=64, C1=320, imgs_per_stat=3), TOL_BF16),
        ("gn_vae_128", case_groupnorm, dict(n_img=2, S=1024, C1=128, eps=1e-6), TOL_BF16),
        ("gn_S1", case_groupnorm, dict(n_img=3, S=1, C1=64), TOL_BF16),
        ("gn2d_1088_fold_lds", case_groupnorm, dict(n_img=2, S=128, C1=1088), TOL_BF16),      # 1024 < C < 1120: the fused fold's scratch exceeds the partial-sum table
        # channels with |mean| up to 30 / 100 x their spread (fp64 F.group_norm on the same bf16 input is the reference)
        ("gn2d_offcentre_30", case_groupnorm, dict(n_img=3, S=1024, C1=320, mean=30.0, seed=11), TOL_BF16),
        ("gn2d_offcentre_100", case_groupnorm, dict(n_img=2, S=1024, C1=640, mean=100.0, seed=12), TOL_BF16),
        ("gn3d_offcentre_100", case_groupnorm, dict(n_img=6, S=256, C1=320, imgs_per_stat=3, mean=100.0, std=0.5, seed=13), TOL_BF16),
        # GroupNorm + SiLU in the operand path of the LDS-haloed convolutions (conv.hip)
        # one-launch GroupNorm of small statistics groups: the 8 x 8 level (2-D, two-source 2-D, 3-D over 18 frames), the 16 x 16 level's 2-D norm,
        # off-centre channels, and shapes it must refuse (3-D at 16 x 16: 4608 rows; 20 channels per group)
        ("gn_small_2d_L3", case_groupnorm_small, dict(n_img=36, S=64, C1=1280), TOL_BF16),
        ("gn_small_2d_L3_concat", case_groupnorm_small, dict(n_img=36, S=64, C1=1280, C2=1280, seed=1), TOL_BF16),
        ("gn_small_3d_L3", case_groupnorm_small, dict(n_img=36, S=64, C1=1280, imgs_per_stat=18, seed=2), TOL_BF16),
        ("gn_small_2d_L2_nosilu", case_groupnorm_small, dict(n_img=36, S=256, C1=1280, silu=False, seed=3), TOL_BF16),
        ("gn_small_offcentre", case_groupnorm_small, dict(n_img=4, S=64, C1=1280, mean=30.0, seed=4), TOL_BF16),
        ("gn_small_odd_rows", case_groupnorm_small, dict(n_img=3, S=35, C1=256, seed=5), TOL_BF16),
        ("gn_small_refused_3d_L2", case_groupnorm_small, dict(n_img=36, S=256, C1=1280, imgs_per_stat=18, supported=False), TOL_BF16),
        ("gn_small_refused_cpg20", case_groupnorm_small, dict(n_img=36, S=64, C1=640, supported=False), TOL_BF16),
        ("conv_gn_64_plain", case_conv_gn, dict(N=320, C1=320, conv=(3, 64, 64)), TOL_BF16),
        ("conv_gn_64_straddle_add_gnout", case_conv_gn, dict(N=320, C1=64, conv=(6, 64, 64), add=True, gn_out=True, seed=1), TOL_BF16),
        ("conv_gn_64_concat_res", case_conv_gn, dict(N=320, C1=64, C2=32, conv=(3, 64, 64), res=1, seed=2), TOL_BF16),
        ("conv_gn_32_n640", case_conv_gn, dict(N=640, C1=96, conv=(6, 32, 32), add=True, gn_out=True, seed=3), TOL_BF16),
        ("conv_gn_16_n1280", case_conv_gn, dict(N=1280, C1=64, C2=64, conv=(12, 16, 16), res=1, gn_out=True, seed=4), TOL_BF16),
        # 48 tiles of 40 chunks on 256 CUs: stream-K tail with five or six blocks per tile (one owner piece + several donors, added in block order)
        ("conv_gn_16_streamk_many_donors", case_conv_gn, dict(N=320, C1=1280, conv=(36, 16, 16), res=1, gn_out=True, seed=10, expect_streamk=True), TOL_BF16),
        ("conv_gn_64_offcentre", case_conv_gn, dict(N=320, C1=64, conv=(3, 64, 64), mean=30.0, seed=5), TOL_BF16),
        ("conv_gn_nosilu", case_conv_gn, dict(N=320, C1=32, conv=(3, 32, 32), silu=False, seed=6, expect_fused=False), TOL_BF16),
        # W = 8 (round 6): three whole 8 x 8 images per tile, two image lines per fragment, up to three statistics groups per halo, stream-K over many donors
        ("conv_gn_8_n1280", case_conv_gn, dict(N=1280, C1=64, conv=(36, 8, 8)), TOL_BF16),
        ("conv_gn_8_concat_res_gnout", case_conv_gn, dict(N=1280, C1=64, C2=32, conv=(36, 8, 8), res=1, gn_out=True, seed=11), TOL_BF16),
        ("conv_gn_8_add_gnout_streamk", case_conv_gn, dict(N=1280, C1=1280, conv=(36, 8, 8), add=True, gn_out=True, seed=12, expect_streamk=True), TOL_BF16),
        ("conv_gn_8_three_images", case_conv_gn, dict(N=320, C1=64, conv=(3, 8, 8), res=1, gn_out=True, mean=20.0, seed=13), TOL_BF16),
        ("conv_gn_unsupported_8x8_four_images", case_conv_gn, dict(N=1280, C1=64, conv=(4, 8, 8), expect_fused=False), TOL_BF16),
        ("convt_gn_T6", case_conv_gn, dict(N=320, C1=64, convt=(2, 6, 1024), res=1, coef=True, seed=7), TOL_BF16),
        ("convt_gn_T18_add_gnout", case_conv_gn, dict(N=320, C1=96, convt=(2, 18, 256), add=True, gn_out=True, seed=8), TOL_BF16),
        ("convt_gn_T12_n640", case_conv_gn, dict(N=640, C1=64, convt=(1, 12, 64), res=1, seed=9), TOL_BF16),
        ("convt_gn_unsupported_T5", case_conv_gn, dict(N=320, C1=64, convt=(2, 5, 64), expect_fused=False), TOL_BF16),
        ("ln_320", case_layernorm, dict(M=257, C=320), TOL_BF16),
        ("ln_1280_add", case_layernorm, dict(M=96, C=1280, add=True), TOL_BF16),
        ("ln_64", case_layernorm, dict(M=33, C=64, add=True), TOL_BF16),
        ("attn_spatial_S64", case_attn_spatial, dict(n_img=3, S=64, heads=2), TOL_BF16),
        ("attn_spatial_S256", case_attn_spatial, dict(n_img=2, S=256, heads=3), TOL_BF16),
        ("attn_spatial_S144_ragged", case_attn_spatial, dict(n_img=2, S=144, heads=1), TOL_BF16),
        ("attn_spatial_S16", case_attn_spatial, dict(n_img=2, S=16, heads=1), TOL_BF16),
        ("attn_spatial_spike", case_attn_spatial, dict(n_img=1, S=512, heads=1, spike=True), TOL_BF16),
        ("attn_temporal_T18", case_attn_temporal, dict(B=2, Tq=18, Tk=18, S=16, heads=5), TOL_BF16),
        ("attn_temporal_T3", case_attn_temporal, dict(B=2, Tq=3, Tk=3, S=4, heads=1), TOL_BF16),
        ("attn_temporal_shard_2of18", case_attn_temporal, dict(B=2, Tq=2, Tk=18, S=8, heads=2), TOL_BF16),
        ("attn_temporal_T25", case_attn_temporal, dict(B=1, Tq=25, Tk=25, S=8, heads=1), TOL_BF16),
        ("attn_vae_C128_S64", case_attn_vae, dict(n_img=2, S=64, C=128), TOL_BF16),
        ("attn_vae_C128_S48_ragged", case_attn_vae, dict(n_img=3, S=48, C=128, bias=False), TOL_BF16),
        ("attn_vae_C256_S200_ragged", case_attn_vae, dict(n_img=1, S=200, C=256), TOL_BF16),
        ("attn_vae_C512_S256", case_attn_vae, dict(n_img=2, S=256, C=512), TOL_BF16),
        ("attn_vae_C512_spike", case_attn_vae, dict(n_img=1, S=512, C=512, spike=True), TOL_BF16),
        ("convt3_split_halo_mid", case_convt3_split_halo, dict(B=2, T=3, S=16, N=64, K=64), TOL_BF16),
        ("convt3_split_halo_first", case_convt3_split_halo, dict(B=2, T=2, S=40, N=72, K=64, first=True), TOL_BF16),
        ("convt3_split_halo_last_v3", case_convt3_split_halo, dict(B=2, T=9, S=256, N=320, K=320, last=True), TOL_BF16),
        # GroupNorm partial sums gathered by the producing GEMM (v3 <GN> epilogue; other kernels fall back to the stand-alone pass inside v3d_gemm)
        ("gn_epi_conv_L0_straddle", case_gemm, dict(M=0, N=320, K=320, mode=GEMM_CONV3X3, conv=(3, 64, 64, 1, 1), gn_rps=4096, res=1), TOL_BF16),
        ("gn_epi_conv_3d", case_gemm, dict(M=0, N=320, K=64, mode=GEMM_CONV3X3, conv=(6, 32, 32, 1, 1), gn_rps=3 * 1024, add=True, seed=1), TOL_BF16),
        ("gn_epi_conv_8x8", case_gemm, dict(M=0, N=1280, K=128, mode=GEMM_CONV3X3, conv=(36, 8, 8, 1, 1), gn_rps=64, seed=2), TOL_BF16),
        ("gn_epi_conv_256tile", case_gemm, dict(M=0, N=512, K=96, mode=GEMM_CONV3X3, conv=(8, 32, 32, 1, 1), gn_rps=1024, seed=3), TOL_BF16),
        ("gn_epi_convt3", case_gemm, dict(M=0, N=320, K=320, mode=GEMM_CONVT3, convt=(2, 3, 1024, 1, 0, 2), gn_rps=3 * 1024, res=1, coef=True, seed=4), TOL_BF16),
        ("gn_epi_linear", case_gemm, dict(M=192 * 9, N=320, K=320, gn_rps=192 * 3, res=1, seed=5), TOL_BF16),
        ("gn_epi_fallback_v2", case_gemm, dict(M=0, N=128, K=64, mode=GEMM_CONV3X3, conv=(2, 16, 16, 1, 1), gn_rps=256, seed=6), TOL_BF16),
        ("gn_epi_fallback_ragged", case_gemm, dict(M=0, N=320, K=64, mode=GEMM_CONV3X3, conv=(3, 20, 20, 1, 1), gn_rps=400, seed=7), TOL_BF16),
        ("ln_proj_qkv_spatial", case_ln_proj, dict(M=3 * 256, N=960, n_rm=640, S=256), TOL_BF16),
        ("ln_proj_qkv_temporal", case_ln_proj, dict(M=128 * 5, N=960, n_rm=960, S=128, seed=1), TOL_BF16),
        ("ln_proj_all_transposed", case_ln_proj, dict(M=2 * 128, N=128, n_rm=0, S=128, seed=2), TOL_BF16),
        ("ln_proj_many_blocks", case_ln_proj, dict(M=128 * 700, N=960, n_rm=640, S=128 * 50, seed=3), TOL_BF16),
        # round 6: the last round's row blocks cut into slab ranges (256 CUs: 288 row blocks = 1 round + 32 in 5 parts; 96 row blocks in 2 parts each; 350 = 1 + 94 in 2)
        ("ln_proj_tail_5_parts", case_ln_proj, dict(M=256 * 288, N=960, n_rm=640, S=256 * 36, seed=4), TOL_BF16),
        ("ln_proj_small_2_parts", case_ln_proj, dict(M=256 * 96, N=960, n_rm=640, S=256 * 24, seed=5), TOL_BF16),
        ("ln_proj_tail_all_rowmajor", case_ln_proj, dict(M=256 * 320, N=960, n_rm=960, S=256 * 40, seed=6), TOL_BF16),
        ("ln_proj_8w_rowmajor", case_ln_proj, dict(M=256 * 300, N=960, n_rm=960, S=256, seed=4), TOL_BF16),
        ("fp8_quant_S256", case_attn_fp8, dict(n_img=2, S=256, heads=2, what="quant"), 7e-2),
        ("fp8_quant_S144_ragged", case_attn_fp8, dict(n_img=2, S=144, heads=1, what="quant"), 7e-2),
        ("fp8_attn_S256", case_attn_fp8, dict(n_img=2, S=256, heads=2), 6e-2),
        ("fp8_attn_S144_ragged", case_attn_fp8, dict(n_img=3, S=144, heads=1, seed=2), 6e-2),
        ("fp8_attn_S1024_vs_bf16", case_attn_fp8, dict(n_img=2, S=1024, heads=3, what="vs_bf16"), 1.5e-1),
        ("softmax_4096", case_softmax, dict(rows=64, L=4096), TOL_BF16),
        ("softmax_64", case_softmax, dict(rows=7, L=64), TOL_BF16),
    ]
    if full:
        C += [
            # [V3D] level-0 shapes at batch 36 are large; keep parity cases to a few images of the same geometry
            ("gemm_V3D_ff1_L0", case_gemm, dict(M=4 * 4096, N=2560, K=320, geglu=True), TOL_BF16),
            ("gemm_V3D_ff2_L0", case_gemm, dict(M=4 * 4096, N=320, K=1280, res=2, coef=True), TOL_BF16),
            ("gemm_V3D_qk_L2", case_gemm, dict(M=36 * 256, N=2560, K=1280, bias=False), TOL_BF16),
            ("gemm_V3D_emb_M36", case_gemm, dict(M=36, N=1280, K=1280, out_fp32=True), TOL_BF16),
            ("conv3x3_V3D_L0", case_gemm, dict(M=0, N=320, K=320, mode=C3, conv=(4, 64, 64, 1, 1), add=True), TOL_BF16),
            ("conv_gn_V3D_L0_in", case_conv_gn, dict(N=320, C1=320, conv=(36, 64, 64), add=True, gn_out=True, seed=21, expect_streamk=False), TOL_BF16),
            ("conv_gn_V3D_L0_concat960", case_conv_gn, dict(N=320, C1=640, C2=320, conv=(6, 64, 64), add=True, gn_out=True, seed=22), TOL_BF16),
            ("conv_gn_V3D_L1_out", case_conv_gn, dict(N=640, C1=640, conv=(36, 32, 32), res=1, gn_out=True, seed=23, expect_streamk=True), TOL_BF16),
            ("conv_gn_V3D_L2_concat2560", case_conv_gn, dict(N=1280, C1=1280, C2=1280, conv=(36, 16, 16), add=True, gn_out=True, seed=24), TOL_BF16),
            ("convt_gn_V3D_L0", case_conv_gn, dict(N=320, C1=320, convt=(2, 18, 4096), res=1, coef=True, seed=25), TOL_BF16),
            ("convt_gn_V3D_L1", case_conv_gn, dict(N=640, C1=640, convt=(2, 18, 1024), add=True, gn_out=True, seed=26, expect_streamk=True), TOL_BF16),
            ("conv3x3_V3D_L3_2560", case_gemm, dict(M=0, N=1280, K=2560, mode=C3, conv=(36, 8, 8, 1, 1), res=1), TOL_BF16),
            ("conv3x3_V3D_down", case_gemm, dict(M=0, N=320, K=320, mode=C3, conv=(4, 64, 64, 2, 1)), TOL_BF16),
            ("conv3x3_V3D_up", case_gemm, dict(M=0, N=640, K=640, mode=C3, conv=(4, 32, 32, 1, 2)), TOL_BF16),
            ("conv3x3_V3D_out4", case_gemm, dict(M=0, N=4, K=320, mode=C3, conv=(4, 64, 64, 1, 1), out_fp32=True), TOL_BF16),
            # partial last rounds of the persistent v3 kernels (192 / 384 tiles of 192 x 320 on 256 CUs).  Only the LDS-haloed convolution kernels
            # carry a stream-K tail: in v3 it was built twice and measured slower both times (NOTES.md 10.5) - these launches stay classic
            ("lin_V3D_L2_ffout_partial_round", case_gemm, dict(M=9216, N=1280, K=5120, res=1, expect_streamk=False, seed=31), TOL_BF16),
            ("lin_V3D_L1_ffout_partial_round", case_gemm, dict(M=36864, N=640, K=2560, res=2, expect_streamk=False, seed=32), TOL_BF16),
            ("convt3_V3D_L2_partial_round", case_gemm, dict(M=0, N=1280, K=1280, mode=CT, convt=(2, 18, 256, 0, 0, 17), res=1, add=True, expect_streamk=False, seed=34), TOL_BF16),
            ("convt3_V3D_L1", case_gemm, dict(M=0, N=640, K=640, mode=CT, convt=(2, 18, 1024, 0, 0, 17), res=1, coef=True, add=True), TOL_BF16),
            # round 6: the 192 x 160 tiles on two 4-wave blocks per CU (gemm_kernel_v6) where v3's tiles leave a partial round
            ("lin_V3D_L1_640_v6", case_gemm, dict(M=36864, N=640, K=640, add=True, res=1, expect_family=6, seed=41), TOL_BF16),
            ("lin_V3D_L1_640_v6_res2", case_gemm, dict(M=36864, N=640, K=640, res=2, expect_family=6, seed=42), TOL_BF16),
            ("lin_V3D_L2_qkv_v6", case_gemm, dict(M=9216, N=3840, K=1280, bias=False, expect_family=6, seed=43), TOL_BF16),
            ("lin_V3D_L0_320_stays_v3", case_gemm, dict(M=147456, N=320, K=320, add=True, res=1, expect_family=3, seed=44), TOL_BF16),
            ("gn2d_V3D_960_64x64", case_groupnorm, dict(n_img=4, S=4096, C1=640, C2=320), TOL_BF16),
            ("gn3d_V3D_L2", case_groupnorm, dict(n_img=36, S=256, C1=1280, imgs_per_stat=18), TOL_BF16),
            ("ln_V3D_L0", case_layernorm, dict(M=2 * 4096, C=320, add=True), TOL_BF16),
            ("attn_spatial_V3D_L0", case_attn_spatial, dict(n_img=2, S=4096, heads=5), TOL_BF16),
            ("attn_spatial_V3D_L1", case_attn_spatial, dict(n_img=4, S=1024, heads=10), TOL_BF16),
            ("attn_temporal_V3D_L1", case_attn_temporal, dict(B=2, Tq=18, Tk=18, S=1024, heads=10), TOL_BF16),
            ("attn_vae_V3D_4096", case_attn_vae, dict(n_img=2, S=4096, C=512), TOL_BF16),
            ("attn_vae_scene_9216", case_attn_vae, dict(n_img=1, S=9216, C=512, seed=4), TOL_BF16),
            ("fp8_attn_scene_9216_vs_bf16", case_attn_fp8, dict(n_img=1, S=9216, heads=5, what="vs_bf16", seed=6), 1.5e-1),
            ("vae_attn_scores", case_gemm, dict(M=1024, N=1024, K=512, batch=2, bias=False, shared_w=False, out_fp32=True), TOL_BF16),
        ]
    return C


def run_case(hip, emu, dev, name, fn, kwargs, tol):
    rel, cos = fn(hip, emu, dev, **kwargs)
    ok = (rel <= tol) and (cos >= (0.995 if name.startswith("fp8_") else 0.999))     # fp8 attention: the scene config's stated bar
    return rel, cos, ok
