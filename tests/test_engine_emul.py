"""Host-logic parity on CPU: the product engine (v3d_amd.engine / v3d_amd.sgm plugin classes) executed with the torch
restatement of the C-ABI ops (oracle/ops_emul.py) injected, against the reference-generated fixtures.

exact=True keeps fp32 storage: any residual error is a wiring error (tolerance 5e-5 relative).  exact=False reproduces
the bf16 rounding points of the HIP kernels and is held to the bf16 tolerance of SURVEY.md §8d."""
import pytest
import torch

from conftest import rel_cos
from oracle.ops_emul import EmulOps
from tiny import SAMPLER_FIXTURES, TINY, build_decoder, build_denoiser, build_sampler, build_unet, ctx5_tokens, decoder_latents, tiny_unet_inputs
from v3d_amd.ops import use_backend
from v3d_amd.sgm.modules.diffusionmodules.wrappers import OpenAIWrapper

torch.set_grad_enabled(False)
MODES = [(True, 5e-5, 0.999999), (False, 4e-2, 0.999)]


@pytest.mark.parametrize("exact,tol,cosmin", MODES)
def test_unet(golden, exact, tol, cosmin):
    p = TINY
    T = p["T"]
    _, _, _, x8, ts, ctx, y = tiny_unet_inputs(T, p["H"], p["W"], p["seed"])
    with use_backend(EmulOps("cpu", exact=exact)):
        net = build_unet()
        ioi = torch.zeros(2, T)
        out = net(x8, ts, context=ctx, y=y, num_video_frames=T, image_only_indicator=ioi)
        assert out.shape == golden["unet_out"].shape
        rel, cos = rel_cos(out, golden["unet_out"])
        assert rel <= tol and cos >= cosmin, (rel, cos)
        ioi[1, 1] = 1.0
        rel, cos = rel_cos(net(x8, ts, context=ctx, y=y, num_video_frames=T, image_only_indicator=ioi), golden["unet_out_ioi"])
        assert rel <= tol and cos >= cosmin, (rel, cos)


@pytest.mark.parametrize("exact,tol,cosmin", MODES)
def test_unet_multi_token_context(golden, exact, tol, cosmin):
    """A context of 5 tokens per image takes the engine's general cross-attention path (engine/unet.py cross_attention: LayerNorm, to_q, to_k | to_v on
    the tokens, v3d_attn_temporal's problem shape with 32-row query blocks and a zero key stride, to_out + residual) instead of the one-token
    collapse - against the fixture the reference's own modules produced; one token per image must still take the collapsed path bit for bit."""
    p = TINY
    T = p["T"]
    _, _, _, x8, ts, ctx, y = tiny_unet_inputs(T, p["H"], p["W"], p["seed"])
    with use_backend(EmulOps("cpu", exact=exact)):
        net = build_unet()
        ioi = torch.zeros(2, T)
        out = net(x8, ts, context=ctx5_tokens(T, p["seed"]), y=y, num_video_frames=T, image_only_indicator=ioi)
        rel, cos = rel_cos(out, golden["unet_out_ctx5"])
        assert rel <= tol and cos >= cosmin, (rel, cos)
        assert (out - golden["unet_out"]).abs().max() > 1e-2           # (another context: another output)
        # the same token five times == that one token (softmax over identical keys): the general path against the collapsed one
        rep = net(x8, ts, context=ctx.repeat(1, 5, 1), y=y, num_video_frames=T, image_only_indicator=ioi)
        rel, cos = rel_cos(rep, golden["unet_out"])
        assert rel <= tol and cos >= cosmin, (rel, cos)


@pytest.mark.parametrize("exact,tol,cosmin", MODES)
def test_decoder(golden, exact, tol, cosmin):
    T = TINY["T"]
    z = decoder_latents(T)
    with use_backend(EmulOps("cpu", exact=exact)):
        dec = build_decoder()
        rel, cos = rel_cos(dec(z, timesteps=T), golden["dec_out"])
        assert rel <= tol and cos >= cosmin, (rel, cos)
        rel, cos = rel_cos(dec(z[:1], timesteps=1), golden["dec_out_T1"])
        assert rel <= tol and cos >= cosmin, (rel, cos)


@pytest.mark.parametrize("exact,tol,cosmin", MODES)
def test_sampler_loop(golden, exact, tol, cosmin):
    p = TINY
    T = p["T"]
    noise, c, uc, *_ = tiny_unet_inputs(T, p["H"], p["W"], p["seed"])
    with use_backend(EmulOps("cpu", exact=exact)):
        net = build_unet()
        sampler, den, wr = build_sampler(T), build_denoiser(), OpenAIWrapper(net)
        extra = {"image_only_indicator": torch.zeros(2, T), "num_video_frames": T}
        x0 = noise.clone()
        z = sampler(lambda i, s, cc: den(wr, i, s, cc, **extra), x0, cond=c, uc=uc)
        rel, cos = rel_cos(z, golden["sample_z"])
        assert rel <= tol and cos >= cosmin, (rel, cos)
        # prepare_sampling_loop scales the CALLER's noise tensor in place, like the reference (sampling.py:50)
        assert torch.allclose(x0, noise * (1 + 700.0 ** 2) ** 0.5, rtol=1e-5)


@pytest.mark.parametrize("kind,key", [kv for kv in SAMPLER_FIXTURES if kv[0] != "euler_linear"])
def test_other_samplers_and_guiders(golden, kind, key):
    """HeunEDMSampler x CentralPredictionGuider and EulerEDMSampler x VanillaCFG behind the same plugin API,
    exact-fp32 emulated kernels against the reference fixtures."""
    p = TINY
    T = p["T"]
    noise, c, uc, *_ = tiny_unet_inputs(T, p["H"], p["W"], p["seed"])
    with use_backend(EmulOps("cpu", exact=True)):
        net = build_unet()
        sampler, den, wr = build_sampler(T, kind=kind), build_denoiser(), OpenAIWrapper(net)
        extra = {"image_only_indicator": torch.zeros(2, T), "num_video_frames": T}
        z = sampler(lambda i, s, cc: den(wr, i, s, cc, **extra), noise.clone(), cond=c, uc=uc)
        rel, cos = rel_cos(z, golden[key])
        assert rel <= 1e-4 and cos >= 0.99999, (rel, cos)


def test_packing_invalidation():
    with use_backend(EmulOps("cpu", exact=True)):
        net = build_unet()
        p1 = net.packed()
        assert net.packed() is p1
        net.load_state_dict(net.state_dict())
        assert net._packed is None
        net.packed()
        net.float()
        assert net._packed is None


@pytest.mark.parametrize("exact,tol,cosmin", MODES)
def test_encoder(golden, exact, tol, cosmin):
    """the product Encoder (engine wiring incl. the asymmetric-pad stride-2 Downsample) on emulated kernels."""
    from tiny import build_encoder, encoder_image
    with use_backend(EmulOps("cpu", exact=exact)):
        out = build_encoder()(encoder_image())
        assert out.shape == golden["enc_moments"].shape
        rel, cos = rel_cos(out, golden["enc_moments"])
        assert rel <= tol and cos >= cosmin, (rel, cos)


def test_decode_first_stage_chunked_product_side():
    """SURVEY a-18 through the PRODUCT (DiffusionEngine.decode_first_stage, emulated kernels, exact fp32): chunks of
    en_and_decode_n_samples_a_time frames are decoded with timesteps = len(chunk) (video_diffusion.py:182-210), which is not the full
    decode (Appendix B-13); the 5-D "b t c h w" input form is reshaped back."""
    from oracle import sgm_oracle as O
    from v3d_amd import configs, synth
    from v3d_amd.sgm.util import instantiate_from_config
    T = 3
    cfg = configs.v3d_512_config(num_frames=T, num_steps=2, model_channels=64, vae_ch=32)["model"]
    cfg["params"]["en_and_decode_n_samples_a_time"] = 2
    with use_backend(EmulOps("cpu", exact=True)):
        eng = instantiate_from_config(cfg).eval()
        eng.load_state_dict(synth.seeded_state_dict(eng, 1234), strict=True)
        z = decoder_latents(T) * 0.18215
        out = eng.decode_first_stage(z)
        out5 = eng.decode_first_stage(z.reshape(1, T, 4, 8, 8))
    dsd = {k: v.float() for k, v in eng.first_stage_model.decoder.state_dict().items()}
    dcfg = synth.decoder_config(32)
    ref = O.decode_first_stage(dsd, dcfg, z, eng.scale_factor, 2)
    rel, cos = rel_cos(out, ref)
    assert rel <= 5e-5 and cos >= 0.999999, (rel, cos)
    full = O.decode_first_stage(dsd, dcfg, z, eng.scale_factor, T)
    assert (ref - full).abs().max() > 1e-2
    assert out5.shape == (1, T, 3, 64, 64) and torch.equal(out5[0], out)


def test_load_last_embedder_is_refused():
    from v3d_amd import configs
    from v3d_amd.sgm.util import instantiate_from_config
    cfg = configs.v3d_512_config(num_frames=3, num_steps=2, model_channels=64, vae_ch=32)["model"]
    cfg["params"]["load_last_embedder"] = True
    with use_backend(EmulOps("cpu", exact=True)), pytest.raises(NotImplementedError, match="load_last_embedder"):
        instantiate_from_config(cfg)


def test_unet_rejects_oversized_context_and_missing_indicator():
    """Contexts of 1 .. 32 tokens per image are served (one: the collapsed path; more: the general cross-attention, round 5) - anything longer is
    refused loudly, not truncated; learned_with_images needs the indicator (util.py:352-354)."""
    p = TINY
    T = p["T"]
    _, _, _, x8, ts, ctx, y = tiny_unet_inputs(T, p["H"], p["W"], p["seed"])
    with use_backend(EmulOps("cpu", exact=True)):
        net = build_unet()
        with pytest.raises(AssertionError, match="1 <= N <= 32 tokens"):
            net(x8, ts, context=ctx.repeat(1, 33, 1), y=y, num_video_frames=T, image_only_indicator=torch.zeros(2, T))
        with pytest.raises(AssertionError, match="image_only_indicator is required"):
            net(x8, ts, context=ctx, y=y, num_video_frames=T, image_only_indicator=None)


def test_ln_proj_path_equals_unfused_path(monkeypatch):
    """The 64x64-level shortcut (v3d_ln_proj: LayerNorm + q | k | v projection in one kernel, V^T written directly) against the
    LayerNorm -> GEMM -> swapped batched GEMM sequence it replaces, on a one-level width-320 network with exact-fp32 emulated kernels."""
    from v3d_amd import synth
    from v3d_amd.sgm.modules.diffusionmodules.video_model import VideoUNet
    cfg = dict(synth.unet_config(320), channel_mult=[1], num_res_blocks=1, attention_resolutions=[1])
    T, H, W = 2, 8, 16                                   # 128 tokens per image: the kernel's block granularity
    g = torch.Generator().manual_seed(5)
    n = 2 * T
    x8, ts = torch.randn(n, 8, H, W, generator=g), torch.randn(n, generator=g)
    ctx, y = torch.randn(n, 1, 1024, generator=g), torch.randn(n, 768, generator=g)
    with use_backend(EmulOps("cpu", exact=True)):
        net = VideoUNet(**cfg).eval()
        net.load_state_dict(synth.seeded_state_dict(net, 21))
        assert net.packed().input_stages[1][1][1].s_wqkv_fused is not None
        monkeypatch.setenv("V3D_LN_PROJ", "1")
        a = net(x8, ts, context=ctx, y=y, num_video_frames=T, image_only_indicator=torch.zeros(2, T))
        monkeypatch.setenv("V3D_LN_PROJ", "0")
        b = net(x8, ts, context=ctx, y=y, num_video_frames=T, image_only_indicator=torch.zeros(2, T))
    rel, cos = rel_cos(a, b)
    assert rel <= 1e-5 and cos >= 0.999999, (rel, cos)


def test_ln_ff_fused_path_equals_unfused_path(monkeypatch):
    """norm3 + feed-forward in one launch (v3d_ln_ff_fused: LayerNorm affine folded into the first Linear at pack time, rows normalised inside
    the kernel) against v3d_layernorm -> v3d_ff_fused, on a one-level width-320 network with exact-fp32 emulated kernels."""
    from v3d_amd import synth
    from v3d_amd.engine import unet as unet_engine
    from v3d_amd.sgm.modules.diffusionmodules.video_model import VideoUNet
    cfg = dict(synth.unet_config(320), channel_mult=[1], num_res_blocks=1, attention_resolutions=[1])
    T, H, W = 2, 8, 16
    g = torch.Generator().manual_seed(6)
    n = 2 * T
    x8, ts = torch.randn(n, 8, H, W, generator=g), torch.randn(n, generator=g)
    ctx, y = torch.randn(n, 1, 1024, generator=g), torch.randn(n, 768, generator=g)
    with use_backend(EmulOps("cpu", exact=True)):
        net = VideoUNet(**cfg).eval()
        net.load_state_dict(synth.seeded_state_dict(net, 22))
        svt = net.packed().input_stages[1][1][1]
        assert svt.s_ff.w1_ln_fused is not None and svt.t_ff.w1_ln_fused is not None and svt.t_ff_in.w1_ln_fused is None
        monkeypatch.setattr(unet_engine, "_LN_FF", True)
        a = net(x8, ts, context=ctx, y=y, num_video_frames=T, image_only_indicator=torch.zeros(2, T))
        monkeypatch.setattr(unet_engine, "_LN_FF", False)
        b = net(x8, ts, context=ctx, y=y, num_video_frames=T, image_only_indicator=torch.zeros(2, T))
    rel, cos = rel_cos(a, b)
    assert rel <= 1e-5 and cos >= 0.999999, (rel, cos)
