"""-m gpu: parity of the product path AT the headline configuration (BASELINE.json configs[1]) and over the whole sampler
loop, against the fp32 CPU oracle (oracle/sgm_oracle.py, pinned to the reference by tests/test_oracle_pinned.py).

  * test_headline_rollout_3_steps_vs_oracle  width 320, 36 images (cfg 2 x T 18), 64 x 64 latents: a 3-step EulerEDM rollout of the
                                           benchmark's own network, every denoiser call teacher-forced against the oracle (3-D GroupNorm
                                           over 18 frames, 18-token temporal attention at 8192 x 5 problems, the persistent 147456-row
                                           GEMM tiles) + the final latent.
  * test_headline_midschedule_eval_vs_oracle  the same network teacher-forced at steps 8 / 14 / 20 of the 25-step schedule (sigma 70.5 / 6.35 /
                                           0.157, where the network output carries the denoised image), one sample of the doubled batch each.
  * test_headline_decoder_T18_vs_oracle    the decode half at its own size: full-width VideoDecoder, T = 18, 64 x 64 -> 512 x 512.
  * test_rollout_25_steps_cosine_and_psnr  SURVEY.md 8d end-to-end bar at width 64: T = 18, 64 x 64 latents, 25 EulerEDM steps,
                                           LinearPredictionGuider 4.5, DiffusionEngine.decode_first_stage -> 512 x 512 frames:
                                           latent cosine >= 0.99 and decoded PSNR >= 35 dB; both numbers are printed and recorded.
  * test_sampler_steps_teacher_forced      every denoiser call of the tiny 3-step samplers re-evaluated by the oracle ON THE HIP
                                           TRAJECTORY'S OWN INPUTS (one-evaluation tolerance, so a 10 % kernel error inside the loop
                                           cannot hide behind the guidance amplification), the guider / Euler / Heun arithmetic
                                           recomputed in fp32 from the recorded tensors, and the HIP run compared with the bf16-mode
                                           emulator (same rounding points, torch arithmetic).
  * test_decode_first_stage_chunked        DiffusionEngine.decode_first_stage with en_and_decode_n_samples_a_time = 2 on T = 3
                                           (reference video_diffusion.py:182-210: chunks see only their own frames) and the 5-D input
                                           form; encode_first_stage with chunking.
  * test_graph_replay_matches_eager        hipGraph capture of an evaluation replayed three times == eager (GroupNorm partial sums
                                           are re-zeroed inside the graph).
Tolerances: one evaluation cos >= 0.999 and max|err|/max|ref| <= 4e-2; 25-step latents cos >= 0.99; PSNR >= 35 dB.
"""
import time

import pytest
import torch

from conftest import device_oracle, full_inputs, odev, psnr, record_parity, rel_cos
from tiny import SAMPLER_FIXTURES, TINY, build_decoder, build_denoiser, build_sampler, build_unet, decoder_latents, tiny_unet_inputs, to_dev
from v3d_amd import configs, synth
from v3d_amd.sgm.modules.diffusionmodules.wrappers import OpenAIWrapper
from v3d_amd.sgm.util import instantiate_from_config

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
DEV = "cuda"


# (first in the file: a timeout can never cut the test that pins the checker every full-width test below relies on - VERDICT r5 item 9)
def test_device_oracle_is_pinned(golden):
    """The checker of the full-width tests is oracle/sgm_oracle.py executed on the GPU (conftest.device_oracle: fp32 ATen kernels, MIOpen and
    TF32 off, math SDPA).  Pinned here the way the CPU-executed oracle is pinned by tests/test_oracle_pinned.py: against the fixtures the
    REFERENCE's own modules generated (tests/golden/v3d_tiny.pt, rtol 1e-4 / atol 1e-5 per SURVEY.md 8d - relative to the tensor's scale) and
    against the CPU-executed oracle on the same inputs, for the U-Net (both image_only_indicator cases), the 3-step sampler and the decoder."""
    from oracle import sgm_oracle as O
    p = TINY
    T = p["T"]
    noise, c, uc, x8, ts, ctx, y = tiny_unet_inputs(T, p["H"], p["W"], p["seed"])
    net, dec = build_unet("cpu"), build_decoder("cpu")
    sd = {k: v.float() for k, v in net.state_dict().items()}
    dsd = {k: v.float() for k, v in dec.state_dict().items()}
    ucfg, dcfg = synth.unet_config(p["model_channels"]), synth.decoder_config(p["vae_ch"])
    ioi = torch.zeros(2, T)
    z = decoder_latents(T)

    def run(dev):
        s_, d_, i_ = odev(sd, dev), odev(dsd, dev), ioi.to(dev)
        i1 = i_.clone()
        i1[1, 1] = 1.0
        unet = lambda a, b, cc, d, ind=i_: O.unet_forward(s_, ucfg, a, b, cc, d, T, ind)
        return {"unet_out": unet(*odev((x8, ts, ctx, y), dev)).cpu(), "unet_out_ioi": unet(*odev((x8, ts, ctx, y), dev), ind=i1).cpu(),
                "sample_z": O.sample_euler_edm(unet, *odev((noise.clone(), c, uc), dev), p["steps"], T, p["min_scale"], p["max_scale"], p["sigma_max"]).cpu(),
                "dec_out": O.decoder_forward(d_, dcfg, odev(z, dev), T).cpu()}

    with device_oracle() as od:
        assert od == "cuda", "the -m gpu suite runs its checker on the GPU (V3D_ORACLE_DEVICE=cpu is a debugging knob)"
        on_gpu = run(od)
    on_cpu = run("cpu")
    worst = {}
    for k, v in on_gpu.items():
        scale = golden[k].abs().max().item()
        e_ref = ((v - golden[k]).abs().max() / scale).item()
        e_cpu = ((v - on_cpu[k]).abs().max() / scale).item()
        worst[k] = (round(e_ref, 7), round(e_cpu, 7))
        # (the sampler fixture carries the guidance / 1 / sigma amplification of three evaluations: 1e-3 like tests/test_oracle_pinned.py)
        tol = 1e-3 if k == "sample_z" else 1e-4
        assert e_ref <= tol and e_cpu <= tol, (k, worst)
    record_parity("device_oracle_pin", {k: {"vs_reference_fixture": a, "vs_cpu_oracle": b} for k, (a, b) in worst.items()})


def test_headline_rollout_3_steps_vs_oracle(full_unet):
    """BASELINE.json configs[1] sizes for the whole sampler stack: width 320, T = 18, 64 x 64 latents, cfg-doubled batch of 36 images,
    3 EulerEDM steps x LinearPredictionGuider (4.5) x Denoiser x OpenAIWrapper on the HIP kernels.
      (1) every denoiser call of the loop re-evaluated by the fp32 oracle ON THE HIP TRAJECTORY'S OWN INPUTS (teacher-forced: the
          one-evaluation tolerance at the headline size, at three different noise levels - the first call is the single-evaluation
          check rounds 1-2 recorded as `headline_unet_eval`);
      (2) the final latent against the fp32 guider / Euler arithmetic applied to the ORACLE's outputs of those calls: what the rollout
          would have produced had every evaluation been exact (the oracle's own 3-step trajectory would cost three more 2-minute
          evaluations for the same information: the two differ only through the inputs of calls 2 and 3, which (1) already holds to
          the per-call bound)."""
    from oracle import sgm_oracle as O
    from tiny import build_denoiser
    from v3d_amd.sgm.modules.diffusionmodules import sampling
    T, H, W, steps, scale = 18, 64, 64, 3, 4.5
    P = "v3d_amd.sgm.modules.diffusionmodules."
    sampler = sampling.EulerEDMSampler(discretization_config={"target": P + "discretizer.EDMDiscretization", "params": {"sigma_max": 700.0}},
                                       num_steps=steps, guider_config={"target": P + "guiders.LinearPredictionGuider",
                                                                       "params": {"max_scale": scale, "min_scale": scale, "num_frames": T}}, device=DEV)
    den, wr = build_denoiser(), OpenAIWrapper(full_unet)
    noise, c, uc = synth.synthetic_conditioning(T, H, W, seed=29)
    extra = {"image_only_indicator": torch.zeros(2, T, device=DEV), "num_video_frames": T}
    calls = []

    def denoiser(i, s, cc):
        out = den(wr, i, s, cc, **extra)
        calls.append((i.detach().float().cpu().clone(), s.detach().float().cpu().clone(), {k: v.detach().float().cpu() for k, v in cc.items()},
                      out.detach().float().cpu().clone()))
        return out

    z = sampler(denoiser, noise.clone().to(DEV), cond=to_dev(c, DEV), uc=to_dev(uc, DEV)).float().cpu()
    assert len(calls) == steps and torch.isfinite(z).all()
    ucfg = synth.unet_config(320)
    sigmas = O.edm_sigmas(steps, sigma_max=700.0)
    gscale = O.guider_scale("linear", T, scale, scale)
    x = noise.clone() * torch.sqrt(1.0 + sigmas[0] ** 2.0)
    per_call, t0 = [], time.time()
    refs = []
    with device_oracle() as od:                    # the fp32 oracle as ATen kernels on the GPU (conftest.device_oracle; pinned by test_device_oracle_is_pinned)
        sd, ioi = odev(full_unet.state_dict(), od), torch.zeros(2, T, device=od)
        for inp, sig, cc, _ in calls:
            refs.append(O.denoise(lambda x8, cn, ctx, vec: O.unet_forward(sd, ucfg, x8, cn, ctx, vec, T, ioi), *odev((inp, sig, cc), od)).cpu())
        del sd
    for i, (inp, sig, cc, out) in enumerate(calls):
        ref = refs[i]
        rel, cos = rel_cos(out, ref)
        per_img = min(rel_cos(out[k], ref[k])[1] for k in range(2 * T))
        per_call.append({"sigma": round(float(sig[0]), 4), "max_rel_err": round(rel, 5), "cosine": round(cos, 6), "min_per_image_cosine": round(per_img, 6)})
        # the exact-evaluation trajectory: guider + Euler update in fp32 on the oracle's output
        xu, xc = ref.chunk(2)
        d = (x - (xu + gscale.reshape(-1, 1, 1, 1) * (xc - xu))) / sigmas[i]
        x = x + (sigmas[i + 1] - sigmas[i]) * d
    dt = time.time() - t0
    rel_z, cos_z = rel_cos(z, x)
    record_parity("headline_rollout_3_steps", {"images": 2 * T, "T": T, "width": 320, "latent": [H, W], "steps": steps, "cfg_scale": scale,
                                               "per_call_teacher_forced": per_call, "final_latent_cosine": round(cos_z, 6),
                                               "final_latent_max_rel_err": round(rel_z, 5), "oracle_seconds": round(dt, 1),
                                               "oracle_device": od})
    record_parity("headline_unet_eval", dict(per_call[0], images=2 * T, T=T, width=320, latent=[H, W], note="call 1 of headline_rollout_3_steps"))
    for pc in per_call:
        assert pc["max_rel_err"] <= 4e-2 and pc["cosine"] >= 0.999 and pc["min_per_image_cosine"] >= 0.998, per_call
    # the guidance combination multiplies a per-evaluation error by up to 1 + 2 * 4.5: the latent bound is the sampler bar of SURVEY 8d
    assert cos_z >= 0.999 and rel_z <= 0.1, (rel_z, cos_z)


@pytest.mark.parametrize("step", [8, 14, 20])
def test_headline_midschedule_eval_vs_oracle(full_unet, step):
    """Teacher-forced full-width evaluations at MID-SCHEDULE noise levels of the headline 25-step schedule (the 3-step rollout
    above only visits sigma = 700, 15.6 and 0.002 - at the last one c_skip ~ 1 and the network output hardly matters).  Steps 8 / 14 / 20 of
    EDMDiscretization(25 steps, sigma_max 700) are sigma = 70.5 / 6.35 / 0.157: c_out * F(x) carries most of the denoised image there.
    The HIP path evaluates the cfg-doubled 36-image batch exactly as the sampler would (Denoiser x OpenAIWrapper x VideoUNet at width 320,
    T = 18, 64 x 64); the fp32 oracle re-evaluates ONE of the two samples of that batch on the same inputs (the unconditional half at steps
    8 and 20, the conditional one at step 14: samples do not interact, so half the batch is an exact check of that half at half the CPU
    time, ~60 s).  Input: a unit-variance latent + sigma * noise, the state the sampler holds at that step."""
    from oracle import sgm_oracle as O
    T, H, W = 18, 64, 64
    sig = float(O.edm_sigmas(25, sigma_max=700.0)[step])
    noise, c, uc = synth.synthetic_conditioning(T, H, W, seed=31 + step)
    g = torch.Generator().manual_seed(100 + step)
    x = torch.randn(T, 4, H, W, generator=g) + sig * noise
    xin = torch.cat([x, x])
    s_in = torch.full((2 * T,), sig)
    cc = {k: torch.cat([uc[k], c[k]]) for k in c}
    den, wr = build_denoiser(), OpenAIWrapper(full_unet)
    extra = {"image_only_indicator": torch.zeros(2, T, device=DEV), "num_video_frames": T}
    out = den(wr, xin.to(DEV), s_in.to(DEV), to_dev(cc, DEV), **extra).float().cpu()
    assert out.shape == (2 * T, 4, H, W) and torch.isfinite(out).all()
    ucfg = synth.unet_config(320)
    t0 = time.time()
    with device_oracle() as od:                    # (round 5: the checker runs on the GPU, so the WHOLE doubled batch is checked, not one half of it)
        sd, ioi = odev(full_unet.state_dict(), od), torch.zeros(2, T, device=od)
        ref = O.denoise(lambda x8, cn, ctx, vec: O.unet_forward(sd, ucfg, x8, cn, ctx, vec, T, ioi), *odev((xin, s_in, cc), od)).cpu()
        del sd
    dt = time.time() - t0
    rel, cos = rel_cos(out, ref)
    per_img = min(rel_cos(out[k], ref[k])[1] for k in range(2 * T))
    # how much of the denoised output is the network's (vs c_skip * x): ||c_out F|| / ||denoised||
    c_skip = 1.0 / (sig * sig + 1.0)
    net_share = float((ref - c_skip * xin).norm() / ref.norm())
    record_parity(f"headline_midschedule_eval_step{step}", {"sigma": round(sig, 4), "images_checked": 2 * T, "width": 320,
                                                           "latent": [H, W], "max_rel_err": round(rel, 5), "cosine": round(cos, 6),
                                                           "min_per_image_cosine": round(per_img, 6), "network_share_of_output": round(net_share, 4),
                                                           "oracle_seconds": round(dt, 1), "oracle_device": od})
    assert rel <= 4e-2 and cos >= 0.999 and per_img >= 0.998, (step, sig, rel, cos, per_img)


def test_headline_decoder_T18_vs_oracle():
    """The decode half of the headline benchmark at its own size: full-width VideoDecoder (128 base channels), T = 18 frames,
    64 x 64 latents -> 512 x 512 - the 3-D GroupNorm over 18 x 512^2 x 128 channels (6 x 10^8 elements per group: statistics in
    slotted fp32 partials, added up in fp64) and the 18-frame temporal convolutions at 4.7 M rows - against the fp32 oracle."""
    from oracle import sgm_oracle as O
    from v3d_amd.sgm.modules.autoencoding.temporal_ae import VideoDecoder
    T = 18
    with torch.device(DEV):
        dec = VideoDecoder(**synth.decoder_config(128)).eval()
    synth.init_module_fast(dec, seed=2)
    z = torch.randn(T, 4, 64, 64, generator=torch.Generator().manual_seed(6))
    out = dec(z.to(DEV), timesteps=T).float().cpu()
    assert out.shape == (T, 3, 512, 512) and torch.isfinite(out).all()
    again = dec(z.to(DEV), timesteps=T).float().cpu()
    assert torch.equal(out, again), "two identical decodes differ: the decoder is not deterministic"
    t0 = time.time()
    with device_oracle() as od:
        ref = O.decoder_forward(odev(dec.state_dict(), od), synth.decoder_config(128), odev(z, od), T).cpu()
    dt = time.time() - t0
    rel, cos = rel_cos(out, ref)
    db = psnr(out, ref)
    record_parity("headline_decoder_T18", {"T": T, "latent": [64, 64], "frames": [512, 512], "vae_ch": 128, "cosine": round(cos, 6),
                                           "max_rel_err": round(rel, 5), "psnr_db": round(db, 2), "oracle_seconds": round(dt, 1)})
    assert rel <= 4e-2 and cos >= 0.999, (rel, cos)


# ---- BASELINE.json configs[1] against the REFERENCE'S OWN MODULES at full width (tests/golden/v3d_full.pt, oracle/gen_golden_full.py) -------------
@pytest.fixture(scope="module")
def golden_full():
    import os
    from conftest import ROOT
    return torch.load(os.path.join(ROOT, "tests", "golden", "v3d_full.pt"))


@pytest.fixture(scope="module")
def ref_engine(golden_full):
    """DiffusionEngine at the headline width with the weights the fixture was generated with: seeded_state_dict is drawn on the CPU generator per
    tensor NAME, so the reference modules in the build container and this engine on the GPU box hold bit-identical fp32 weights."""
    p = golden_full["params"]
    cfg = configs.v3d_512_config(num_frames=p["T"], num_steps=p["steps"], min_scale=p["scale"], max_scale=p["scale"], sigma_max=p["sigma_max"],
                                 model_channels=p["model_channels"], vae_ch=p["vae_ch"])["model"]
    with torch.device(DEV):
        eng = instantiate_from_config(cfg).eval()
    unet, dec = eng.model.diffusion_model, eng.first_stage_model.decoder
    unet.load_state_dict(synth.seeded_state_dict(unet, p["unet_seed"]), strict=True)
    dec.load_state_dict(synth.seeded_state_dict(dec, p["dec_seed"]), strict=True)
    assert abs(eng.scale_factor - p["scale_factor"]) < 1e-12
    return eng


@pytest.mark.parametrize("call", [0, 8, 14, 20])
def test_headline_eval_vs_reference_modules(golden_full, ref_engine, call):
    """One full-width denoiser evaluation (Denoiser x OpenAIWrapper x VideoUNet, 36 images, 64 x 64) on the HIP kernels against the output
    of the REFERENCE's unmodified modules for the same call of its own 25-step rollout (teacher-forced on the reference's state x_k):
    headline parity against the reference itself, not against the port.  The targets are stored in fp16 (2^-11
    relative: 40x under the bound)."""
    p = golden_full["params"]
    T = p["T"]
    _, c, uc = synth.synthetic_conditioning(T, p["H"], p["W"], seed=p["cond_seed"])
    x = golden_full[f"call{call}_x"]
    sig = golden_full["meta"]["sigmas"][call]
    xin, s_in = torch.cat([x, x]), torch.full((2 * T,), sig)
    cc = {k: torch.cat([uc[k], c[k]]) for k in c}
    extra = {"image_only_indicator": torch.zeros(2, T, device=DEV), "num_video_frames": T}
    out = ref_engine.denoiser(ref_engine.model, xin.to(DEV), s_in.to(DEV), to_dev(cc, DEV), **extra).float().cpu()
    ref = golden_full[f"call{call}_out"].float()
    rel, cos = rel_cos(out, ref)
    per_img = min(rel_cos(out[k], ref[k])[1] for k in range(2 * T))
    record_parity(f"headline_eval_vs_reference_call{call}", {"sigma": round(sig, 4), "images": 2 * T, "width": 320, "max_rel_err": round(rel, 5),
                                                            "cosine": round(cos, 6), "min_per_image_cosine": round(per_img, 6)})
    assert rel <= 4e-2 and cos >= 0.999 and per_img >= 0.998, (call, sig, rel, cos, per_img)


def test_headline_rollout_25_steps_vs_reference_modules(golden_full, ref_engine):
    """SURVEY.md 8d end-to-end bar AT THE HEADLINE WIDTH: the 25-step EulerEDM x LinearPredictionGuider(4.5) rollout of the
    width-320 network on 18 x 4 x 64 x 64 latents + decode_first_stage to 512 x 512, on the HIP kernels, against the same run by the
    reference's own modules in fp32 (sampling.py:112-133, video_diffusion.py:182-210): final latent cosine >= 0.99, decoded frames PSNR >= 35 dB."""
    p = golden_full["params"]
    T = p["T"]
    noise, c, uc = synth.synthetic_conditioning(T, p["H"], p["W"], seed=p["cond_seed"])
    eng = ref_engine
    extra = {"image_only_indicator": torch.zeros(2, T, device=DEV), "num_video_frames": T}
    z = eng.sampler(lambda i, s, cc: eng.denoiser(eng.model, i, s, cc, **extra), noise.clone().to(DEV), cond=to_dev(c, DEV), uc=to_dev(uc, DEV))
    frames = eng.decode_first_stage(z).float().cpu()
    rel_z, cos_z = rel_cos(z, golden_full["z"])
    kept = list(p["frames_kept"])
    f_ref = golden_full["frames"].float()
    rel_f, cos_f = rel_cos(frames[kept], f_ref)
    db = psnr(frames[kept], f_ref)
    stats = torch.stack([frames.mean(dim=(1, 2, 3)), frames.std(dim=(1, 2, 3))], dim=1)
    stat_err = (stats - golden_full["frame_stats"]).abs().max().item()
    # the decoder alone, on the REFERENCE's final latent
    f_dec = eng.decode_first_stage(golden_full["z"].to(DEV)).float().cpu()
    db_dec = psnr(f_dec[kept], f_ref)
    record_parity("rollout_25_steps_width320_vs_reference", {
        "T": T, "latent": [p["H"], p["W"]], "steps": p["steps"], "cfg_scale": p["scale"], "latent_cosine": round(cos_z, 6), "latent_max_rel_err": round(rel_z, 5),
        "frames_checked": kept, "frames_cosine": round(cos_f, 6), "frames_psnr_db": round(db, 2), "decoder_only_psnr_db": round(db_dec, 2),
        "all_frames_mean_std_max_abs_err": round(stat_err, 5), "reference_cpu_seconds": golden_full["meta"]["sampler_seconds"] + golden_full["meta"]["decode_seconds"]})
    assert cos_z >= 0.99, (rel_z, cos_z)
    assert db >= 35.0 and db_dec >= 35.0, (db, db_dec)
    assert stat_err <= 0.05, stat_err


def _engine(T, steps, scale, mc=64, vae_ch=32, **kw):
    cfg = configs.v3d_512_config(num_frames=T, num_steps=steps, min_scale=scale, max_scale=scale, model_channels=mc, vae_ch=vae_ch)["model"]
    cfg["params"].update(kw)
    eng = instantiate_from_config(cfg).eval()
    eng.load_state_dict(synth.seeded_state_dict(eng, 1234), strict=True)
    return eng.to(DEV)


def test_rollout_25_steps_cosine_and_psnr():
    from oracle import sgm_oracle as O
    T, H, W, steps, scale = 18, 64, 64, 25, 4.5
    eng = _engine(T, steps, scale)
    noise, c, uc = synth.synthetic_conditioning(T, H, W, seed=23)
    extra = {"image_only_indicator": torch.zeros(2, T, device=DEV), "num_video_frames": T}
    z = eng.sampler(lambda i, s, cc: eng.denoiser(eng.model, i, s, cc, **extra), noise.clone().to(DEV), cond=to_dev(c, DEV), uc=to_dev(uc, DEV))
    frames = eng.decode_first_stage(z)
    assert frames.shape == (T, 3, 8 * H, 8 * W)
    # ---- fp32 oracle of the same run ----
    ucfg, dcfg = synth.unet_config(64), synth.decoder_config(32)
    t0 = time.time()
    with device_oracle() as od:
        usd, dsd = odev(eng.model.diffusion_model.state_dict(), od), odev(eng.first_stage_model.decoder.state_dict(), od)
        ioi = torch.zeros(2, T, device=od)
        z_ref = O.sample_edm(lambda x8, cn, ctx, vec: O.unet_forward(usd, ucfg, x8, cn, ctx, vec, T, ioi), odev(noise.clone(), od), odev(c, od), odev(uc, od),
                             steps, T, scale, scale, 700.0)
        t_samp = time.time() - t0
        f_ref = O.decode_first_stage(dsd, dcfg, z_ref, eng.scale_factor, T).cpu()
        z_ref = z_ref.cpu()
    rel_z, cos_z = rel_cos(z, z_ref)
    rel_f, cos_f = rel_cos(frames, f_ref)
    db = psnr(frames, f_ref)
    # decoder alone on the oracle's latents (separates the decoder's own error from the sampler's)
    f_dec = eng.decode_first_stage(z_ref.to(DEV))
    db_dec = psnr(f_dec, f_ref)
    # the uint8 frames a consumer sees (V3D_512.py:286-303): clamp((x+1)/2) * 255
    to8 = lambda t: (torch.clamp((t.float().cpu() + 1.0) / 2.0, 0.0, 1.0) * 255).to(torch.uint8).float()
    mse8 = ((to8(frames) - to8(f_ref)) ** 2).mean().item()
    import math
    db8 = float("inf") if mse8 == 0 else 10.0 * math.log10(255.0 ** 2 / mse8)
    record_parity("rollout_25_steps_width64", {
        "T": T, "latent": [H, W], "steps": steps, "cfg_scale": scale, "latent_cosine": round(cos_z, 6), "latent_max_rel_err": round(rel_z, 5),
        "frames_cosine": round(cos_f, 6), "frames_psnr_db": round(db, 2), "frames_psnr_db_uint8": round(db8, 2),
        "decoder_only_psnr_db": round(db_dec, 2), "oracle_sampler_seconds": round(t_samp, 1)})
    assert cos_z >= 0.99, (rel_z, cos_z)
    assert db >= 35.0, db


@pytest.mark.parametrize("kind,key", SAMPLER_FIXTURES)
def test_sampler_steps_teacher_forced(golden, kind, key):
    """Why `sampler_heun_central` sits at cosine 0.9964 / max rel 0.125 against the reference fixture while every other record is >= 0.999
    (VERDICT r3): nothing in that run is less accurate per evaluation - part (1) below holds every one of its 5 denoiser calls to the
    one-evaluation bound (<= 4e-2, cosine >= 0.999) on the trajectory's own inputs, part (2) holds the guider / Heun arithmetic to 2e-5.  The
    trajectory-level number is that per-call error times the amplification of the sampler: the CentralPredictionGuider scales (c - uc) by up
    to 2 * max_scale = 7 at the middle frames (a per-evaluation error e becomes up to (1 + 2 * 7) e in the guided prediction), and Heun's
    corrector divides a difference of two nearly equal states by next_sigma (|dt| / (2 next_sigma) = 3900 for the 15.59 -> 0.002 step of the
    3-step tiny schedule).  The bf16-mode EMULATOR (torch arithmetic, same rounding points) sits at the same distance from the fp32 fixture
    (rel 0.12 / cosine 0.9964), and the HIP run is compared with it in part (3): the number is a property of bf16 x this sampler, not of a kernel."""
    from oracle import sgm_oracle as O
    from oracle.ops_emul import EmulOps
    from v3d_amd.ops import use_backend
    p = TINY
    T = p["T"]
    noise, c, uc, *_ = tiny_unet_inputs(T, p["H"], p["W"], p["seed"])
    net = build_unet(DEV)
    sampler, den, wr = build_sampler(T, device=DEV, kind=kind), build_denoiser(), OpenAIWrapper(net)
    extra = {"image_only_indicator": torch.zeros(2, T, device=DEV), "num_video_frames": T}
    calls = []

    def denoiser(i, s, cc):
        out = den(wr, i, s, cc, **extra)
        calls.append((i.detach().float().cpu().clone(), s.detach().float().cpu().clone(), {k: v.detach().float().cpu() for k, v in cc.items()},
                      out.detach().float().cpu().clone()))
        return out

    z = sampler(denoiser, noise.clone().to(DEV), cond=to_dev(c, DEV), uc=to_dev(uc, DEV))
    assert len(calls) == (2 * p["steps"] - 1 if kind.startswith("heun") else p["steps"])
    # (1) every network evaluation of the loop, on its own inputs, against the fp32 oracle: one-evaluation tolerance
    sd = {k: v.float().cpu() for k, v in net.state_dict().items()}
    ucfg = synth.unet_config(p["model_channels"])
    ioi = torch.zeros(2, T)
    worst = (0.0, 1.0)
    per_call = []
    for inp, sig, cc, out in calls:
        ref = O.denoise(lambda x8, cn, ctx, vec: O.unet_forward(sd, ucfg, x8, cn, ctx, vec, T, ioi), inp, sig, cc)
        rel, cos = rel_cos(out, ref)
        per_call.append((round(float(sig[0]), 4), round(rel, 5), round(cos, 6)))
        worst = (max(worst[0], rel), min(worst[1], cos))
    print(f"[teacher-forced {kind}] (sigma, rel, cos) per denoiser call: {per_call}")
    assert worst[0] <= 4e-2 and worst[1] >= 0.999, (kind, per_call)
    # (2) the elementwise sampler arithmetic (guider + Euler / Heun update), recomputed in fp32 from the recorded denoiser outputs
    scale = O.guider_scale({"euler_linear": "linear", "heun_central": "central", "euler_vanilla": "vanilla"}[kind], T, p["min_scale"], p["max_scale"])
    sigmas = O.edm_sigmas(p["steps"], sigma_max=p["sigma_max"])

    def guided(out):
        xu, xc = out.chunk(2)
        return xu + scale.repeat(xu.shape[0] // T).reshape(-1, 1, 1, 1) * (xc - xu)

    x = noise.clone() * torch.sqrt(1.0 + sigmas[0] ** 2.0)
    prev_max = float(x.abs().max())
    amp = 1.0
    it = iter(calls)
    for i in range(p["steps"]):
        inp, sig, _, out = next(it)
        # (x_{i+1} = x + dt d cancels a state of magnitude |x_i| down to |x_{i+1}|: fp32 rounding of the LARGER state is the honest bound)
        assert torch.allclose(inp[: x.shape[0]], x, rtol=1e-4, atol=4e-6 * prev_max * amp), (i, (inp[: x.shape[0]] - x).abs().max().item(), prev_max, amp)
        prev_max = float(x.abs().max())
        # Heun's d_new = (euler - denoised2) / next_sigma divides a difference of nearly equal fp32 numbers by next_sigma: the rounding of
        # that difference comes back multiplied by |dt| / (2 next_sigma) (7800 / 2 for the 15.59 -> 0.002 step) - in the reference too
        amp = 1.0
        if kind.startswith("heun") and float(sigmas[i + 1]) > 0:
            amp += float((sigmas[i] - sigmas[i + 1]) / sigmas[i + 1]) / 2.0
        d = (x - guided(out)) / sigmas[i]
        euler = x + (sigmas[i + 1] - sigmas[i]) * d
        if kind.startswith("heun") and float(sigmas[i + 1]) > 0:
            _, _, _, out2 = next(it)
            d2 = (euler - guided(out2)) / sigmas[i + 1]
            x = x + (sigmas[i + 1] - sigmas[i]) * (d + d2) / 2.0
        else:
            x = euler
    rel_arith = ((z.float().cpu() - x).abs().max() / x.abs().max()).item()
    assert rel_arith <= 2e-5, rel_arith
    # (3) the same loop on the bf16-mode emulator (identical rounding points, torch arithmetic on the GPU)
    with use_backend(EmulOps(DEV, exact=False)):
        net_e = build_unet(DEV)
        wr_e = OpenAIWrapper(net_e)
        sampler_e = build_sampler(T, device=DEV, kind=kind)
        z_e = sampler_e(lambda i, s, cc: den(wr_e, i, s, cc, **extra), noise.clone().to(DEV), cond=to_dev(c, DEV), uc=to_dev(uc, DEV))
    rel_e, cos_e = rel_cos(z, z_e)
    rel_g, cos_g = rel_cos(z, golden[key])
    record_parity(f"sampler_{kind}", {"worst_eval_rel": round(worst[0], 5), "worst_eval_cos": round(worst[1], 6), "arith_rel": rel_arith,
                                      "vs_bf16_emulator_rel": round(rel_e, 5), "vs_bf16_emulator_cos": round(cos_e, 6),
                                      "vs_reference_fixture_rel": round(rel_g, 5), "vs_reference_fixture_cos": round(cos_g, 6)})
    # two bf16 implementations with different rounding order differ like each differs from fp32 (the guidance combination multiplies the
    # per-evaluation error by up to 1 + 2 scale; Heun x Central runs 5 evaluations at scales up to 7): the emulator itself sits at
    # rel 0.12 / cosine 0.9964 from the fp32 fixture there
    # measured (profiles/r05_parity.json / r06_parity.json): heun_central 0.99416 / 0.99373 vs the emulator, 0.99622 / 0.99599 vs the reference's fp32
    # fixture (the kernels' rounding points moved between the rounds); the others 0.9996 / 0.9997
    assert cos_e >= (0.99 if kind == "heun_central" else 0.995), (rel_e, cos_e)
    # against the REFERENCE's own fixture: 0.995 for the Euler samplers; Heun x Central sits at 0.9960 - 0.9962 and is held to 0.994 (a 0.995 bar would
    # be inside the run-to-run spread of a 5-evaluation chain at guidance scales up to 7)
    assert cos_g >= (0.994 if kind == "heun_central" else 0.995), (kind, rel_g, cos_g)


def test_decode_first_stage_chunked():
    from oracle import sgm_oracle as O
    T = 3
    eng = _engine(T, 2, 2.0, en_and_decode_n_samples_a_time=2)
    dsd = {k: v.detach().float().cpu() for k, v in eng.first_stage_model.decoder.state_dict().items()}
    dcfg = synth.decoder_config(32)
    z = decoder_latents(T) * 0.18215
    out = eng.decode_first_stage(z.to(DEV))
    ref = O.decode_first_stage(dsd, dcfg, z, eng.scale_factor, 2)          # chunks of 2 + 1 frames, timesteps = len(chunk)
    rel, cos = rel_cos(out, ref)
    assert out.shape == ref.shape and rel <= 4e-2 and cos >= 0.999, (rel, cos)
    full = O.decode_first_stage(dsd, dcfg, z, eng.scale_factor, T)
    assert (ref - full).abs().max() > 1e-2                                  # chunking changes the result (Appendix B-13) ...
    rel_full, _ = rel_cos(out, full)
    assert rel_full > rel * 2                                                # ... and the product follows the chunked semantics
    out5 = eng.decode_first_stage(z.reshape(1, T, 4, 8, 8).to(DEV))          # "b t c h w" input form
    assert out5.shape == (1, T, 3, 64, 64)
    assert torch.equal(out5[0], out)                 # same work decomposition, no atomics anywhere (round 3): bit-equal
    record_parity("decode_first_stage_chunked", {"T": T, "chunk": 2, "max_rel_err": round(rel, 5), "cosine": round(cos, 6)})
    # encode_first_stage with chunking (video_diffusion.py:212-238); input_key != "latents" runs the VAE encoder + regulariser, whose
    # posterior sample draws torch.randn(mean.shape) on the CPU generator per chunk (distributions.py:37-41)
    eng.input_key = "frames"
    esd = {k: v.detach().float().cpu() for k, v in eng.first_stage_model.encoder.state_dict().items()}
    img = torch.rand(3, 3, 64, 48, generator=torch.Generator().manual_seed(3)) * 2 - 1
    torch.manual_seed(5)
    zz = eng.encode_first_stage(img.to(DEV))
    torch.manual_seed(5)
    refs = []
    for i in range(0, 3, 2):
        mean, logvar = torch.chunk(O.encoder_forward(esd, synth.encoder_config(32), img[i:i + 2]), 2, dim=1)
        refs.append(mean + torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)) * torch.randn(mean.shape))
    zref = eng.scale_factor * torch.cat(refs, dim=0)
    rel, cos = rel_cos(zz, zref)
    assert zz.shape == zref.shape and rel <= 4e-2 and cos >= 0.999, (rel, cos)


def test_graph_replay_matches_eager():
    from v3d_amd.engine.graph import graphed
    p = TINY
    T = p["T"]
    _, _, _, x8, ts, ctx, y = tiny_unet_inputs(T, p["H"], p["W"], p["seed"])
    net = build_unet(DEV)
    ioi = torch.zeros(2, T, device=DEV)

    def ev(x, t, cx, yy):
        return net(x, t, context=cx, y=yy, num_video_frames=T, image_only_indicator=ioi)

    args = (x8.to(DEV), ts.to(DEV), ctx.to(DEV), y.to(DEV))
    eager = ev(*args).float().clone()
    g = graphed(ev, enabled=True)
    assert torch.equal(ev(*args).float(), eager), "two identical eager evaluations differ: the engine is not deterministic"
    for rep in range(3):
        # bit-equal (round 3): GroupNorm statistics are written one slot per writer with plain stores and added up in a fixed order, no
        # kernel on the path uses floating-point atomics.  (Rounds 1-2: fp32 atomics, two eager runs differed by 1-2e-2 max rel and this
        # test could only bound the replay at 4e-2.)  Stale partial sums (every GroupNorm seeing 2x, 3x .. the sums) are an O(1) error.
        assert torch.equal(g(*args).float(), eager), f"graph replay {rep} differs from eager"
    # other inputs through the same captured graph
    args2 = tuple(a * 0.5 for a in args)
    assert torch.equal(g(*args2).float(), ev(*args2).float())
    # the decoder graph (3-D GroupNorm statistics spanning T frames)
    dec = build_decoder(DEV)
    z = decoder_latents(T, DEV)
    gd = graphed(lambda zz: dec(zz, timesteps=T), enabled=True)
    want = dec(z, timesteps=T).float().clone()
    for rep in range(3):
        assert torch.equal(gd(z).float(), want), f"decoder graph replay {rep} differs from eager"
