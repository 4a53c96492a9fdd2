"""TEST INFRASTRUCTURE — generate tests/golden/v3d_tiny.pt by running the REFERENCE's own modules.

Run in the build container only (needs /root/reference):   python -m oracle.gen_golden
The reference modules are imported unmodified through oracle/ref_import.py, loaded with
v3d_amd.synth.seeded_state_dict weights and executed on seeded synthetic inputs (fp32, CPU, no autocast,
spatial_transformer_attn_type="softmax", sampler device="cpu" — the deviations SURVEY.md §8c lists as required for
the reference to run on CPU at all).  Only outputs are stored; inputs and weights are regenerated from seeds.
"""
from __future__ import annotations

import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402
from v3d_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "v3d_tiny.pt")

TINY = dict(model_channels=64, vae_ch=32, T=3, H=32, W=32, seed=7, weight_seed=11, steps=3, min_scale=1.5, max_scale=3.5,
            sigma_max=700.0)


def tiny_unet_inputs(T: int, H: int, W: int, seed: int):
    noise, c, uc = synth.synthetic_conditioning(T, H, W, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    x8 = torch.randn(2 * T, 8, H, W, generator=g)
    timesteps = torch.randn(2 * T, generator=g) * 0.7
    context = torch.cat([uc["crossattn"], c["crossattn"]], 0)
    y = torch.cat([uc["vector"], c["vector"]], 0)
    return noise, c, uc, x8, timesteps, context, y


def ctx5_tokens(T: int, seed: int):
    return torch.randn(2 * T, 5, 1024, generator=torch.Generator().manual_seed(seed + 5))


@torch.no_grad()
def main():
    torch.set_grad_enabled(False)
    m = ref_import.load()
    p = TINY
    T, H, W = p["T"], p["H"], p["W"]
    out = {"params": dict(p)}

    # ---- VideoUNet single evaluation ----
    ucfg = synth.unet_config(p["model_channels"], attn_type="softmax")
    net = m["video_model"].VideoUNet(**ucfg).eval()
    sd = synth.seeded_state_dict(net, p["weight_seed"])
    net.load_state_dict(sd, strict=True)
    noise, c, uc, x8, timesteps, context, y = tiny_unet_inputs(T, H, W, p["seed"])
    ioi0 = torch.zeros(2, T)
    out["unet_out"] = net(x8, timesteps, context=context, y=y, num_video_frames=T, image_only_indicator=ioi0).clone()
    ioi1 = ioi0.clone()
    ioi1[1, 1] = 1.0
    out["unet_out_ioi"] = net(x8, timesteps, context=context, y=y, num_video_frames=T, image_only_indicator=ioi1).clone()

    # ---- general cross-attention: a context of 5 tokens per image (attention.py:286-349 without the one-token shortcut every V3D / SVD
    #      configuration takes; round 5).  Tokens differ per image, so the temporal block's frame-0 context (video_attention.py:249-253) shows.
    out["unet_out_ctx5"] = net(x8, timesteps, context=ctx5_tokens(T, p["seed"]), y=y, num_video_frames=T, image_only_indicator=ioi0).clone()

    # ---- one VideoResBlock / one SpatialVideoTransformer in isolation (block-level pins) ----
    g = torch.Generator().manual_seed(p["seed"] + 2)
    xb = torch.randn(2 * T, 64, 16, 16, generator=g)
    emb = torch.randn(2 * T, 256, generator=g)
    rb = net.input_blocks[1][0]
    out["resblock_out"] = rb(xb, emb, T, ioi0).clone()
    st = net.input_blocks[1][1]
    out["svt_out"] = st(xb, context, None, T, ioi0).clone()

    # ---- sampler: EulerEDM x LinearPredictionGuider x Denoiser x OpenAIWrapper over the same U-Net ----
    sampler = m["sampling"].EulerEDMSampler(
        discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.EDMDiscretization", "params": {"sigma_max": p["sigma_max"]}},
        num_steps=p["steps"],
        guider_config={"target": "sgm.modules.diffusionmodules.guiders.LinearPredictionGuider",
                       "params": {"max_scale": p["max_scale"], "min_scale": p["min_scale"], "num_frames": T}},
        device="cpu")
    denoiser = m["denoiser"].Denoiser({"target": "sgm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"})
    wrapped = m["wrappers"].OpenAIWrapper(net)
    extra = {"image_only_indicator": ioi0, "num_video_frames": T}

    def den(inp, sigma, cc):
        return denoiser(wrapped, inp, sigma, cc, **extra)

    out["sample_z"] = sampler(den, noise.clone(), cond=c, uc=uc).clone()

    # ---- the other sampler / guiders behind the same API ----
    disc = {"target": "sgm.modules.diffusionmodules.discretizer.EDMDiscretization", "params": {"sigma_max": p["sigma_max"]}}
    heun = m["sampling"].HeunEDMSampler(
        discretization_config=disc, num_steps=p["steps"],
        guider_config={"target": "sgm.modules.diffusionmodules.guiders.CentralPredictionGuider",
                       "params": {"max_scale": p["max_scale"], "min_scale": p["min_scale"], "num_frames": T}},
        device="cpu")
    out["sample_z_heun_central"] = heun(den, noise.clone(), cond=c, uc=uc).clone()
    vanilla = m["sampling"].EulerEDMSampler(
        discretization_config=disc, num_steps=p["steps"],
        guider_config={"target": "sgm.modules.diffusionmodules.guiders.VanillaCFG", "params": {"scale": p["max_scale"]}},
        device="cpu")
    out["sample_z_euler_vanilla"] = vanilla(den, noise.clone(), cond=c, uc=uc).clone()

    # ---- VideoDecoder ----
    dcfg = synth.decoder_config(p["vae_ch"])
    dec = m["temporal_ae"].VideoDecoder(**dcfg).eval()
    dsd = synth.seeded_state_dict(dec, p["weight_seed"] + 1)
    dec.load_state_dict(dsd, strict=True)
    g = torch.Generator().manual_seed(p["seed"] + 3)
    z = torch.randn(T, 4, 8, 8, generator=g)
    out["dec_out"] = dec(z, timesteps=T).clone()
    out["dec_out_T1"] = dec(z[:1], timesteps=1).clone()

    # ---- VAE Encoder (moments) on a 64 x 48 image batch, + the AutoencodingEngine-style mode() ----
    ecfg = synth.encoder_config(p["vae_ch"])
    enc = m["model"].Encoder(**ecfg).eval()
    enc.load_state_dict(synth.seeded_state_dict(enc, p["weight_seed"] + 2), strict=True)
    g = torch.Generator().manual_seed(p["seed"] + 4)
    img = torch.rand(2, 3, 64, 48, generator=g) * 2.0 - 1.0
    out["enc_moments"] = enc(img).clone()

    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    torch.save(out, OUT)
    for k, v in out.items():
        if torch.is_tensor(v):
            print(f"{k:16s} {tuple(v.shape)} mean|x|={v.abs().mean():.4f} max|x|={v.abs().max():.4f}")
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
