"""Shared builders for the tiny (topology-identical, model_channels=64) parity configuration of tests/golden."""
from __future__ import annotations

import torch

from oracle.gen_golden import TINY, ctx5_tokens, tiny_unet_inputs  # noqa: F401  (seeds / shapes of the committed fixtures)
from v3d_amd import synth

P = "v3d_amd.sgm.modules.diffusionmodules."


def build_unet(device="cpu"):
    from v3d_amd.sgm.modules.diffusionmodules.video_model import VideoUNet
    net = VideoUNet(**synth.unet_config(TINY["model_channels"]))
    net.load_state_dict(synth.seeded_state_dict(net, TINY["weight_seed"]), strict=True)
    return net.to(device).eval()


def build_encoder(device="cpu"):
    from v3d_amd.sgm.modules.diffusionmodules.model import Encoder
    enc = Encoder(**synth.encoder_config(TINY["vae_ch"]))
    enc.load_state_dict(synth.seeded_state_dict(enc, TINY["weight_seed"] + 2), strict=True)
    return enc.to(device).eval()


def encoder_image():
    g = torch.Generator().manual_seed(TINY["seed"] + 4)
    return torch.rand(2, 3, 64, 48, generator=g) * 2.0 - 1.0


def build_decoder(device="cpu"):
    from v3d_amd.sgm.modules.autoencoding.temporal_ae import VideoDecoder
    dec = VideoDecoder(**synth.decoder_config(TINY["vae_ch"]))
    dec.load_state_dict(synth.seeded_state_dict(dec, TINY["weight_seed"] + 1), strict=True)
    return dec.to(device).eval()


def build_sampler(T, steps=None, device="cpu", kind="euler_linear"):
    """kind: euler_linear (V3D_512), heun_central, euler_vanilla - the three (sampler, guider) pairs with reference fixtures."""
    from v3d_amd.sgm.modules.diffusionmodules import sampling
    p = TINY
    cls = sampling.HeunEDMSampler if kind.startswith("heun") else sampling.EulerEDMSampler
    if kind.endswith("vanilla"):
        guider = {"target": P + "guiders.VanillaCFG", "params": {"scale": p["max_scale"]}}
    else:
        name = "CentralPredictionGuider" if kind.endswith("central") else "LinearPredictionGuider"
        guider = {"target": P + "guiders." + name, "params": {"max_scale": p["max_scale"], "min_scale": p["min_scale"], "num_frames": T}}
    return cls(discretization_config={"target": P + "discretizer.EDMDiscretization", "params": {"sigma_max": p["sigma_max"]}},
               num_steps=steps or p["steps"], guider_config=guider, device=device)


SAMPLER_FIXTURES = [("euler_linear", "sample_z"), ("heun_central", "sample_z_heun_central"), ("euler_vanilla", "sample_z_euler_vanilla")]


def build_denoiser():
    from v3d_amd.sgm.modules.diffusionmodules.denoiser import Denoiser
    return Denoiser({"target": P + "denoiser_scaling.VScalingWithEDMcNoise"})


def decoder_latents(T, device="cpu"):
    g = torch.Generator().manual_seed(TINY["seed"] + 3)
    return torch.randn(T, 4, 8, 8, generator=g).to(device)


def to_dev(d, device):
    return {k: v.to(device) for k, v in d.items()}
