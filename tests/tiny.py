"""Shared builders for the tiny (topology-identical, model_channels=64) parity configuration of tests/golden."""
from __future__ import annotations

import torch

from oracle.gen_golden import TINY, tiny_unet_inputs  # noqa: F401  (seeds / shapes of the committed fixtures)
from v3d_amd import synth

P = "v3d_amd.sgm.modules.diffusionmodules."


def build_unet(device="cpu"):
    from v3d_amd.sgm.modules.diffusionmodules.video_model import VideoUNet
    net = VideoUNet(**synth.unet_config(TINY["model_channels"]))
    net.load_state_dict(synth.seeded_state_dict(net, TINY["weight_seed"]), strict=True)
    return net.to(device).eval()


def build_decoder(device="cpu"):
    from v3d_amd.sgm.modules.autoencoding.temporal_ae import VideoDecoder
    dec = VideoDecoder(**synth.decoder_config(TINY["vae_ch"]))
    dec.load_state_dict(synth.seeded_state_dict(dec, TINY["weight_seed"] + 1), strict=True)
    return dec.to(device).eval()


def build_sampler(T, steps=None, device="cpu"):
    from v3d_amd.sgm.modules.diffusionmodules.sampling import EulerEDMSampler
    p = TINY
    return EulerEDMSampler(
        discretization_config={"target": P + "discretizer.EDMDiscretization", "params": {"sigma_max": p["sigma_max"]}},
        num_steps=steps or p["steps"],
        guider_config={"target": P + "guiders.LinearPredictionGuider",
                       "params": {"max_scale": p["max_scale"], "min_scale": p["min_scale"], "num_frames": T}},
        device=device)


def build_denoiser():
    from v3d_amd.sgm.modules.diffusionmodules.denoiser import Denoiser
    return Denoiser({"target": P + "denoiser_scaling.VScalingWithEDMcNoise"})


def decoder_latents(T, device="cpu"):
    g = torch.Generator().manual_seed(TINY["seed"] + 3)
    return torch.randn(T, 4, 8, 8, generator=g).to(device)


def to_dev(d, device):
    return {k: v.to(device) for k, v in d.items()}
