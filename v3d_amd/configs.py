"""Configuration dictionaries for the V3D dense-multi-view path, in the reference's `target:` / `params:` format.

`v3d_512_config()` restates the inference-relevant content of the reference's scripts/pub/configs/V3D_512.yaml with every
target already pointing at v3d_amd.sgm.*; `load_reference_yaml()` takes the user's own copy of a reference YAML and
remaps its targets (the drop-in switch is exactly that rewrite, see v3d_amd.sgm.util.remap_targets).
"""
from __future__ import annotations

from . import synth
from .sgm.util import remap_targets

S = "v3d_amd.sgm."
TRAINING_ONLY_KEYS = ("scheduler_config", "loss_fn_config", "optimizer_config")


def v3d_512_config(num_frames: int = 18, num_steps: int = 30, min_scale: float = 3.5, max_scale: float = 3.5,
                   sigma_max: float = 700.0, model_channels: int = 320, vae_ch: int = 128, ckpt_path=None) -> dict:
    """{'model': {...}} equivalent of V3D_512.yaml (defaults: yaml:134-146; script overrides: V3D_512.py:84-105)."""
    emb = S + "modules.encoders.modules."
    conditioner = {"target": S + "modules.GeneralConditioner", "params": {"emb_models": [
        {"is_trainable": False, "ucg_rate": 0.2, "input_key": "cond_frames_without_noise", "target": emb + "IdentityEncoder"},
        {"input_key": "fps_id", "is_trainable": True, "target": emb + "ConcatTimestepEmbedderND", "params": {"outdim": 256}},
        {"input_key": "motion_bucket_id", "is_trainable": True, "target": emb + "ConcatTimestepEmbedderND", "params": {"outdim": 256}},
        {"input_key": "cond_frames", "is_trainable": False, "ucg_rate": 0.2, "target": emb + "IdentityEncoder"},
        {"input_key": "cond_aug", "is_trainable": True, "target": emb + "ConcatTimestepEmbedderND", "params": {"outdim": 256}},
    ]}}
    dec = synth.decoder_config(vae_ch)
    enc = {k: v for k, v in dec.items() if k != "video_kernel_size"}
    first_stage = {"target": S + "models.autoencoder.AutoencodingEngine", "params": {
        "loss_config": {"target": "torch.nn.Identity"},
        "regularizer_config": {"target": S + "modules.autoencoding.regularizers.DiagonalGaussianRegularizer"},
        "encoder_config": {"target": S + "modules.diffusionmodules.model.Encoder", "params": enc},
        "decoder_config": {"target": S + "modules.autoencoding.temporal_ae.VideoDecoder", "params": dec}}}
    d = S + "modules.diffusionmodules."
    params = {
        "scale_factor": 0.18215, "disable_first_stage_autocast": True, "input_key": "latents", "log_keys": [],
        "denoiser_config": {"target": d + "denoiser.Denoiser", "params": {"scaling_config": {"target": d + "denoiser_scaling.VScalingWithEDMcNoise"}}},
        "network_config": {"target": d + "video_model.VideoUNet", "params": synth.unet_config(model_channels)},
        "conditioner_config": conditioner,
        "first_stage_config": first_stage,
        "sampler_config": {"target": d + "sampling.EulerEDMSampler", "params": {
            "num_steps": num_steps,
            "discretization_config": {"target": d + "discretizer.EDMDiscretization", "params": {"sigma_max": sigma_max}},
            "guider_config": {"target": d + "guiders.LinearPredictionGuider",
                              "params": {"max_scale": max_scale, "min_scale": min_scale, "num_frames": num_frames}}}},
    }
    if ckpt_path is not None:
        params["ckpt_path"] = ckpt_path
    return {"model": {"target": S + "models.video_diffusion.DiffusionEngine", "params": params}}


def load_reference_yaml(path: str) -> dict:
    """Read a reference-format YAML (PyYAML; omegaconf is optional in this image) and point its inference targets here."""
    import yaml
    with open(path) as f:
        cfg = yaml.safe_load(f)
    cfg = remap_targets(cfg)
    params = cfg.get("model", {}).get("params", {})
    for k in TRAINING_ONLY_KEYS:           # training-only plugins are kept verbatim but never instantiated
        params.pop(k, None)
    return cfg
