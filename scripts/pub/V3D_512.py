#!/usr/bin/env python
"""Single image -> 18-frame 512x512 orbit video on MI355X (entry point compatible with the reference's
scripts/pub/V3D_512.py: same `sample_one` keyword arguments, same order of operations).

    python scripts/pub/V3D_512.py --input_path assets/img.png --checkpoint_path ckpts/V3D_512.ckpt ...
    python scripts/pub/V3D_512.py --synthetic                 # random-init weights + synthetic conditioning (no checkpoints)

The hot path (sampler loop over VideoUNet + VideoDecoder decode) runs on the hand-written gfx950 kernels, and so does the
conditioning front-end of SURVEY.md §8f rank 1: the OpenCLIP ViT-H/14 image embedding (`clip_model(image)`, V3D_512.py:146-153,
238) and the VAE encode (`ae_model.encode(image)`, :239) of the input view.  The reference's image clean-up before that (rembg
matting, kiui recentering - third-party CPU code) is not part of this build: pass an already prepared RGB image; `--synthetic`
fabricates weights (and, without an image, the conditioning tensors) with the right shapes.
"""
from __future__ import annotations

import argparse
import math
import os
import sys
from typing import Any, Optional

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from v3d_amd import configs, synth  # noqa: E402
from v3d_amd.sgm.util import instantiate_from_config  # noqa: E402


def load_model(config, device: str, num_frames: int, num_steps: int, ckpt_path: Optional[str] = None,
               min_cfg: Optional[float] = None, max_cfg: Optional[float] = None, sigma_max: Optional[float] = None):
    """Build the DiffusionEngine from a config dict / reference YAML path, applying the same overrides the reference
    applies in code (sampler num_steps, guider num_frames and scales, sigma_max, ckpt_path)."""
    cfg = configs.load_reference_yaml(config) if isinstance(config, str) else config
    sp = cfg["model"]["params"]["sampler_config"]["params"]
    sp["num_steps"] = num_steps
    gp = sp["guider_config"]["params"]
    gp["num_frames"] = num_frames
    if max_cfg is not None:
        gp["max_scale"] = max_cfg
    if min_cfg is not None:
        gp["min_scale"] = min_cfg
    if sigma_max is not None:
        sp["discretization_config"]["params"]["sigma_max"] = sigma_max
    cfg["model"]["params"]["from_scratch"] = False
    if ckpt_path is not None:
        cfg["model"]["params"]["ckpt_path"] = str(ckpt_path)
    sp["device"] = device
    with torch.device(device):
        model = instantiate_from_config(cfg["model"])
    return model.to(device).eval(), None


def get_batch(value_dict: dict, T: int, device: str):
    """fps_id / motion_bucket_id / cond_aug repeated over the T frames; the two image conditionings once per sample."""
    batch = {
        "fps_id": torch.tensor([value_dict["fps_id"]], dtype=torch.float32, device=device).repeat(T),
        "motion_bucket_id": torch.tensor([value_dict["motion_bucket_id"]], dtype=torch.float32, device=device).repeat(T),
        "cond_aug": torch.tensor([value_dict["cond_aug"]], dtype=torch.float32, device=device).repeat(T),
        "cond_frames": value_dict["cond_frames"],
        "cond_frames_without_noise": value_dict["cond_frames_without_noise"],
    }
    batch_uc = {k: v.clone() for k, v in batch.items()}
    batch["num_video_frames"] = T
    return batch, batch_uc


CLIP_IMAGE_CONFIG = {   # reference: configs/embedder/clip_image.yaml with sgm. -> v3d_amd.sgm.
    "target": "v3d_amd.sgm.modules.encoders.modules.FrozenOpenCLIPImagePredictionEmbedder",
    "params": {"n_cond_frames": 1, "n_copies": 1,
               "open_clip_embedding_config": {"target": "v3d_amd.sgm.modules.encoders.modules.FrozenOpenCLIPImageEmbedder",
                                              "params": {"freeze": True}}}}


def load_clip_model(device: str, ckpt_path: Optional[str] = None, synthetic: bool = False, clip_config=None):
    """The reference's `clip_model` (V3D_512.py:146-153): FrozenOpenCLIPImagePredictionEmbedder with the
    `conditioner.embedders.0.*` tensors of svd_xt.safetensors."""
    clip_model = instantiate_from_config(clip_config or CLIP_IMAGE_CONFIG).eval()
    if ckpt_path is not None:
        if ckpt_path.endswith(".safetensors"):
            from safetensors.torch import load_file
            sd = load_file(ckpt_path)
        else:
            sd = torch.load(ckpt_path, map_location="cpu")
            sd = sd.get("state_dict", sd)
        clip_sd = {k.replace("conditioner.embedders.0.", ""): v for k, v in sd.items() if "conditioner.embedders.0" in k}
        clip_model.load_state_dict(clip_sd)
    elif not synthetic:
        raise SystemExit("no --clip_checkpoint_path (svd_xt.safetensors) given for the OpenCLIP image embedder: pass --synthetic "
                         "to run it on random-init weights")
    return clip_model.to(device)


@torch.no_grad()
def sample_one(input_path: str = "assets/test_image.png", checkpoint_path: Optional[str] = None, num_frames: Optional[int] = None,
               num_steps: Optional[int] = None, fps_id: int = 1, motion_bucket_id: int = 300, cond_aug: float = 0.02, seed: int = 23,
               decoding_t: int = 24, device: str = "cuda", output_folder: Optional[str] = None, noise: torch.Tensor = None,
               save: bool = False, cached_model: Any = None, border_ratio: float = 0.3, min_guidance_scale: float = 3.5,
               max_guidance_scale: float = 3.5, sigma_max: float = None, ignore_alpha: bool = False, *, config=None,
               cond_frames: torch.Tensor = None, cond_frames_without_noise: torch.Tensor = None, image: torch.Tensor = None,
               synthetic: bool = False, clip_checkpoint_path: Optional[str] = None, clip_config=None,
               height: int = 512, width: int = 512, model_channels: int = 320, vae_ch: int = 128,
               ae_checkpoint_path: Optional[str] = None):
    """Returns (frames uint8 [T, H, W, 3] on the host, model).  Keyword arguments up to `ignore_alpha` are the reference's."""
    num_frames = 18 if num_frames is None else num_frames       # the reference reads it from the guider config (18)
    num_steps = 25 if num_steps is None else num_steps
    decoding_t = min(decoding_t, num_frames)
    cfg = config if config is not None else configs.v3d_512_config(num_frames=num_frames, num_steps=num_steps, model_channels=model_channels, vae_ch=vae_ch)
    if cached_model is None:
        model, _ = load_model(cfg, device, num_frames, num_steps, ckpt_path=checkpoint_path, min_cfg=min_guidance_scale,
                              max_cfg=max_guidance_scale, sigma_max=sigma_max)
        if checkpoint_path is None:
            if not synthetic:
                raise SystemExit("no --checkpoint_path given: pass --synthetic to run on random-init weights")
            synth.init_module_fast(model.model.diffusion_model, seed=1)
            synth.init_module_fast(model.first_stage_model.decoder, seed=2)
    else:
        model = cached_model
    torch.manual_seed(seed)
    F = 8
    h, w = height // F, width // F
    if cond_frames is None and image is not None:
        # native VAE encode of the conditioning view: image [1,3,H,W] in [-1,1].  The reference encodes with a SEPARATE autoencoder whose
        # weights are svd_xt.safetensors' first_stage_model.* (`ae_model`, V3D_512.py:155-163,239), not with V3D_512.ckpt's first stage:
        # `ae_checkpoint_path` (default: the CLIP checkpoint, which is that same svd_xt file) reproduces that; without it the model's own
        # first stage is used and a warning says so.
        ae_model = getattr(model, "_v3d_ae_model", None)
        ae_ckpt = ae_checkpoint_path or clip_checkpoint_path
        if ae_model is None and ae_ckpt is not None and ae_ckpt.endswith("safetensors"):
            from safetensors.torch import load_file
            fsd = {k[len("first_stage_model."):]: v for k, v in load_file(ae_ckpt).items() if k.startswith("first_stage_model.")}
            if fsd:
                fcfg = cfg["model"]["params"]["first_stage_config"] if "model" in cfg else cfg["params"]["first_stage_config"]
                ae_model = instantiate_from_config(fcfg).eval()
                # strict, like the reference's `ae_model.load_state_dict(...)` (V3D_512.py:162): a wrong or truncated svd_xt file must not
                # silently encode the conditioning view with partly random weights
                ae_model.load_state_dict(fsd, strict=True)
                print(f"conditioning autoencoder restored from {ae_ckpt} ({len(fsd)} tensors)")
                ae_model = ae_model.to(device)
                # cached OUTSIDE the module tree: a plain attribute assignment would register it as a submodule and put `_v3d_ae_model.*`
                # keys into model.state_dict() / parameters() / .to()
                object.__setattr__(model, "_v3d_ae_model", ae_model)
        if ae_model is None:
            ae_model = model.first_stage_model
            if checkpoint_path is not None:
                print("warning: the conditioning view is encoded with the model's own first stage; the reference uses svd_xt.safetensors' "
                      "first_stage_model.* - pass --ae_checkpoint_path (or --clip_checkpoint_path) to match it")
            elif synthetic and cached_model is None:
                synth.init_module_fast(model.first_stage_model.encoder, seed=3)
        cond_frames = ae_model.encode(image.to(device).float())
    if cond_frames_without_noise is None and image is not None:
        # native OpenCLIP image embedding (reference: configs/embedder/clip_image.yaml -> `clip_model(image)`, V3D_512.py:146-153,238);
        # weights = the checkpoint's conditioner.embedders.0.* (svd_xt.safetensors), random-initialised under --synthetic
        clip_model = getattr(model, "_v3d_clip_model", None)
        if clip_model is None:
            clip_model = load_clip_model(device, clip_checkpoint_path, synthetic, clip_config)
            object.__setattr__(model, "_v3d_clip_model", clip_model)      # (outside the module tree, like the conditioning autoencoder)
        cond_frames_without_noise = clip_model(image.to(device).float())
    if cond_frames is None or cond_frames_without_noise is None:
        if not synthetic:
            raise SystemExit(
                f"no conditioning for {input_path}: pass `image` [1,3,H,W] in [-1,1] (or cond_frames [1,4,H/8,W/8] and "
                "cond_frames_without_noise [1,1,1024] tensors), or --synthetic")
        g = torch.Generator().manual_seed(seed)
        if cond_frames_without_noise is None:
            cond_frames_without_noise = torch.randn(1, 1, 1024, generator=g).to(device)
        if cond_frames is None:
            cond_frames = torch.randn(1, 4, h, w, generator=g).to(device)
    cond_frames = cond_frames.to(device) + cond_aug * torch.randn_like(cond_frames.to(device))
    value_dict = dict(motion_bucket_id=motion_bucket_id, fps_id=fps_id, cond_aug=cond_aug, cond_frames=cond_frames,
                      cond_frames_without_noise=cond_frames_without_noise.to(device))
    batch, batch_uc = get_batch(value_dict, num_frames, device)
    c, uc = model.conditioner.get_unconditional_conditioning(batch, batch_uc=batch_uc,
                                                             force_uc_zero_embeddings=["cond_frames", "cond_frames_without_noise"])
    for k in ("crossattn", "concat"):      # one embedding / latent repeated over the frames
        uc[k] = uc[k].repeat_interleave(num_frames, dim=0)
        c[k] = c[k].repeat_interleave(num_frames, dim=0)
    randn = torch.randn((num_frames, 4, h, w), device=device) if noise is None else noise.to(device)
    extra = {"image_only_indicator": torch.zeros(2, num_frames, device=device), "num_video_frames": num_frames}

    def denoiser(inp, sigma, cc):
        return model.denoiser(model.model, inp, sigma, cc, **extra)

    samples_z = model.sampler(denoiser, randn, cond=c, uc=uc)
    model.en_and_decode_n_samples_a_time = decoding_t
    samples_x = model.decode_first_stage(samples_z)
    # output stage in one kernel (clamp, layout, uint8); `model.last_frames_u8` keeps the device tensor [T, H, W, 3] for consumers
    # that take frames without a trip through an mp4 file (the reference hands them to recon/train_from_vid.py as a video)
    from v3d_amd.ops import get_ops
    model.last_frames_u8 = get_ops().frames_to_uint8(samples_x.float().contiguous())
    frames = model.last_frames_u8.cpu().numpy()
    if save:
        folder = output_folder or "outputs/V3D_512"
        os.makedirs(folder, exist_ok=True)
        path = os.path.join(folder, f"{len(os.listdir(folder)):06d}")
        try:
            import mediapy
            mediapy.write_video(path + ".mp4", frames, fps=3)
        except ImportError:
            import numpy as np
            np.save(path + ".npy", frames)
            print(f"mediapy not installed: wrote raw frames to {path}.npy")
    return frames, model


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--input_path", default="assets/test_image.png")
    ap.add_argument("--checkpoint_path", default=None)
    ap.add_argument("--config", default=None, help="a reference-format YAML (targets are remapped onto v3d_amd.sgm.*)")
    ap.add_argument("--num_frames", type=int, default=None)
    ap.add_argument("--num_steps", type=int, default=None)
    ap.add_argument("--fps_id", type=int, default=1)
    ap.add_argument("--motion_bucket_id", type=int, default=300)
    ap.add_argument("--cond_aug", type=float, default=0.02)
    ap.add_argument("--seed", type=int, default=23)
    ap.add_argument("--decoding_t", type=int, default=24)
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--output_folder", default=None)
    ap.add_argument("--save", action="store_true")
    ap.add_argument("--min_guidance_scale", type=float, default=3.5)
    ap.add_argument("--max_guidance_scale", type=float, default=3.5)
    ap.add_argument("--sigma_max", type=float, default=None)
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--clip_checkpoint_path", default=None, help="svd_xt.safetensors (conditioner.embedders.0.* = the OpenCLIP image tower)")
    ap.add_argument("--ae_checkpoint_path", default=None, help="svd_xt.safetensors (first_stage_model.* = the autoencoder the reference encodes the "
                                                               "conditioning view with); defaults to --clip_checkpoint_path")
    ap.add_argument("--border_ratio", type=float, default=0.3, help="reference argument of its rembg / kiui recentering step, which this build does "
                                                                    "not run: the input must already be a prepared (matted, centred) RGB view")
    ap.add_argument("--ignore_alpha", action="store_true")
    a = ap.parse_args()
    if not os.path.isfile(a.input_path) and not a.synthetic:
        raise SystemExit(f"input image {a.input_path} not found (pass --synthetic to run on synthetic conditioning)")
    image = None
    if os.path.isfile(a.input_path):
        from PIL import Image as _Image
        if _Image.open(a.input_path).mode in ("RGBA", "LA") and not a.ignore_alpha:
            print("warning: the input has an alpha channel; the reference composites it on white after rembg / recentering (border_ratio "
                  f"{a.border_ratio}) - this build takes the RGB channels as they are")
        # plain load + resize to 512 x 512 -> [-1, 1]; the reference's matting / recentering (rembg, kiui) stays outside this build
        from PIL import Image
        import numpy as np
        im = Image.open(a.input_path).convert("RGB").resize((512, 512))
        image = torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1)[None].float() / 127.5 - 1.0
    frames, _ = sample_one(a.input_path, a.checkpoint_path, a.num_frames, a.num_steps, a.fps_id, a.motion_bucket_id, a.cond_aug, a.seed,
                           a.decoding_t, a.device, a.output_folder, save=a.save, min_guidance_scale=a.min_guidance_scale,
                           max_guidance_scale=a.max_guidance_scale, sigma_max=a.sigma_max, config=a.config, synthetic=a.synthetic, image=image,
                           clip_checkpoint_path=a.clip_checkpoint_path, ae_checkpoint_path=a.ae_checkpoint_path, border_ratio=a.border_ratio,
                           ignore_alpha=a.ignore_alpha)
    print("frames", frames.shape, frames.dtype, "mean", float(frames.mean()))
    # parity status of what just ran: sampler / U-Net / decoder / VAE encoder are pinned to fixtures generated from the
    # reference's own modules; the OpenCLIP ViT tower is pinned to transformers' implementation of the same architecture (tests/golden/clip_tower.pt);
    # kornia (the antialiased resize in front of it) is third-party code absent from the reference tree and this image
    print("parity: sampler/unet/decoder/vae_encoder pinned to reference fixtures; clip tower pinned to transformers' CLIPVisionModelWithProjection; "
          "kornia resize: restated from its published algorithm, unpinned (oracle/clip_oracle.py)")


if __name__ == "__main__":
    main()
