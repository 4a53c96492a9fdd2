"""DiffusionEngine (video) — owner of model / denoiser / sampler / conditioner / first stage
(reference: sgm/models/video_diffusion.py:34-238,363-378).  Inference surface of the reference's Lightning module as a
plain nn.Module: same constructor params (training-only ones accepted and ignored), same attributes the entry script
touches (`.model .denoiser .sampler .conditioner .first_stage_model .scale_factor .en_and_decode_n_samples_a_time`).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Union

import torch
import torch.nn as nn

from ..modules import UNCONDITIONAL_CONFIG
from ..modules.autoencoding.temporal_ae import VideoDecoder
from ..modules.diffusionmodules.wrappers import OPENAIUNETWRAPPER
from ..util import default, disabled_train, get_obj_from_str, instantiate_from_config


class DiffusionEngine(nn.Module):
    def __init__(self, network_config, denoiser_config, first_stage_config, conditioner_config: Union[None, Dict] = None,
                 sampler_config: Union[None, Dict] = None, optimizer_config: Union[None, Dict] = None,
                 scheduler_config: Union[None, Dict] = None, loss_fn_config: Union[None, Dict] = None,
                 network_wrapper: Union[None, str] = None, ckpt_path: Union[None, str] = None, use_ema: bool = False,
                 ema_decay_rate: float = 0.9999, scale_factor: float = 1.0, disable_first_stage_autocast=False,
                 input_key: str = "frames", log_keys: Union[List, None] = None, no_cond_log: bool = False,
                 compile_model: bool = False, en_and_decode_n_samples_a_time: Optional[int] = None,
                 load_last_embedder: bool = False, from_scratch: bool = False):
        super().__init__()
        self.log_keys = log_keys
        self.input_key = input_key
        model = instantiate_from_config(network_config)
        self.model = get_obj_from_str(default(network_wrapper, OPENAIUNETWRAPPER))(model, compile_model=compile_model)
        self.denoiser = instantiate_from_config(denoiser_config)
        self.sampler = instantiate_from_config(sampler_config) if sampler_config is not None else None
        self.conditioner = instantiate_from_config(default(conditioner_config, UNCONDITIONAL_CONFIG))
        self.scheduler_config = scheduler_config          # training-only, kept for config round-trips
        self.loss_fn_config = loss_fn_config              # training-only: the loss is not part of the inference path
        self._init_first_stage(first_stage_config)
        if use_ema:
            raise NotImplementedError("EMA weights are a training feature")
        self.use_ema = False
        self.scale_factor = scale_factor
        self.disable_first_stage_autocast = disable_first_stage_autocast
        self.no_cond_log = no_cond_log
        if load_last_embedder:
            # the reference re-creates the last conditioner embedder's projection from the checkpoint when keys are missing
            # (video_diffusion.py:165-168, _load_last_embedder); silently skipping that would load different weights
            raise NotImplementedError("load_last_embedder=True (video_diffusion.py:165-168) is not supported by this build")
        self.load_last_embedder = False
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, from_scratch)
        self.en_and_decode_n_samples_a_time = en_and_decode_n_samples_a_time

    def init_from_ckpt(self, path: str, from_scratch: bool = False) -> None:
        """.ckpt -> torch.load(...)["state_dict"], .safetensors -> load_file; shape-mismatched keys dropped; strict=False
        (video_diffusion.py:123-168)."""
        if path.endswith("ckpt"):
            sd = torch.load(path, map_location="cpu")["state_dict"]
        elif path.endswith("safetensors"):
            from safetensors.torch import load_file
            sd = load_file(path)
        else:
            raise NotImplementedError
        deleted = []
        for k, v in self.state_dict().items():
            if k in sd and v.shape != sd[k].shape:
                del sd[k]
                deleted.append(k)
        if from_scratch:
            sd = {k: v for k, v in sd.items() if "first_stage_model" in k}
        missing, unexpected = self.load_state_dict(sd, strict=False)
        print(f"Restored from {path} with {len(missing)} missing and {len(unexpected)} unexpected keys")
        if deleted:
            print(f"Deleted Keys: {deleted}")

    def _init_first_stage(self, config):
        model = instantiate_from_config(config).eval()
        model.train = disabled_train
        for p in model.parameters():
            p.requires_grad = False
        self.first_stage_model = model

    def get_input(self, batch):
        return batch[self.input_key]

    @torch.no_grad()
    def decode_first_stage(self, z):
        """z / scale_factor, decoded `en_and_decode_n_samples_a_time` frames at a time with timesteps=len(chunk)
        (video_diffusion.py:182-210).  Chunked decode differs from full decode by construction (3-D GroupNorm statistics and
        the temporal conv see only the chunk), exactly as in the reference."""
        z = z * (1.0 / self.scale_factor)
        is_video_input = z.dim() == 5
        bs = z.shape[0]
        if is_video_input:
            z = z.reshape((-1,) + tuple(z.shape[2:]))
        n_samples = default(self.en_and_decode_n_samples_a_time, z.shape[0])
        outs = []
        for i in range(math.ceil(z.shape[0] / n_samples)):
            chunk = z[i * n_samples:(i + 1) * n_samples]
            kwargs = {"timesteps": len(chunk)} if isinstance(self.first_stage_model.decoder, VideoDecoder) else {}
            outs.append(self.first_stage_model.decode(chunk, **kwargs))
        out = torch.cat(outs, dim=0) if len(outs) > 1 else outs[0]
        if is_video_input:
            out = out.reshape((bs, -1) + tuple(out.shape[1:]))
        return out

    @torch.no_grad()
    def encode_first_stage(self, x):
        if self.input_key == "latents":
            return x * self.scale_factor
        if x.dim() == 5:
            x = x.reshape((-1,) + tuple(x.shape[2:]))
        n_samples = default(self.en_and_decode_n_samples_a_time, x.shape[0])
        outs = [self.first_stage_model.encode(x[i * n_samples:(i + 1) * n_samples])
                for i in range(math.ceil(x.shape[0] / n_samples))]
        return self.scale_factor * torch.cat(outs, dim=0)

    @torch.no_grad()
    def sample(self, cond: Dict, uc: Union[Dict, None] = None, batch_size: int = 16, shape: Union[None, tuple, list] = None,
               **kwargs):
        """Draw latents with the configured sampler (video_diffusion.py:363-378)."""
        randn = torch.randn(batch_size, *shape).to(next(self.model.parameters()).device)

        def denoiser(input, sigma, c):
            return self.denoiser(self.model, input, sigma, c, **kwargs)

        return self.sampler(denoiser, randn, cond, uc=uc)
