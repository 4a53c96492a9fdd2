"""Conditioner front-end of the V3D_512 config (reference: sgm/modules/encoders/modules.py:42-206 AbstractEmbModel /
GeneralConditioner, 229-234 IdentityEncoder, 937-953 ConcatTimestepEmbedderND).

Runs once per sample before the sampling loop (not part of the timed hot path); the CLIP image embedder
(FrozenOpenCLIPImageEmbedder / FrozenOpenCLIPImagePredictionEmbedder, modules.py:594-752,1054-1072, below) and the VAE encoder
that produce `cond_frames_without_noise` / `cond_frames` are applied by the entry script, exactly as in
scripts/pub/V3D_512.py:238-243, and enter here through IdentityEncoder.
"""
from __future__ import annotations

import math
from contextlib import nullcontext
from typing import Dict, List, Optional, Union

import torch
import torch.nn as nn

from ...util import count_params, disabled_train, instantiate_from_config


class AbstractEmbModel(nn.Module):
    def __init__(self):
        super().__init__()
        self.is_trainable = None
        self.ucg_rate = None
        self.input_key = None


class IdentityEncoder(AbstractEmbModel):
    def encode(self, x):
        return x

    def forward(self, x):
        return x


class Timestep(nn.Module):
    """Sinusoidal embedding [cos | sin] of a scalar (openaimodel.py `Timestep` -> util.timestep_embedding)."""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def forward(self, t):
        half = self.dim // 2
        freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
        args = t[:, None].float() * freqs[None]
        emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
        if self.dim % 2:
            emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
        return emb


class ConcatTimestepEmbedderND(AbstractEmbModel):
    """Embeds each dimension independently and concatenates them (modules.py:937-953)."""

    def __init__(self, outdim):
        super().__init__()
        self.timestep = Timestep(outdim)
        self.outdim = outdim

    def forward(self, x):
        if x.ndim == 1:
            x = x[:, None]
        assert len(x.shape) == 2
        b, dims = x.shape
        emb = self.timestep(x.reshape(b * dims))
        return emb.reshape(b, dims * self.outdim)


class GeneralConditioner(nn.Module):
    OUTPUT_DIM2KEYS = {2: "vector", 3: "crossattn", 4: "concat", 5: "concat"}
    KEY2CATDIM = {"vector": 1, "crossattn": 2, "concat": 1}

    def __init__(self, emb_models: Union[List, tuple]):
        super().__init__()
        embedders = []
        for n, embconfig in enumerate(emb_models):
            embedder = instantiate_from_config(embconfig)
            assert isinstance(embedder, AbstractEmbModel), \
                f"embedder model {embedder.__class__.__name__} has to inherit from AbstractEmbModel"
            embedder.is_trainable = embconfig.get("is_trainable", False)
            embedder.ucg_rate = embconfig.get("ucg_rate", 0.0)
            if not embedder.is_trainable:
                embedder.train = disabled_train
                for param in embedder.parameters():
                    param.requires_grad = False
                embedder.eval()
            if "input_key" in embconfig:
                embedder.input_key = embconfig["input_key"]
            elif "input_keys" in embconfig:
                embedder.input_keys = embconfig["input_keys"]
            else:
                raise KeyError(f"need either 'input_key' or 'input_keys' for embedder {embedder.__class__.__name__}")
            embedder.legacy_ucg_val = embconfig.get("legacy_ucg_value", None)
            if embedder.legacy_ucg_val is not None:
                raise NotImplementedError("legacy_ucg_value is a training-time feature")
            embedders.append(embedder)
        self.embedders = nn.ModuleList(embedders)

    def forward(self, batch: Dict, force_zero_embeddings: Optional[List] = None) -> Dict:
        output = dict()
        force_zero_embeddings = force_zero_embeddings or []
        for embedder in self.embedders:
            ctx = nullcontext if embedder.is_trainable else torch.no_grad
            with ctx():
                if getattr(embedder, "input_key", None) is not None:
                    emb_out = embedder(batch[embedder.input_key])
                else:
                    emb_out = embedder(*[batch[k] for k in embedder.input_keys])
            if not isinstance(emb_out, (list, tuple)):
                emb_out = [emb_out]
            for emb in emb_out:
                out_key = self.OUTPUT_DIM2KEYS[emb.dim()]
                if embedder.ucg_rate > 0.0:
                    keep = torch.bernoulli((1.0 - embedder.ucg_rate) * torch.ones(emb.shape[0], device=emb.device))
                    emb = keep.reshape((-1,) + (1,) * (emb.dim() - 1)) * emb
                if getattr(embedder, "input_key", None) in force_zero_embeddings:
                    emb = torch.zeros_like(emb)
                if out_key in output:
                    output[out_key] = torch.cat((output[out_key], emb), self.KEY2CATDIM[out_key])
                else:
                    output[out_key] = emb
        return output

    def get_unconditional_conditioning(self, batch_c: Dict, batch_uc: Optional[Dict] = None,
                                       force_uc_zero_embeddings: Optional[List[str]] = None,
                                       force_cond_zero_embeddings: Optional[List[str]] = None):
        force_uc_zero_embeddings = force_uc_zero_embeddings or []
        rates = []
        for e in self.embedders:
            rates.append(e.ucg_rate)
            e.ucg_rate = 0.0
        c = self(batch_c, force_cond_zero_embeddings)
        uc = self(batch_c if batch_uc is None else batch_uc, force_uc_zero_embeddings)
        for e, r in zip(self.embedders, rates):
            e.ucg_rate = r
        return c, uc


class FrozenOpenCLIPImageEmbedder(AbstractEmbModel):
    """OpenCLIP ViT image embedding on the HIP kernels (modules.py:594-752).  Same constructor parameters; `arch` selects the
    vision config (ViT-H-14 is the one SVD / V3D checkpoints carry; `vision_cfg` overrides it, e.g. for tests).  `version` names
    pretrained weights in the reference - here weights come from the checkpoint's `conditioner.embedders.0.*` keys (or stay
    random-initialised); nothing is downloaded."""

    ARCHS = {"ViT-H-14": "VIT_H_14"}

    def __init__(self, arch="ViT-H-14", version="laion2b_s32b_b79k", device="cuda", max_length=77, freeze=True, antialias=True,
                 ucg_rate=0.0, unsqueeze_dim=False, repeat_to_max_len=False, num_image_crops=0, output_tokens=False,
                 init_device=None, vision_cfg: Optional[Dict] = None):
        super().__init__()
        from . import open_clip_vit
        if vision_cfg is None:
            if arch not in self.ARCHS:
                raise NotImplementedError(f"FrozenOpenCLIPImageEmbedder: arch {arch!r} (known: {sorted(self.ARCHS)})")
            vision_cfg = getattr(open_clip_vit, self.ARCHS[arch])
        if output_tokens or num_image_crops:
            raise NotImplementedError("FrozenOpenCLIPImageEmbedder: output_tokens / num_image_crops are not used by SVD / V3D")
        self.model = open_clip_vit.CLIPVisualOnly(**vision_cfg)
        self.max_crops = num_image_crops
        self.pad_to_max_len = False
        self.repeat_to_max_len = repeat_to_max_len
        self.device = device
        self.max_length = max_length
        if freeze:
            self.freeze()
        self.antialias = antialias
        self.register_buffer("mean", torch.Tensor([0.48145466, 0.4578275, 0.40821073]), persistent=False)
        self.register_buffer("std", torch.Tensor([0.26862954, 0.26130258, 0.27577711]), persistent=False)
        self.ucg_rate = ucg_rate
        self.unsqueeze_dim = unsqueeze_dim
        self.output_tokens = output_tokens

    def freeze(self):
        self.model = self.model.eval()
        for param in self.parameters():
            param.requires_grad = False

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        """The reference's `self.model` is a whole open_clip CLIP with only `.transformer` deleted, so checkpoints also carry its
        text-side leftovers (token_embedding, positional_embedding, ln_final, text_projection, logit_scale, attn_mask): dropped."""
        sd = {k: v for k, v in state_dict.items() if not (k.startswith("model.") and not k.startswith("model.visual."))}
        return super().load_state_dict(sd, strict=strict, **kw)

    def encode_with_vision_transformer(self, img):
        from ....engine.clip import run_clip_visual
        vis = self.model.visual
        return run_clip_visual(vis.packed(), img, self.antialias, self.mean.tolist(), self.std.tolist())

    @torch.no_grad()
    def forward(self, image, no_dropout=False):
        z = self.encode_with_vision_transformer(image).to(image.dtype)
        if self.ucg_rate > 0.0 and not no_dropout:
            z = torch.bernoulli((1.0 - self.ucg_rate) * torch.ones(z.shape[0], device=z.device))[:, None] * z
        if self.unsqueeze_dim:
            z = z[:, None, :]
        if self.repeat_to_max_len:
            z_ = z[:, None, :] if z.dim() == 2 else z
            return z_.expand(-1, self.max_length, -1).contiguous(), z
        return z

    def encode(self, text):
        return self(text)


class FrozenOpenCLIPImagePredictionEmbedder(AbstractEmbModel):
    """modules.py:1054-1072: (b t) d -> b t d over n_cond_frames, each repeated n_copies times."""

    def __init__(self, open_clip_embedding_config: Dict, n_cond_frames: int, n_copies: int):
        super().__init__()
        self.n_cond_frames = n_cond_frames
        self.n_copies = n_copies
        self.open_clip = instantiate_from_config(open_clip_embedding_config)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        sd = {k: v for k, v in state_dict.items()
              if not (k.startswith("open_clip.model.") and not k.startswith("open_clip.model.visual."))}
        return super().load_state_dict(sd, strict=strict, **kw)

    def forward(self, vid):
        vid = self.open_clip(vid)
        vid = vid.reshape(-1, self.n_cond_frames, vid.shape[-1])
        return vid.repeat_interleave(self.n_copies, dim=0)
