"""TEST INFRASTRUCTURE ONLY - fp32 CPU restatement of the CLIP image front-end of the V3D conditioning path.

Reference call sites: sgm/modules/encoders/modules.py:594-752 (FrozenOpenCLIPImageEmbedder: `preprocess` = kornia.geometry.resize
(224, bicubic, align_corners=True, antialias) -> (x+1)/2 -> kornia.enhance.normalize(mean, std); then `self.model.visual(img)`),
modules.py:1054-1072 (FrozenOpenCLIPImagePredictionEmbedder), scripts/pub/V3D_512.py:146-153,238.

PARITY: the ViT TOWER is pinned (round 6) - tests/golden/clip_tower.pt holds `image_embeds` of transformers' CLIPVisionModelWithProjection (an
independent implementation of the same architecture, importable in this image) on seeded weights, reduced depth and full ViT-H/14
(oracle/gen_golden_clip.py; restatement vs transformers: max abs diff ~1e-6).  The kornia resize in front of it remains RESTATED, unpinned.
The arithmetic lives in two third-party packages that are absent from /root/reference and from this image:
  * open_clip (`open-clip-torch`, requirements.txt, un-pinned): `VisionTransformer.forward` of model ViT-H-14 - conv1 (patch 14,
    no bias), class token + positional embedding, ln_pre, 32 x ResidualAttentionBlock {x + nn.MultiheadAttention(ln_1 x);
    x + c_proj(GELU(c_fc(ln_2 x)))}, ln_post on the class token, @ proj (1280 -> 1024); restated below from its published source
    with torch's own nn.MultiheadAttention / LayerNorm / GELU / Conv2d.
  * kornia (`kornia==0.6.9`, requirements.txt): `geometry.transform.resize` - when down-scaling and antialias: gaussian_blur2d with
    sigma = max((factor - 1) / 2, 0.001) per axis, kernel int(max(4 sigma, 3)) made odd, border "reflect"; then
    F.interpolate(mode="bicubic", align_corners=True).
Neither can be imported, and the reference has no test for this path; the resize restatement is anchored on the reference's call sites and
kornia's published source only.
"""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def kornia_resize_bicubic(x: torch.Tensor, size: int, antialias: bool = True) -> torch.Tensor:
    H, W = x.shape[-2:]
    fy, fx = H / size, W / size
    if antialias and max(fy, fx) > 1:
        sig = (max((fy - 1.0) / 2.0, 0.001), max((fx - 1.0) / 2.0, 0.001))
        ks = [int(max(2.0 * 2 * s, 3)) for s in sig]
        ks = [k + 1 if k % 2 == 0 else k for k in ks]

        def gauss(k, s):
            t = torch.arange(k, dtype=x.dtype) - k // 2
            w = torch.exp(-t.pow(2) / (2 * s * s))
            return w / w.sum()
        k2 = gauss(ks[0], sig[0])[:, None] * gauss(ks[1], sig[1])[None, :]
        xp = F.pad(x, (ks[1] // 2, ks[1] // 2, ks[0] // 2, ks[0] // 2), mode="reflect")
        x = F.conv2d(xp, k2[None, None].repeat(x.shape[1], 1, 1, 1), groups=x.shape[1])
    return F.interpolate(x, size=(size, size), mode="bicubic", align_corners=True)


def preprocess(x: torch.Tensor, size: int = 224, antialias: bool = True) -> torch.Tensor:
    x = kornia_resize_bicubic(x.float(), size, antialias)
    x = (x + 1.0) / 2.0
    return (x - torch.tensor(CLIP_MEAN).view(1, 3, 1, 1)) / torch.tensor(CLIP_STD).view(1, 3, 1, 1)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, width, heads, mlp_width):
        super().__init__()
        self.ln_1 = nn.LayerNorm(width)
        self.attn = nn.MultiheadAttention(width, heads)
        self.ln_2 = nn.LayerNorm(width)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(width, mlp_width)), ("gelu", nn.GELU()), ("c_proj", nn.Linear(mlp_width, width))]))

    def forward(self, x):                      # x [tokens, batch, width] (open_clip runs the transformer sequence-first)
        h = self.ln_1(x)
        x = x + self.attn(h, h, h, need_weights=False)[0]
        return x + self.mlp(self.ln_2(x))


class VisionTransformer(nn.Module):
    def __init__(self, image_size=224, patch_size=14, width=1280, layers=32, heads=16, mlp_ratio=4.0, embed_dim=1024):
        super().__init__()
        g = image_size // patch_size
        self.image_size = image_size
        self.conv1 = nn.Conv2d(3, width, patch_size, patch_size, bias=False)
        self.class_embedding = nn.Parameter(torch.zeros(width))
        self.positional_embedding = nn.Parameter(torch.zeros(g * g + 1, width))
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = nn.Module()
        self.transformer.resblocks = nn.ModuleList([ResidualAttentionBlock(width, heads, int(width * mlp_ratio)) for _ in range(layers)])
        self.ln_post = nn.LayerNorm(width)
        self.proj = nn.Parameter(torch.zeros(width, embed_dim))

    def forward(self, x):
        x = self.conv1(x)
        x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
        x = torch.cat([self.class_embedding.to(x.dtype) + torch.zeros(x.shape[0], 1, x.shape[-1], dtype=x.dtype), x], dim=1)
        x = x + self.positional_embedding.to(x.dtype)
        x = self.ln_pre(x)
        x = x.permute(1, 0, 2)
        for blk in self.transformer.resblocks:
            x = blk(x)
        x = x.permute(1, 0, 2)
        pooled = self.ln_post(x[:, 0])
        return pooled @ self.proj


def seeded_visual_state_dict(vision_cfg, seed):
    """A reproducible open_clip-named state dict of the visual tower (plain CPU torch.Generator draws in key order): what oracle/gen_golden_clip.py
    fed the independent implementation, and what the tests rebuild to compare against tests/golden/clip_tower.pt."""
    g = torch.Generator().manual_seed(seed)
    vit = VisionTransformer(**vision_cfg)
    sd = {}
    for k, v in vit.state_dict().items():
        r = torch.randn(v.shape, generator=g, dtype=torch.float32)
        if k.endswith(".weight") and v.dim() == 1:          # LayerNorm scale
            sd[k] = 1.0 + 0.1 * r
        elif v.dim() == 1 and "class_embedding" not in k:   # biases
            sd[k] = 0.02 * r
        elif k in ("class_embedding", "positional_embedding"):
            sd[k] = 0.05 * r
        else:                                               # matrices / the patch convolution / proj: fan-in scaled
            fan_in = v[0].numel() if k != "proj" else v.shape[0]
            sd[k] = r / fan_in ** 0.5
    return sd


def seeded_image(seed, size):
    g = torch.Generator().manual_seed(seed)
    return torch.rand((1, 3, *size), generator=g) * 2 - 1


@torch.no_grad()
def image_embedding(visual_state_dict, vision_cfg, img, antialias=True, n_cond_frames=1, n_copies=1):
    """FrozenOpenCLIPImagePredictionEmbedder.forward: img [B, 3, H, W] in [-1, 1] -> [B / n_cond_frames * n_copies, n_cond_frames, embed]."""
    vit = VisionTransformer(**vision_cfg).float().eval()
    vit.load_state_dict({k: v.detach().float().cpu() for k, v in visual_state_dict.items()})
    z = vit(preprocess(img.detach().float().cpu(), vit.image_size, antialias))
    z = z.reshape(-1, n_cond_frames, z.shape[-1])
    return z.repeat_interleave(n_copies, dim=0)
