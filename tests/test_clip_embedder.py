"""CLIP image embedder: plugin classes, state-dict contract, engine vs oracle.
The ViT tower of the oracle is PINNED to transformers' CLIPVisionModelWithProjection (tests/golden/clip_tower.pt, oracle/gen_golden_clip.py);
the kornia resize in front of it is a restatement (kornia is absent - see oracle/clip_oracle.py)."""
import os

import pytest
import torch

from conftest import rel_cos
from oracle import clip_oracle
from oracle.ops_emul import EmulOps
from v3d_amd.ops import use_backend
from v3d_amd.sgm.modules.encoders.modules import FrozenOpenCLIPImagePredictionEmbedder
from v3d_amd.sgm.util import instantiate_from_config

TINY_VIT = dict(image_size=56, patch_size=14, width=64, layers=2, heads=4, mlp_ratio=4.0, embed_dim=32)
CFG = {"target": "v3d_amd.sgm.modules.encoders.modules.FrozenOpenCLIPImagePredictionEmbedder",
       "params": {"n_cond_frames": 1, "n_copies": 1,
                  "open_clip_embedding_config": {"target": "v3d_amd.sgm.modules.encoders.modules.FrozenOpenCLIPImageEmbedder",
                                                 "params": {"freeze": True}}}}


def build(vision_cfg=None, seed=0):
    torch.manual_seed(seed)
    cfg = {"target": CFG["target"], "params": dict(CFG["params"])}
    oc = dict(cfg["params"]["open_clip_embedding_config"])
    oc["params"] = dict(oc["params"], vision_cfg=vision_cfg) if vision_cfg else dict(oc["params"])
    cfg["params"]["open_clip_embedding_config"] = oc
    emb = instantiate_from_config(cfg).eval()
    for name, prm in emb.named_parameters():           # away from the LayerNorm / bias defaults
        if prm.dim() == 1 and "class_embedding" not in name:
            prm.data = (prm.data + 0.1 * torch.randn_like(prm)).detach()
    return emb


def visual_sd(emb):
    pre = "open_clip.model.visual."
    return {k[len(pre):]: v for k, v in emb.state_dict().items() if k.startswith(pre)}


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "clip_tower.pt")


@pytest.mark.parametrize("case", ["reduced", "vit_h_14"])
def test_oracle_tower_is_pinned_to_independent_implementation(case):
    """clip_oracle.VisionTransformer on the seeded weights / image of the fixture == image_embeds of transformers' CLIP vision tower."""
    fx = torch.load(GOLDEN)[case]
    sd = clip_oracle.seeded_visual_state_dict(fx["vision_cfg"], fx["weight_seed"])
    img = clip_oracle.seeded_image(fx["image_seed"], fx["image_size"])
    px = clip_oracle.preprocess(img, fx["vision_cfg"]["image_size"], antialias=True)
    assert abs(float(px.double().sum()) - fx["pixel_checksum"]) <= 1e-3 * max(1.0, abs(fx["pixel_checksum"]))     # same tower input as the generator's
    got = clip_oracle.image_embedding(sd, fx["vision_cfg"], img)[:, 0]
    torch.testing.assert_close(got, fx["image_embeds"], rtol=1e-4, atol=1e-4)


def load_seeded(emb, fx):
    sd = clip_oracle.seeded_visual_state_dict(fx["vision_cfg"], fx["weight_seed"])
    emb.load_state_dict({"open_clip.model.visual." + k: v for k, v in sd.items()})
    return emb


def test_engine_matches_independent_implementation_emulated():
    """the engine's tower (kernel contracts executed in fp32 on the CPU) against the transformers fixture, reduced depth"""
    fx = torch.load(GOLDEN)["reduced"]
    emb = load_seeded(build(fx["vision_cfg"]), fx)
    img = clip_oracle.seeded_image(fx["image_seed"], fx["image_size"])
    with use_backend(EmulOps("cpu", exact=True)):
        got = emb(img)[:, 0]
    torch.testing.assert_close(got, fx["image_embeds"], rtol=5e-4, atol=5e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("case,tol", [("reduced", 3e-2), ("vit_h_14", 4e-2)])
def test_hip_tower_matches_independent_implementation(case, tol):
    """HIP kernels (bf16) against image_embeds of transformers' CLIPVisionModelWithProjection: reduced depth and the full ViT-H/14 on 512 x 512."""
    from v3d_amd.hip import HipOps
    fx = torch.load(GOLDEN)[case]
    emb = load_seeded(build(fx["vision_cfg"]), fx).to("cuda")
    img = clip_oracle.seeded_image(fx["image_seed"], fx["image_size"])
    with use_backend(HipOps()):
        got = emb(img.cuda())[:, 0].float().cpu()
    rel, cos = rel_cos(got, fx["image_embeds"])
    from conftest import record_parity
    record_parity(f"clip_tower_{case}_vs_transformers", {"max_rel_err": round(rel, 5), "cosine": round(cos, 6), "layers": fx["vision_cfg"]["layers"], "width": fx["vision_cfg"]["width"]})
    assert rel <= tol and cos >= 0.999, (case, rel, cos)


def test_state_dict_contract_of_vit_h_14():
    """Key names / shapes of the checkpoint's conditioner.embedders.0.* entries (open_clip ViT-H-14 visual tower); text-side
    leftovers of the reference's CLIP object are ignored on load (scripts/pub/V3D_512.py:148-152 loads strictly)."""
    with torch.device("meta"):
        emb = instantiate_from_config(CFG)
    sd = emb.state_dict()
    pre = "open_clip.model.visual."
    want = {"conv1.weight": (1280, 3, 14, 14), "class_embedding": (1280,), "positional_embedding": (257, 1280), "proj": (1280, 1024),
            "ln_pre.weight": (1280,), "ln_post.bias": (1280,), "transformer.resblocks.31.attn.in_proj_weight": (3840, 1280),
            "transformer.resblocks.0.attn.in_proj_bias": (3840,), "transformer.resblocks.5.attn.out_proj.weight": (1280, 1280),
            "transformer.resblocks.7.mlp.c_fc.weight": (5120, 1280), "transformer.resblocks.7.mlp.c_proj.bias": (1280,),
            "transformer.resblocks.30.ln_2.weight": (1280,)}
    for k, shp in want.items():
        assert tuple(sd[pre + k].shape) == shp, k
    assert len(sd) == 8 + 32 * 12 and all(k.startswith(pre) for k in sd)      # buffers mean / std are non-persistent
    tiny = build(TINY_VIT)
    extra = dict(tiny.state_dict())
    extra.update({"open_clip.model.logit_scale": torch.zeros(()), "open_clip.model.token_embedding.weight": torch.zeros(4, 4),
                  "open_clip.model.ln_final.weight": torch.zeros(4), "open_clip.model.attn_mask": torch.zeros(2, 2)})
    tiny.load_state_dict(extra)                                                     # strict


@pytest.mark.parametrize("size,antialias", [((64, 64), True), ((128, 96), True), ((40, 40), True), ((64, 64), False)])
def test_engine_matches_oracle_emulated(size, antialias):
    emb = build(TINY_VIT, seed=1)
    emb.open_clip.antialias = antialias
    img = torch.rand(2, 3, *size) * 2 - 1
    want = clip_oracle.image_embedding(visual_sd(emb), TINY_VIT, img, antialias=antialias)
    with use_backend(EmulOps("cpu", exact=True)):
        got = emb(img)
    assert got.shape == want.shape == (2, 1, 32)
    torch.testing.assert_close(got, want, rtol=2e-4, atol=2e-4)


def test_prediction_embedder_copies():
    emb = build(TINY_VIT, seed=2)
    emb.n_cond_frames, emb.n_copies = 2, 3
    img = torch.rand(4, 3, 56, 56) * 2 - 1
    with use_backend(EmulOps("cpu", exact=True)):
        got = emb(img)
    want = clip_oracle.image_embedding(visual_sd(emb), TINY_VIT, img, n_cond_frames=2, n_copies=3)
    assert got.shape == (6, 2, 32)
    torch.testing.assert_close(got, want, rtol=2e-4, atol=2e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("vcfg,size,tol", [(TINY_VIT, (128, 96), 3e-2), (None, (512, 512), 4e-2)], ids=["tiny", "vit_h_14_512px"])
def test_hip_matches_oracle(vcfg, size, tol):
    """HIP kernels (bf16) vs the fp32 oracle: the tiny tower and the full ViT-H-14 on a 512x512 image (the V3D_512 input size)."""
    from v3d_amd.hip import HipOps
    emb = build(vcfg, seed=3)
    img = torch.rand(1, 3, *size) * 2 - 1
    cfgv = vcfg or dict(image_size=224, patch_size=14, width=1280, layers=32, heads=16, mlp_ratio=4.0, embed_dim=1024)
    want = clip_oracle.image_embedding(visual_sd(emb), cfgv, img)
    emb = emb.to("cuda")
    with use_backend(HipOps()):
        got = emb(img.cuda()).float().cpu()
    rel, cos = rel_cos(got, want)
    assert rel <= tol and cos >= 0.999, (rel, cos)


@pytest.mark.gpu
def test_hip_preprocess_matches_torch_restatement():
    """v3d_clip_preprocess vs the torch restatement of kornia's antialiased bicubic resize + normalise + unfold (bf16 output)."""
    from v3d_amd.hip import HipOps
    hip, emu = HipOps(), EmulOps("cpu", exact=False)
    for (H, W), aa in (((512, 512), True), ((300, 417), True), ((100, 100), True), ((512, 384), False)):
        img = torch.rand(2, 3, H, W) * 2 - 1
        got = hip.clip_preprocess(img.cuda(), 224, 14, aa, clip_oracle.CLIP_MEAN, clip_oracle.CLIP_STD, 608).float().cpu()
        want = emu.clip_preprocess(img, 224, 14, aa, clip_oracle.CLIP_MEAN, clip_oracle.CLIP_STD, 608).float()
        assert (got[:, 588:] == 0).all()
        assert (got - want).abs().max() <= 0.02, ((H, W), aa, (got - want).abs().max())      # bf16 ulp at |x| <= 2.7
