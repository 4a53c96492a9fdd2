"""TEST INFRASTRUCTURE ONLY - pins oracle/clip_oracle.py's ViT tower to an INDEPENDENT implementation of the same architecture.

open_clip (the package the reference's FrozenOpenCLIPImageEmbedder wraps, sgm/modules/encoders/modules.py:594-752,1054-1072) is absent from this
image, but `transformers` (5.x) is installed and its CLIPVisionModelWithProjection is the same ViT (conv patch embedding without bias, class token +
learned positions, pre-LayerNorm, N x {x + MHA(LN x); x + fc2(GELU(fc1(LN x)))}, post-LayerNorm on the class token, bias-free projection) - written
by other people from the same paper / checkpoints (the HF hub serves laion/CLIP-ViT-H-14-laion2B-s32B-b79K in both formats).  This script
  1. draws a seeded open_clip-named state dict (clip_oracle.seeded_visual_state_dict - plain torch.Generator draws, reproducible anywhere),
  2. remaps it to the transformers names (q | k | v un-concatenated, proj transposed), loads it STRICTLY into CLIPVisionModelWithProjection,
  3. runs the tower on clip_oracle.preprocess(seeded image) and stores `image_embeds` in tests/golden/clip_tower.pt.
The tests hold clip_oracle.VisionTransformer (CPU) and the HIP tower (GPU) to these vectors.  Only the kornia resize in front of the tower remains
restated (kornia is not installed): the fixture's input is the already-preprocessed tensor.

  python oracle/gen_golden_clip.py        (here, CPU; ~1 min: the full-depth ViT-H/14 has 632 M parameters)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import clip_oracle  # noqa: E402

CASES = {
    # name: (vision_cfg of clip_oracle.VisionTransformer, weight seed, image seed, image size)
    "reduced": (dict(image_size=56, patch_size=14, width=256, layers=4, heads=8, mlp_ratio=4.0, embed_dim=128), 11, 12, (96, 80)),
    "vit_h_14": (dict(image_size=224, patch_size=14, width=1280, layers=32, heads=16, mlp_ratio=4.0, embed_dim=1024), 21, 22, (512, 512)),
}


def to_transformers_names(sd, layers):
    """open_clip visual.* names -> transformers CLIPVisionModelWithProjection names."""
    out = {"vision_model.embeddings.class_embedding": sd["class_embedding"],
           "vision_model.embeddings.patch_embedding.weight": sd["conv1.weight"],
           "vision_model.embeddings.position_embedding.weight": sd["positional_embedding"],
           "vision_model.pre_layrnorm.weight": sd["ln_pre.weight"], "vision_model.pre_layrnorm.bias": sd["ln_pre.bias"],
           "vision_model.post_layernorm.weight": sd["ln_post.weight"], "vision_model.post_layernorm.bias": sd["ln_post.bias"],
           "visual_projection.weight": sd["proj"].t().contiguous()}
    for i in range(layers):
        s, d = f"transformer.resblocks.{i}.", f"vision_model.encoder.layers.{i}."
        w, b = sd[s + "attn.in_proj_weight"], sd[s + "attn.in_proj_bias"]
        C = w.shape[1]
        for j, n in enumerate(("q_proj", "k_proj", "v_proj")):
            out[d + f"self_attn.{n}.weight"] = w[j * C:(j + 1) * C].contiguous()
            out[d + f"self_attn.{n}.bias"] = b[j * C:(j + 1) * C].contiguous()
        for a, z in (("attn.out_proj", "self_attn.out_proj"), ("ln_1", "layer_norm1"), ("ln_2", "layer_norm2"), ("mlp.c_fc", "mlp.fc1"), ("mlp.c_proj", "mlp.fc2")):
            out[d + z + ".weight"] = sd[s + a + ".weight"]
            out[d + z + ".bias"] = sd[s + a + ".bias"]
    return out


@torch.no_grad()
def main():
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    torch.set_num_threads(8)
    fix = {}
    for name, (vc, wseed, iseed, size) in CASES.items():
        sd = clip_oracle.seeded_visual_state_dict(vc, wseed)
        hf_cfg = CLIPVisionConfig(hidden_size=vc["width"], intermediate_size=int(vc["width"] * vc["mlp_ratio"]), num_hidden_layers=vc["layers"],
                                  num_attention_heads=vc["heads"], patch_size=vc["patch_size"], image_size=vc["image_size"], projection_dim=vc["embed_dim"],
                                  hidden_act="gelu", layer_norm_eps=1e-5, attention_dropout=0.0)
        hf = CLIPVisionModelWithProjection(hf_cfg).float().eval()
        missing, unexpected = hf.load_state_dict(to_transformers_names(sd, vc["layers"]), strict=False)
        missing = [k for k in missing if "position_ids" not in k]              # (a buffer, not a weight)
        assert not missing and not unexpected, (missing, unexpected)
        img = clip_oracle.seeded_image(iseed, size)
        px = clip_oracle.preprocess(img, vc["image_size"], antialias=True)
        emb = hf(pixel_values=px).image_embeds
        mine = clip_oracle.image_embedding(sd, vc, img)[:, 0]
        d = (mine - emb).abs().max().item()
        print(f"{name}: image_embeds {tuple(emb.shape)}  |emb| max {emb.abs().max().item():.3f}  restatement vs transformers max abs diff {d:.2e}", flush=True)
        fix[name] = dict(vision_cfg=vc, weight_seed=wseed, image_seed=iseed, image_size=size, image_embeds=emb.clone(), pixel_checksum=float(px.double().sum()))
    fix["generator"] = "oracle/gen_golden_clip.py (transformers %s CLIPVisionModelWithProjection, fp32, CPU)" % __import__("transformers").__version__
    out = os.path.join(ROOT, "tests", "golden", "clip_tower.pt")
    torch.save(fix, out)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
