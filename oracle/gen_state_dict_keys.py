"""TEST INFRASTRUCTURE - dump the state-dict key / shape / order lists of the REFERENCE modules (imported unmodified from /root/reference
through oracle/ref_import.py) into tests/golden/reference_state_dict_keys.json: the fixture tests/test_plugin_api.py holds the product's
parameter owners against.  Build container only (the GPU box has no /root/reference).

    python -m oracle.gen_state_dict_keys            # rewrites the fixture; prints what changed

Entries: unet_mc64 / decoder_ch32 / encoder_ch32 (the reduced-width parity models) and unet_mc320 / decoder_ch128 / encoder_ch128 (the V3D_512
checkpoint's own widths: what ckpts/V3D_512.ckpt and svd_xt.safetensors hold under model.diffusion_model.* / first_stage_model.*)."""
from __future__ import annotations

import json
import os

import torch

from oracle import ref_import
from v3d_amd import synth

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_state_dict_keys.json")


def keys_of(module) -> dict:
    return {k: list(v.shape) for k, v in module.state_dict().items()}     # insertion order = the module's registration order


def main():
    ref = ref_import.load()
    out = {}
    with torch.device("meta"):
        for mc in (64, 320):
            out[f"unet_mc{mc}"] = keys_of(ref["video_model"].VideoUNet(**synth.unet_config(mc, attn_type="softmax")))
        for ch in (32, 128):
            out[f"decoder_ch{ch}"] = keys_of(ref["temporal_ae"].VideoDecoder(**synth.decoder_config(ch)))
            out[f"encoder_ch{ch}"] = keys_of(ref["model"].Encoder(**synth.encoder_config(ch)))
    old = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for k, v in out.items():
        state = "new" if k not in old else ("unchanged" if (old[k] == v and list(old[k]) == list(v)) else "CHANGED")
        print(f"{k:16s} {len(v):5d} tensors, {sum(int(torch.tensor(s).prod()) if s else 1 for s in v.values()) / 1e6:9.2f} M parameters  [{state}]")
    json.dump(out, open(OUT, "w"))
    print("wrote", OUT)


if __name__ == "__main__":
    main()
