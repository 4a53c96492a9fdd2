"""TEST INFRASTRUCTURE — full-width fixtures from the REFERENCE's own modules: tests/golden/v3d_full.pt.

Run in the build container only (needs /root/reference; ~45 min on 8 cores, ~25 GB of RAM):   python -m oracle.gen_golden_full
The headline configuration (BASELINE.json configs[1]) executed end to end by the reference's unmodified modules
(oracle/ref_import.py), fp32 on the CPU:
    EulerEDMSampler(25 steps, sigma_max 700) x LinearPredictionGuider(4.5) x Denoiser(VScalingWithEDMcNoise) x OpenAIWrapper x
    VideoUNet(model_channels 320, 36 images = cfg 2 x T 18, 64 x 64 latents)                    sampling.py:44-133, guiders.py:61-101,
    denoiser.py:23-39, wrappers.py:24-34, video_model.py:442-493
    -> decode_first_stage (z / 0.18215, decoding_t = 18) -> VideoDecoder(ch 128) -> 18 x 3 x 512 x 512        video_diffusion.py:182-210,
    temporal_ae.py:293-349
Weights: v3d_amd.synth.seeded_state_dict (CPU generator keyed by tensor name: bit-identical on every machine), U-Net seed 1234,
decoder seed 1235; conditioning / noise: synth.synthetic_conditioning(seed 23) — nothing but OUTPUTS is stored:
    call{0,8,14,20}_x     fp32 [18,4,64,64]   the sampler state entering that step (the denoiser input is cat([x, x]); step 0 also regenerable)
    call{0,8,14,20}_out   fp16 [36,4,64,64]   the reference Denoiser's output for the cfg-doubled batch at that step (teacher-forcing targets)
    z                     fp32 [18,4,64,64]   final latent of the 25-step rollout
    frames                fp16 [2,3,512,512]  decoded frames 0 and 9;  frame_stats fp32 [18,2] = (mean, std) of every decoded frame
    meta                  dict: seeds, sigmas, wall times, threads (the `kind: reference` CPU baseline of a REAL full run)
"""
from __future__ import annotations

import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402
from v3d_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "v3d_full.pt")
FULL = dict(model_channels=320, vae_ch=128, T=18, H=64, W=64, cond_seed=23, unet_seed=1234, dec_seed=1235, steps=25, scale=4.5, sigma_max=700.0,
            scale_factor=0.18215, record_calls=(0, 8, 14, 20), frames_kept=(0, 9))


@torch.no_grad()
def main():
    torch.set_grad_enabled(False)
    nthr = int(os.environ.get("V3D_GEN_THREADS", "0"))
    if nthr:
        torch.set_num_threads(nthr)
    m = ref_import.load()
    p = FULL
    T, H, W = p["T"], p["H"], p["W"]
    out = {"params": {k: (list(v) if isinstance(v, tuple) else v) for k, v in p.items()}}
    t_all = time.time()
    net = m["video_model"].VideoUNet(**synth.unet_config(p["model_channels"], attn_type="softmax")).eval()
    net.load_state_dict(synth.seeded_state_dict(net, p["unet_seed"]), strict=True)
    print(f"[gen_full] reference VideoUNet built + seeded in {time.time() - t_all:.0f} s", flush=True)
    noise, c, uc = synth.synthetic_conditioning(T, H, W, seed=p["cond_seed"])
    sampler = m["sampling"].EulerEDMSampler(
        discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.EDMDiscretization", "params": {"sigma_max": p["sigma_max"]}},
        num_steps=p["steps"],
        guider_config={"target": "sgm.modules.diffusionmodules.guiders.LinearPredictionGuider",
                       "params": {"max_scale": p["scale"], "min_scale": p["scale"], "num_frames": T}},
        device="cpu")
    denoiser = m["denoiser"].Denoiser({"target": "sgm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"})
    wrapped = m["wrappers"].OpenAIWrapper(net)
    extra = {"image_only_indicator": torch.zeros(2, T), "num_video_frames": T}
    state = {"i": 0, "times": [], "sigmas": []}

    def den(inp, sigma, cc):
        t0 = time.time()
        o = denoiser(wrapped, inp, sigma, cc, **extra)
        dt = time.time() - t0
        i = state["i"]
        state["times"].append(round(dt, 1))
        state["sigmas"].append(float(sigma[0]))
        if i in p["record_calls"]:
            out[f"call{i}_x"] = inp[:T].clone()
            out[f"call{i}_out"] = o.to(torch.float16).clone()
        print(f"[gen_full] step {i:2d} sigma {float(sigma[0]):10.4f}  {dt:6.1f} s", flush=True)
        state["i"] = i + 1
        return o

    t0 = time.time()
    z = sampler(den, noise.clone(), cond=c, uc=uc)
    t_samp = time.time() - t0
    out["z"] = z.clone()
    del net, wrapped
    dec = m["temporal_ae"].VideoDecoder(**synth.decoder_config(p["vae_ch"])).eval()
    dec.load_state_dict(synth.seeded_state_dict(dec, p["dec_seed"]), strict=True)
    t0 = time.time()
    frames = dec(1.0 / p["scale_factor"] * z, timesteps=T)           # video_diffusion.py:182-210 (one chunk: decoding_t = T)
    t_dec = time.time() - t0
    out["frames"] = frames[list(p["frames_kept"])].to(torch.float16).clone()
    out["frame_stats"] = torch.stack([frames.mean(dim=(1, 2, 3)), frames.std(dim=(1, 2, 3))], dim=1).clone()
    out["meta"] = {"threads": torch.get_num_threads(), "sampler_seconds": round(t_samp, 1), "decode_seconds": round(t_dec, 1),
                   "per_eval_seconds": state["times"], "sigmas": state["sigmas"],
                   "frames_per_s": round(T / (t_samp + t_dec), 6), "torch": str(torch.__version__)}
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    torch.save(out, OUT)
    for k, v in out.items():
        if torch.is_tensor(v):
            print(f"{k:14s} {tuple(v.shape)} {v.dtype} mean|x|={v.float().abs().mean():.4f} max|x|={v.float().abs().max():.4f}")
    print("meta", out["meta"])
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
