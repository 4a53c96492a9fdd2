"""BASELINE.json configs[4] ("scene": T = 24 frames, 576 x 1024 -> 72 x 128 latents) on the full-width network: wall time of one guided
denoiser evaluation (48 images) and of the 24-frame decode, with the per-op-family table of bench.py's live HIP-event instrumentation
(each family with its own bound).  `--fp8` switches the spatial self-attention to the fp8 (OCP e4m3, MX-scaled MFMA) kernel and reports
both.  Usage (GPU box):  python tools/scene_bench.py [--fp8] [--evals 3]      -> gpurun_out/scene_bench.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

torch.set_grad_enabled(False)
import bench  # noqa: E402
from v3d_amd import synth  # noqa: E402
from v3d_amd.ops import get_ops  # noqa: E402

T, H, W = 24, 72, 128
dev = "cuda"


def table(fam):
    rows = []
    for f, (ms, fl, by, n) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
        mfma = fl > 0 and (fl / max(by, 1) > 312 or f.startswith("attn_spatial") or f.startswith("attn_vae"))
        ach = fl / (ms * 1e-3) / 1e12 if mfma else by / (ms * 1e-3) / 1e9
        rows.append({"kernel": f, "bound": "mfma" if mfma else "hbm", "ms": round(ms, 3), "launches": n, "achieved": round(ach, 1),
                     "unit": "TFLOP/s" if mfma else "GB/s", "frac": round(ach / (bench.PEAK_BF16_TFLOPS if mfma else bench.PEAK_HBM_GBPS), 4)})
    return rows


def main():
    n_eval = int(sys.argv[sys.argv.index("--evals") + 1]) if "--evals" in sys.argv else 3
    unet, wrapped, dec, sampler, denoiser = bench.build_models(dev, frames=T)
    noise, c, uc = synth.synthetic_conditioning(T, H, W, seed=23, device=dev)
    x = torch.cat([noise, noise])
    sig = torch.full((2 * T,), 10.0, device=dev)
    cond = {k: torch.cat([uc[k], c[k]]) for k in c}
    extra = {"image_only_indicator": torch.zeros(2, T, device=dev), "num_video_frames": T}
    ops = get_ops()
    out = {"shape": {"T": T, "latent": [H, W], "images": 2 * T, "tokens_per_level": [H * W, H * W // 4, H * W // 16, H * W // 64]}}
    modes = [("bf16", "0")] + ([("fp8_attention", "1")] if "--fp8" in sys.argv else [])
    ref = None
    for name, flag in modes:
        os.environ["V3D_ATTN_FP8"] = flag
        denoiser(wrapped, x, sig, cond, **extra)                      # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_eval):
            y = denoiser(wrapped, x, sig, cond, **extra)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n_eval * 1e3
        with bench._Timed(ops) as tm:
            denoiser(wrapped, x, sig, cond, **extra)
            torch.cuda.synchronize()
            fam = tm.families()
        entry = {"ms_per_eval": round(ms, 2), "per_kernel": table(fam)}
        if ref is None:
            ref = y.float().clone()
        else:
            a, b = y.float().flatten().double(), ref.flatten().double()
            entry["vs_bf16"] = {"cosine": round(torch.nn.functional.cosine_similarity(a, b, dim=0).item(), 6),
                                "max_rel_err": round(((a - b).abs().max() / b.abs().max()).item(), 5)}
        out[name] = entry
        print(name, json.dumps(entry)[:1500], flush=True)
    os.environ["V3D_ATTN_FP8"] = "0"
    z = torch.randn(T, 4, H, W, device=dev)
    dec(z, timesteps=T)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fr = dec(z, timesteps=T)
    torch.cuda.synchronize()
    out["decode_24_frames_576x1024_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    with bench._Timed(ops) as tm:
        dec(z, timesteps=T)
        torch.cuda.synchronize()
        out["decode_per_kernel"] = table(tm.families())
    assert fr.shape == (T, 3, 8 * H, 8 * W) and torch.isfinite(fr).all()
    out["peak_memory_GB"] = round(torch.cuda.max_memory_allocated() / 1e9, 2)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "scene_bench.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if not isinstance(v, (list, dict)) or k == "shape"}))


if __name__ == "__main__":
    main()
