"""bench.py — V3D_512 dense-multi-view generation throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

One "step" = one pass of the hot path over one batch: a full 18-frame sample — 25 EulerEDM steps (each one
denoiser evaluation of the cfg-doubled 36-image batch through VideoUNet) followed by the 18-frame VideoDecoder decode
(BASELINE.json configs[1]: random-init SVD-XT weights, 1x18x4x64x64 latent, 25 steps, cfg on, bf16, synthetic inputs
already resident in HBM).  metric = multi-view frames / second; N ranks each generate their own sample (independent
inputs, no data-path collective -> weak scaling), value = N * K * 18 / max-over-ranks time.

Also reported on the same JSON line:
  roofline     — for the dominant kernel family (the tap-GEMM `gemm_kernel<...>`: conv3x3 / temporal conv / linear):
                 algorithmic FLOPs (2*M*N*K*taps of every launch of one sample) / summed launch durations measured with
                 HIP events on the launch stream in one extra instrumented sample, against the dense bf16 MFMA peak.
  cpu_baseline — the fp32 CPU oracle (oracle/sgm_oracle.py, kind "port") timed on the host cores on a bounded sample of
                 the same workload and extrapolated (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak, MI355X (MI355X_MICROARCH.md: ~2.5 PF dense)
PEAK_HBM_GBPS = 8000.0
F_UNET_TFLOP = 45.677          # algorithmic FLOPs of one denoiser evaluation (reference graph, cfg batch 36)
F_VAE_TFLOP_PER_FRAME = 3.043

T_FRAMES, STEPS, CFG, LAT = 18, 25, 4.5, 64
P = "v3d_amd.sgm.modules.diffusionmodules."


def build_models(device):
    from v3d_amd import synth
    from v3d_amd.sgm.modules.autoencoding.temporal_ae import VideoDecoder
    from v3d_amd.sgm.modules.diffusionmodules.denoiser import Denoiser
    from v3d_amd.sgm.modules.diffusionmodules.sampling import EulerEDMSampler
    from v3d_amd.sgm.modules.diffusionmodules.video_model import VideoUNet
    from v3d_amd.sgm.modules.diffusionmodules.wrappers import OpenAIWrapper

    with torch.device(device):
        unet = VideoUNet(**synth.unet_config(320)).eval()
        dec = VideoDecoder(**synth.decoder_config(128)).eval()
    synth.init_module_fast(unet, seed=1)
    synth.init_module_fast(dec, seed=2)
    sampler = EulerEDMSampler(
        discretization_config={"target": P + "discretizer.EDMDiscretization", "params": {"sigma_max": 700.0}},
        num_steps=STEPS,
        guider_config={"target": P + "guiders.LinearPredictionGuider", "params": {"max_scale": CFG, "min_scale": CFG, "num_frames": T_FRAMES}},
        device=device)
    denoiser = Denoiser({"target": P + "denoiser_scaling.VScalingWithEDMcNoise"})
    return unet, OpenAIWrapper(unet), dec, sampler, denoiser


def make_step(wrapped, dec, sampler, denoiser, noise, c, uc, device, graph=False):
    """One sample = 25 guided network evaluations + the 18-frame decode.  With `graph` the network evaluation and the
    decode are captured once (first call) into HIP graphs and replayed (v3d_amd/engine/graph.py)."""
    from v3d_amd.engine.graph import graphed
    extra = {"image_only_indicator": torch.zeros(2, T_FRAMES, device=device), "num_video_frames": T_FRAMES}

    def den(inp, sigma, cc):
        return denoiser(wrapped, inp, sigma, cc, **extra)

    def decode(z):
        return dec(z, timesteps=T_FRAMES)

    den_g = graphed(den, enabled=bool(graph))
    dec_g = graphed(decode, enabled=bool(graph))

    def step():
        z = sampler(den_g, noise.clone(), cond=c, uc=uc)
        # DiffusionEngine.decode_first_stage: z / scale_factor, all 18 frames in one chunk (decoding_t = 18)
        return dec_g(z * (1.0 / 0.18215))

    return step


def measure_gemm_roofline(step):
    """One extra instrumented sample: HIP events around every v3d_gemm / v3d_ff_fused launch on the launch stream."""
    from v3d_amd.ops import get_ops
    ops = get_ops()
    orig = ops.gemm
    rec = []

    def timed(g):
        taps = {0: 1, 1: 9, 2: 3}[g.mode]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig(g)
        e1.record()
        # algorithmic bytes of the launch: activation rows once + packed weights once + output (+ residuals) once
        nout = g.N // 2 if g.geglu else g.N
        by = (g.A.shape[0] * g.K * 2 + taps * g.N * g.K * 2) * g.batch + g.M * nout * g.out.element_size() * g.batch
        by += sum(g.M * nout * 2 for r in (g.res1, g.res2) if r is not None)
        rec.append((e0, e1, 2.0 * g.M * g.N * g.K * taps * g.batch, by))

    orig_ff = ops.ff_fused

    def timed_ff(x, w1p, b1, w2p, b2, out, **kw):
        # the fused feed-forward is both GEMMs of the block in one launch: 2 M C (2 hidden) + 2 M hidden C flops; algorithmic bytes =
        # x, both weight matrices, the output and the residuals once (the hidden tensor never exists in memory)
        M, C, hidden = x.shape[0], x.shape[1], w2p.shape[-1]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig_ff(x, w1p, b1, w2p, b2, out, **kw)
        e1.record()
        by = 2 * M * C * 2 + 3 * C * hidden * 2 + sum(M * C * 2 for k in ("res1", "res2") if kw.get(k) is not None)
        rec.append((e0, e1, 6.0 * M * C * hidden, by))
        return r

    ops.gemm = timed
    ops.ff_fused = timed_ff
    try:
        step()
        torch.cuda.synchronize()
    finally:
        ops.gemm = orig
        ops.ff_fused = orig_ff
    tot_ms = sum(a.elapsed_time(b) for a, b, _, _ in rec)
    flops = sum(f for _, _, f, _ in rec)
    alg_bytes = sum(b for _, _, _, b in rec)
    n = len(rec)
    achieved = flops / (tot_ms * 1e-3) / 1e12
    # HBM-side bytes per launch from the committed rocprofv3 PMC passes of the same workload (FETCH_SIZE / WRITE_SIZE in
    # separate runs, calibrated on a copy of known size as MI355X_MICROARCH.md prescribes; tools/pmc_eval.py + pmc_traffic.py)
    traffic = None
    try:
        pmc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01m_pmc_traffic.json")))
        fams = [v for k, v in pmc["families"].items() if k.startswith("gemm_")]
        traffic = round(sum(v["read_GB_per_eval"] + v["write_GB_per_eval"] for v in fams) * 1e9 / sum(v["launches_per_eval"] for v in fams))
    except Exception:
        pass
    return {"bound": "mfma", "kernel": "v3d_gemm family: gemm_kernel_v3<192x320 | 256x256> + gemm_kernel_v2<128x128 ...> (conv3x3 / convt3 / linear / GEGLU) + ff_fused_kernel<320> (both GEMMs of the 64x64 feed-forwards)",
            "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
            "traffic": traffic, "traffic_unit": "HBM-side bytes per launch, U-Net launches (rocprofv3 PMC, profiles/r01m_pmc_traffic.txt)",
            "algorithmic_bytes_per_launch": round(alg_bytes / n), "launches_per_sample": n, "avg_launch_us": round(tot_ms * 1e3 / n, 2),
            "algorithmic_tflop_per_sample": round(flops / 1e12, 2), "gemm_ms_per_sample": round(tot_ms, 2),
            "measured_on": "one extra instrumented sample after the timed region (HIP events per launch)"}


def cpu_baseline(unet, dec, budget_s=40.0):
    """fp32 CPU oracle on a bounded sample: one U-Net evaluation on 4 of the 36 images (cfg 2 x 2 frames, 64x64 latents,
    full width) + decode of 2 frames; extrapolated linearly to 25 x 36-image evaluations + 18 decoded frames."""
    from oracle import sgm_oracle as O
    from v3d_amd import synth
    torch.set_grad_enabled(False)
    cores = torch.get_num_threads()
    Tb = 2
    sd = {k: v.detach().float().cpu() for k, v in unet.state_dict().items()}
    g = torch.Generator().manual_seed(0)
    n = 2 * Tb
    x8 = torch.randn(n, 8, LAT, LAT, generator=g)
    ts = torch.randn(n, generator=g)
    ctx = torch.randn(n, 1, 1024, generator=g)
    y = torch.randn(n, 768, generator=g)
    t0 = time.time()
    ref = O.unet_forward(sd, synth.unet_config(320), x8, ts, ctx, y, Tb, torch.zeros(2, Tb))
    t_unet = time.time() - t0
    del sd
    # full-width parity spot check on the same inputs: HIP engine (bf16) vs the fp32 oracle
    dev = next(unet.parameters()).device
    got = unet(x8.to(dev), ts.to(dev), context=ctx.to(dev), y=y.to(dev), num_video_frames=Tb, image_only_indicator=torch.zeros(2, Tb, device=dev))
    cos_unet = torch.nn.functional.cosine_similarity(got.float().cpu().flatten(), ref.flatten(), dim=0).item()
    rel_unet = ((got.float().cpu() - ref).abs().max() / ref.abs().max()).item()
    dsd = {k: v.detach().float().cpu() for k, v in dec.state_dict().items()}
    z = torch.randn(Tb, 4, LAT, LAT, generator=g)
    t0 = time.time()
    dref = O.decoder_forward(dsd, synth.decoder_config(128), z, Tb)
    t_vae = time.time() - t0
    dgot = dec(z.to(dev), timesteps=Tb)
    cos_vae = torch.nn.functional.cosine_similarity(dgot.float().cpu().flatten(), dref.flatten(), dim=0).item()
    t_sample = STEPS * t_unet * (2 * T_FRAMES / n) + t_vae * (T_FRAMES / Tb)
    return {"value": round(T_FRAMES / t_sample, 6), "unit": "frames/s", "cores": cores, "kind": "port",
            "parity_full_width": {"unet_eval_cosine": round(cos_unet, 6), "unet_eval_max_rel_err": round(rel_unet, 5),
                                  "vae_decode_cosine": round(cos_vae, 6), "note": "HIP bf16 engine vs fp32 CPU oracle on the timed sample's inputs"},
            "sample": f"fp32 oracle: 1 U-Net eval on {n}/36 images ({t_unet:.1f} s) + decode of {Tb}/18 frames ({t_vae:.1f} s), "
                      f"extrapolated to 25 evals x 36 images + 18 frames = {t_sample:.0f} s/sample"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="replay captured HIP graphs of the network evaluation / decode instead of launching from Python "
                         "(measured 9.59 vs 9.62 frames/s: ROCm 7.2 graph replay does not close the launch gaps, so it is off by default)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if args.gpus > 1 and world == 1:
        raise SystemExit("--gpus N > 1 must be launched with python -m torch.distributed.run --nproc-per-node N ...")
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(device))
    torch.set_grad_enabled(False)

    from v3d_amd import synth
    from v3d_amd.ops import get_ops
    assert get_ops().name == "hip"
    unet, wrapped, dec, sampler, denoiser = build_models(device)
    # every rank generates its own sample (different seed per rank): independent objects, no data-path collective
    noise, c, uc = synth.synthetic_conditioning(T_FRAMES, LAT, LAT, seed=23 + rank, device=device)
    step = make_step(wrapped, dec, sampler, denoiser, noise, c, uc, device, graph=args.graph)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    assert out.shape == (T_FRAMES, 3, LAT * 8, LAT * 8) and torch.isfinite(out).all()
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    result = None
    if rank == 0:
        frames = world * args.steps * T_FRAMES
        sample_tflop = STEPS * F_UNET_TFLOP + T_FRAMES * F_VAE_TFLOP_PER_FRAME
        result = {
            "metric": "multi-view frames/sec, V3D_512 18-frame 25-step EDM", "value": round(frames / dt, 4), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: V3D_512 random-init SVD-XT weights, 1x18x4x64x64 latent, 25 EulerEDM steps, "
                                   "cfg 4.5 (LinearPredictionGuider), 18-frame VideoDecoder decode to 512x512, one sample per GPU",
                       "frames": T_FRAMES, "edm_steps": STEPS, "latent": [T_FRAMES, 4, LAT, LAT], "parallelism": f"replica x{world}"},
            "achieved_tflops_reference_graph": round(world * args.steps * sample_tflop / dt, 1),
            "frac_of_bf16_peak_reference_graph": round(args.steps * sample_tflop / dt / PEAK_BF16_TFLOPS, 4),
        }
    if rank == 0 and not args.no_roofline:
        # per-launch HIP events need the launches to come from Python: the instrumented sample runs un-captured
        result["roofline"] = measure_gemm_roofline(make_step(wrapped, dec, sampler, denoiser, noise, c, uc, device, graph=False))
        result["config"]["hip_graph"] = bool(args.graph)
    if world > 1:
        dist.barrier()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            result["cpu_baseline"] = cpu_baseline(unet, dec)
        except Exception as e:  # the baseline is informational; never lose the GPU number to a host-side failure
            result["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port", "sample": f"failed: {e}"}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
