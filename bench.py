"""bench.py — V3D_512 dense-multi-view generation throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W [--shard replica|frames]     (N > 1: launched by torch.distributed.run, one rank per GPU)

One "step" = one pass of the hot path over one batch: a full 18-frame sample — 25 EulerEDM steps (each one
denoiser evaluation of the cfg-doubled 36-image batch through VideoUNet) followed by the 18-frame VideoDecoder decode
(BASELINE.json configs[1]: random-init SVD-XT weights, 1x18x4x64x64 latent, 25 steps, cfg on, bf16, synthetic inputs
already resident in HBM).  metric = multi-view frames / second.

Multi-GPU modes (--gpus N > 1):
  --shard replica (default)  every rank generates its own sample (independent inputs, no data-path collective): weak scaling,
                             value = N * K * 18 / max-over-ranks time.  After the timed region the SAME job is also measured in
                             frame-sharded mode and reported as the extra object "frame_shard" (it never changes `value`).
  --shard frames             BASELINE.json configs[2]/[3]: ONE sample, its 18 frames sharded over the N ranks for the whole path
                             (v3d_amd/dist.py: K|V exchange before each temporal attention, +-1 frame halos for the (3,1,1) convs,
                             all-reduced 3-D GroupNorm sums; weights replicated): strong scaling, value = K * 18 / max-over-ranks time.

Also reported on the same JSON line:
  roofline     — for the dominant kernel family (the `v3d_gemm` launches + the fused feed-forward): algorithmic FLOPs / summed launch
                 durations measured LIVE with HIP events on the launch stream in one extra instrumented sample, against the dense bf16
                 MFMA peak; `per_kernel` lists every timed op family of that sample with its own bound (mfma / hbm), achieved rate and
                 fraction of that bound's peak.  `traffic` (HBM-side bytes per launch) comes from the committed rocprofv3 PMC passes
                 (PMC counters cannot be collected inside this process); `traffic_profile` says which profile and whether the kernel
                 sources changed since it was taken.
  cpu_baseline — the fp32 CPU oracle (oracle/sgm_oracle.py, kind "port") timed on the host cores on a bounded sample of
                 the same workload and extrapolated (rank 0, N = 1 only), plus parity of the HIP path against it: one full-width
                 evaluation on the timed inputs and the 25-step width-64 rollout (latent cosine, decoded-frame PSNR).
"""
from __future__ import annotations

import argparse
import glob
import hashlib
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak, MI355X (MI355X_MICROARCH.md: ~2.5 PF dense)
PEAK_FP8_TFLOPS = 5000.0       # dense fp8 MFMA peak (the scene workload's e4m3 attention kernel is priced against this one)
PEAK_HBM_GBPS = 8000.0
RIDGE_FLOP_PER_BYTE = PEAK_BF16_TFLOPS * 1e12 / (PEAK_HBM_GBPS * 1e9)      # 312.5: launches below it are HBM-bound (classified per launch)
F_UNET_TFLOP = 45.677          # algorithmic FLOPs of one denoiser evaluation (reference graph, cfg batch 36)
F_VAE_TFLOP_PER_FRAME = 3.043
PMC_PROFILE = "r06_pmc_traffic.json"     # profiles/<this>: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/profile.sh)

T_FRAMES, STEPS, CFG, LAT = 18, 25, 4.5, 64
P = "v3d_amd.sgm.modules.diffusionmodules."


def csrc_digest() -> str:
    """Content hash of the kernel sources: profiles record it, so a profile taken before the last kernel change is detectable."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "v3d_amd", "csrc", "*"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def build_models(device, width=320, vae_ch=128, steps=STEPS, cfg=CFG, frames=T_FRAMES):
    from v3d_amd import synth
    from v3d_amd.sgm.modules.autoencoding.temporal_ae import VideoDecoder
    from v3d_amd.sgm.modules.diffusionmodules.denoiser import Denoiser
    from v3d_amd.sgm.modules.diffusionmodules.sampling import EulerEDMSampler
    from v3d_amd.sgm.modules.diffusionmodules.video_model import VideoUNet
    from v3d_amd.sgm.modules.diffusionmodules.wrappers import OpenAIWrapper

    with torch.device(device):
        unet = VideoUNet(**synth.unet_config(width)).eval()
        dec = VideoDecoder(**synth.decoder_config(vae_ch)).eval()
    synth.init_module_fast(unet, seed=1)
    synth.init_module_fast(dec, seed=2)
    sampler = EulerEDMSampler(
        discretization_config={"target": P + "discretizer.EDMDiscretization", "params": {"sigma_max": 700.0}},
        num_steps=steps,
        guider_config={"target": P + "guiders.LinearPredictionGuider", "params": {"max_scale": cfg, "min_scale": cfg, "num_frames": frames}},
        device=device)
    denoiser = Denoiser({"target": P + "denoiser_scaling.VScalingWithEDMcNoise"})
    return unet, OpenAIWrapper(unet), dec, sampler, denoiser


def make_step(wrapped, dec, sampler, denoiser, noise, c, uc, device, graph=False, frames=T_FRAMES, inputs=1):
    """One sample = 25 guided network evaluations + the 18-frame decode.  With `graph` the network evaluation and the
    decode are captured once (first call) into HIP graphs and replayed (v3d_amd/engine/graph.py)."""
    from v3d_amd.engine.graph import graphed
    extra = {"image_only_indicator": torch.zeros(2 * inputs, frames, device=device), "num_video_frames": frames}

    def den(inp, sigma, cc):
        return denoiser(wrapped, inp, sigma, cc, **extra)

    def decode(z):
        # DiffusionEngine.decode_first_stage with en_and_decode_n_samples_a_time = decoding_t = frames (V3D_512.py:187): one sample's frames per
        # decoder call (with --inputs 4 the 72-frame activation of the 512 x 512 level would also exceed the 4 GiB a buffer descriptor spans)
        if z.shape[0] == frames:
            return dec(z, timesteps=frames)
        return torch.cat([dec(z[i:i + frames], timesteps=frames) for i in range(0, z.shape[0], frames)], dim=0)

    den_g = graphed(den, enabled=bool(graph))
    dec_g = graphed(decode, enabled=bool(graph))

    def step():
        z = sampler(den_g, noise.clone(), cond=c, uc=uc)
        # DiffusionEngine.decode_first_stage: z / scale_factor, all 18 frames in one chunk (decoding_t = 18)
        return dec_g(z * (1.0 / 0.18215))

    return step


def make_sharded_step(shard, wrapped, dec, sampler, denoiser, noise, c, uc, B=1):
    """One sample with its frames sharded over the ranks for the whole path (v3d_amd/dist.py::sharded_sample): returns the gathered
    [18, 3, 512, 512] frames on every rank."""
    from v3d_amd.dist import sharded_sample

    def decode(z):
        return dec(z * (1.0 / 0.18215), timesteps=shard.T_local)

    def step():
        return sharded_sample(shard, sampler, denoiser, wrapped, decode, noise.clone(), c, uc, B=B)

    return step


# ---- live per-launch timing of one instrumented sample --------------------------------------------------------------------------
class _Timed:
    """Wraps the primitive ops of the backend with HIP events on the launch stream and algorithmic flop / byte counts."""

    def __init__(self, ops):
        self.ops = ops
        self.rec = []            # (family, e0, e1, flops, bytes)
        self.saved = {}

    def _wrap(self, name, meter):
        orig = getattr(self.ops, name)
        self.saved[name] = orig

        def timed(*a, **k):
            fam, flops, by = meter(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(*a, **k)
            e1.record()
            self.rec.append((fam, e0, e1, flops, by))
            return r

        setattr(self.ops, name, timed)

    def __enter__(self):
        def m_gemm(g):
            taps = {0: 1, 1: 9, 2: 3}[g.mode]
            nout = g.N // 2 if g.geglu else g.N
            # algorithmic bytes of the launch: activation rows once + packed weights once + output (+ residuals) once
            by = (g.A.shape[-2] * g.K * 2 + taps * g.N * g.K * 2) * g.batch + g.M * nout * g.out.element_size() * g.batch
            by += sum(g.M * nout * 2 for r in (g.res1, g.res2) if r is not None)
            fam = {0: "gemm_linear_geglu" if g.geglu else ("gemm_batched" if g.batch > 1 else "gemm_linear"), 1: "gemm_conv3x3", 2: "gemm_convt3"}[g.mode]
            fl = 2.0 * g.M * g.N * g.K * taps * g.batch
            if g.gn_in is not None:
                fam += "_gn"          # GroupNorm + SiLU applied in the operand path (conv.hip): same algorithmic flops / bytes, no apply pass
            elif fl / by < RIDGE_FLOP_PER_BYTE:
                fam += "_hbm"         # classified PER LAUNCH: below the ridge the launch is HBM-bound (the K = 320 / 640 linears at 64x64 / 32x32)
            return fam, fl, by

        def m_ff(x, w1p, b1, w2p, b2, out, **kw):
            # both GEMMs of the block in one launch: 2 M C (2 hidden) + 2 M hidden C flops; bytes = x, both weight matrices, the output and
            # the residuals once (the hidden tensor never exists in memory)
            M, C, hidden = x.shape[0], x.shape[1], w2p.shape[-1]
            by = 2 * M * C * 2 + 3 * C * hidden * 2 + sum(M * C * 2 for k in ("res1", "res2") if kw.get(k) is not None)
            return "gemm_ff_fused", 6.0 * M * C * hidden, by

        def m_attn(q, k, vT, out, n_img, S, heads, scale):
            return "attn_spatial", 4.0 * n_img * heads * S * S * 64, 4 * n_img * S * heads * 64 * 2

        def m_tattn(q, k, v, out, heads, scale):
            B, Tq, S, C = q.shape
            Tk = k.shape[1]
            return "attn_temporal", 4.0 * B * S * heads * Tq * Tk * 64, (2 * B * Tq * S * C + 2 * B * Tk * S * C) * 2

        def m_gns(x1, x2, stats, n_img, S, groups, ips):
            C = x1.shape[-1] + (0 if x2 is None else x2.shape[-1])
            return "gn_stats", 0.0, n_img * S * C * 2

        def m_gna(x1, x2, table, out, n_img, S, *a):
            C = x1.shape[-1] + (0 if x2 is None else x2.shape[-1])
            return "gn_apply", 0.0, 2 * n_img * S * C * 2

        def m_gnf(stats, sums, gamma, beta, count, eps, table):
            return "gn_finalize", 0.0, (0 if stats is None else stats.numel() * 4) + (0 if table is None else table.numel() * 4)

        def m_ln(x, gamma, beta, out, eps, add=None, add_rpg=0, add_ld=0, xsum_out=None):
            return "layernorm", 0.0, (3 if xsum_out is not None else 2) * x.numel() * 2

        def m_lnp(x, eps, wp, bias, n_rm, S):
            # LayerNorm + q | k | v projection in one launch (proj.hip): the GEMM's flops; bytes = token rows, weights and outputs once
            M, C, N = x.shape[0], x.shape[1], wp.shape[0]
            return "gemm_ln_proj", 2.0 * M * N * C, M * C * 2 + N * C * 2 + M * N * 2

        self._wrap("gemm", m_gemm)
        if hasattr(self.ops, "ln_proj"):
            self._wrap("ln_proj", m_lnp)
        if hasattr(self.ops, "ff_fused"):
            self._wrap("ff_fused", m_ff)
        if hasattr(self.ops, "ln_ff_fused"):          # the same kernel with the LayerNorm on its resident rows
            self._wrap("ln_ff_fused", lambda x, eps, w1p, b1, w2p, b2, out, **kw: m_ff(x, w1p, b1, w2p, b2, out, **kw))
        self._wrap("attn_spatial", m_attn)
        self._wrap("attn_temporal", m_tattn)
        self._wrap("groupnorm_stats", m_gns)
        if hasattr(self.ops, "groupnorm_stats_table"):      # ABI 5: statistics + table in one launch (same kernel family, same bytes: the tensor once)
            self._wrap("groupnorm_stats_table", lambda x1, x2, stats, tickets, n_img, S, groups, ips, *a: m_gns(x1, x2, stats, n_img, S, groups, ips))
        self._wrap("groupnorm_apply", m_gna)
        self._wrap("groupnorm_finalize", m_gnf)
        self._wrap("layernorm", m_ln)
        # round-3 kernels that bypassed the meters: the one-launch GroupNorm of the small levels and the im2col / tap-sum helpers
        # of the U-Net's first / last convolution (their GEMMs are metered as gemm_linear)
        if hasattr(self.ops, "groupnorm_small"):
            self._wrap("groupnorm_small", lambda x1, x2, gamma, beta, out, n_img, S, **kw: ("gn_small", 0.0, 2 * out.numel() * 2))
        if hasattr(self.ops, "pack_input_im2col3x3"):
            self._wrap("pack_input_im2col3x3", lambda x, scale, cond, Kpad: ("conv_io_helpers", 0.0, x.numel() * 4 + (cond.numel() * 4 if cond is not None else 0) + x.shape[0] * x.shape[2] * x.shape[3] * Kpad * 2))
        if hasattr(self.ops, "tapsum3x3"):
            self._wrap("tapsum3x3", lambda y, bias, n, H, W, C: ("conv_io_helpers", 0.0, y.numel() * 4 + n * H * W * C * 4))
        if hasattr(self.ops, "attn_spatial_fp8"):     # scene-config variant (V3D_ATTN_FP8=1): attention + its two quantisation passes
            self._wrap("attn_spatial_fp8", lambda qk8, sc, v8, vs, out, n_img, S, heads, scale: ("attn_spatial_fp8", 4.0 * n_img * heads * S * S * 64, 3 * n_img * S * heads * 64 + n_img * S * heads * 128))
            self._wrap("quant_fp8_tiles", lambda x, n_img, S: ("quant_fp8", 0.0, x.shape[0] * x.shape[1] * 3))
            self._wrap("quant_fp8_slab", lambda vT, heads: ("quant_fp8", 0.0, vT.numel() * 5))
        if hasattr(self.ops, "attn_vae"):
            self._wrap("attn_vae", lambda q, k, vT, bias, out, n_img, S, C, scale: ("attn_vae_d512", 4.0 * n_img * S * S * C, 4 * n_img * S * C * 2))
        return self

    def __exit__(self, *exc):
        for k in self.saved:
            self.ops.__dict__.pop(k, None)     # the originals are class attributes: dropping the instance override restores them

    def families(self):
        fam = {}
        for f, e0, e1, fl, by in self.rec:
            a = fam.setdefault(f, [0.0, 0.0, 0.0, 0])
            a[0] += e0.elapsed_time(e1)
            a[1] += fl
            a[2] += by
            a[3] += 1
        return fam


def _safe(fn):
    """context values must never cost the run its metric line"""
    try:
        return fn()
    except Exception as e:      # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}


def live_clock_probe():
    """Shader clock this chip holds with every SIMD issuing MFMAs back to back (v3d_debug_clock_probe: s_memtime against the 100 MHz s_memrealtime, ~10 ms of
    v_mfma_f32_16x16x32_bf16 on random operands, two waves per SIMD, no memory traffic).  Round 6 saw 12.0 and 12.85 frames/s from ONE library on boxes of one pool:
    this number says which kind of box a run was on.  None if the library lacks the entry (an older build)."""
    import ctypes
    from v3d_amd.ops import get_ops
    lib = get_ops().lib
    if not hasattr(lib, "v3d_debug_clock_probe"):
        return None
    fn = lib.v3d_debug_clock_probe
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    out = torch.zeros(8, dtype=torch.int64, device="cuda")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    best = None
    for iters in (20000, 60000, 60000):          # a warm-up launch (the clock takes milliseconds to settle under load), then two measured ones
        if fn(iters, ctypes.c_void_p(out.data_ptr()), stream) != 0:
            return None
        torch.cuda.synchronize()
        c0, c1, r0, r1 = [int(v) for v in out[:4].tolist()]
        if r1 > r0:
            best = {"ghz": round((c1 - c0) / ((r1 - r0) * 10.0), 3), "window_ms": round((r1 - r0) * 1e-5, 2)}
    return best


def measure_rooflines(step):
    """One extra instrumented sample: HIP events around every timed op launch on the launch stream."""
    from v3d_amd.ops import get_ops
    with _Timed(get_ops()) as t:
        step()
        torch.cuda.synchronize()
        fam = t.families()
    per = []
    for f, (ms, fl, by, n) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
        mfma = fl / max(by, 1) > RIDGE_FLOP_PER_BYTE and fl > 0 and not f.endswith("_hbm")       # above the ridge -> MFMA-bound
        if f.startswith("attn_spatial") or f.startswith("attn_vae"):
            mfma = True                                                                        # (QK^T / PV re-read K,V from L2, never HBM-bound)
        ach = fl / (ms * 1e-3) / 1e12 if mfma else by / (ms * 1e-3) / 1e9
        per.append({"kernel": f, "bound": "mfma" if mfma else "hbm", "ms_per_sample": round(ms, 2), "launches": n,
                    "achieved": round(ach, 1), "unit": "TFLOP/s" if mfma else "GB/s",
                    "frac": round(ach / (PEAK_BF16_TFLOPS if mfma else PEAK_HBM_GBPS), 4)})
    gem = {k: v for k, v in fam.items() if k.startswith("gemm_")}
    tot_ms = sum(v[0] for v in gem.values())
    flops = sum(v[1] for v in gem.values())
    alg_bytes = sum(v[2] for v in gem.values())
    n = sum(v[3] for v in gem.values())
    achieved = flops / (tot_ms * 1e-3) / 1e12
    traffic, tinfo = None, {"file": "profiles/" + PMC_PROFILE, "found": False}
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_PROFILE)))
        fams = [v for k, v in pmc["families"].items() if k.startswith("gemm_")]
        traffic = round(sum(v["read_GB_per_eval"] + v["write_GB_per_eval"] for v in fams) * 1e9 / sum(v["launches_per_eval"] for v in fams))
        tinfo = {"file": "profiles/" + PMC_PROFILE, "found": True, "csrc_sha256_16": pmc.get("csrc_sha256_16"), "taken": pmc.get("taken"),
                 "kernels_changed_since": pmc.get("csrc_sha256_16") != csrc_digest()}
    except Exception:
        pass
    return {"bound": "mfma", "kernel": "v3d_gemm family: conv_halo_kernel (GroupNorm + SiLU + conv3x3 / conv(3,1,1) in one kernel) + gemm_kernel_v3 / v6 / v2 (conv3x3 / convt3 / linear / GEGLU launches, incl. the GroupNorm-statistics epilogue of the 3x3 convolutions) + ff_fused_kernel<320> (both GEMMs of the 64x64 feed-forwards) + ln_proj_kernel<320> (LayerNorm + q|k|v projection)",
            "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
            "traffic": traffic, "traffic_unit": "HBM-side bytes per launch, U-Net launches (rocprofv3 PMC passes)", "traffic_profile": tinfo,
            "algorithmic_bytes_per_launch": round(alg_bytes / max(n, 1)), "launches_per_sample": n, "avg_launch_us": round(tot_ms * 1e3 / max(n, 1), 2),
            "algorithmic_tflop_per_sample": round(flops / 1e12, 2), "gemm_ms_per_sample": round(tot_ms, 2),
            "timed_ms_per_sample_all_families": round(sum(v[0] for v in fam.values()), 2),
            "measured_on": "one extra instrumented sample after the timed region (HIP events per launch, csrc " + csrc_digest() + ")",
            # context, not the contract's `frac`: the shader clock measured inside a full-occupancy MFMA main loop on this chip (tools/clock_probe.py:
            # s_memtime against the 100 MHz s_memrealtime) is 1.77 GHz with all 256 CUs busy (2.41 GHz with 16) - `peak` above is the 2.4 GHz figure
            "sustained_clock_context": {"ghz_all_cus_busy": 1.773, "ghz_16_cus_busy": 2.407, "bf16_dense_peak_at_that_clock_tflops": 1859.0,
                                        "frac_of_that": round(achieved / 1859.0, 4), "source": "profiles/r06_clock_probe.txt (recorded inside the fused feed-forward)",
                                        # measured on THIS box right now (MFMA-only probe kernel: an upper bound of the clock under the real kernels, comparable box to box)
                                        "live_mfma_probe": _safe(live_clock_probe)},
            "per_kernel": per}


def cpu_baseline(unet, dec):
    """fp32 CPU oracle on a bounded sample: a warm-up evaluation, then TWO timed U-Net evaluations on 6 of the 36 images (one cfg half,
    T = 6 frames, 64x64 latents, full width) + decode of 2 frames; extrapolated linearly to 25 x 36-image evaluations + 18 decoded
    frames.  The HIP engine is checked against the oracle on the same inputs."""
    from oracle import sgm_oracle as O
    from v3d_amd import synth
    torch.set_grad_enabled(False)
    cores = torch.get_num_threads()
    sd = {k: v.detach().float().cpu() for k, v in unet.state_dict().items()}
    cfg = synth.unet_config(320)
    g = torch.Generator().manual_seed(0)

    def inputs(n):
        return (torch.randn(n, 8, LAT, LAT, generator=g), torch.randn(n, generator=g), torch.randn(n, 1, 1024, generator=g), torch.randn(n, 768, generator=g))

    xw, tw, cw, yw = inputs(2)
    O.unet_forward(sd, cfg, xw, tw, cw, yw, 2, torch.zeros(1, 2))                     # warm-up (thread pool, oneDNN primitives, first touch)
    Tb = 6
    x8, ts, ctx, y = inputs(Tb)
    times = []
    for _ in range(2):
        t0 = time.time()
        ref = O.unet_forward(sd, cfg, x8, ts, ctx, y, Tb, torch.zeros(1, Tb))
        times.append(time.time() - t0)
    t_unet = sum(times) / len(times)
    del sd
    # full-width parity spot check on the same inputs: HIP engine (bf16) vs the fp32 oracle
    dev = next(unet.parameters()).device
    got = unet(x8.to(dev), ts.to(dev), context=ctx.to(dev), y=y.to(dev), num_video_frames=Tb, image_only_indicator=torch.zeros(1, Tb, device=dev))
    cos_unet = torch.nn.functional.cosine_similarity(got.float().cpu().flatten(), ref.flatten(), dim=0).item()
    rel_unet = ((got.float().cpu() - ref).abs().max() / ref.abs().max()).item()
    dsd = {k: v.detach().float().cpu() for k, v in dec.state_dict().items()}
    Td = 2
    z = torch.randn(Td, 4, LAT, LAT, generator=g)
    t0 = time.time()
    dref = O.decoder_forward(dsd, synth.decoder_config(128), z, Td)
    t_vae = time.time() - t0
    dgot = dec(z.to(dev), timesteps=Td)
    cos_vae = torch.nn.functional.cosine_similarity(dgot.float().cpu().flatten(), dref.flatten(), dim=0).item()
    t_sample = STEPS * t_unet * (2 * T_FRAMES / Tb) + t_vae * (T_FRAMES / Td)
    reference = None
    try:        # the reference's OWN modules timed where /root/reference exists (tools/cpu_reference_baseline.py, build container; record committed)
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "cpu_reference_baseline.json")) as f:
            reference = json.load(f)
    except Exception:
        pass
    return {"value": round(T_FRAMES / t_sample, 6), "unit": "frames/s", "cores": cores, "kind": "port", "reference": reference,
            "parity_full_width": {"unet_eval_cosine": round(cos_unet, 6), "unet_eval_max_rel_err": round(rel_unet, 5),
                                  "vae_decode_cosine": round(cos_vae, 6), "note": "HIP bf16 engine vs fp32 CPU oracle on the timed sample's inputs"},
            "sample": f"fp32 oracle, {cores} threads: warm-up eval, then 2 timed U-Net evals on {Tb}/36 images at T={Tb} ({times[0]:.1f} s, {times[1]:.1f} s) + decode of {Td}/18 "
                      f"frames ({t_vae:.1f} s), extrapolated to 25 evals x 36 images + 18 frames = {t_sample:.0f} s/sample"}


def parity_rollout(device, lat=32):
    """SURVEY.md 8d end-to-end bar, measured live on a bounded case: width-64 network, T = 18, 25 EulerEDM steps, cfg 4.5, decode - HIP
    path vs the fp32 oracle: latent cosine and decoded-frame PSNR (peak = value range of the oracle's frames).  Live at 32x32 latents
    (-> 256x256 frames; the 25-step oracle rollout at 64x64 takes minutes of CPU time); the 64x64 numbers of the -m gpu test
    tests/test_headline_parity_gpu.py::test_rollout_25_steps_cosine_and_psnr are attached from profiles/ with their provenance."""
    from oracle import sgm_oracle as O
    from v3d_amd import synth
    from v3d_amd.sgm.modules.autoencoding.temporal_ae import VideoDecoder
    from v3d_amd.sgm.modules.diffusionmodules.video_model import VideoUNet
    from v3d_amd.sgm.modules.diffusionmodules.wrappers import OpenAIWrapper
    unet = VideoUNet(**synth.unet_config(64)).eval()
    dec = VideoDecoder(**synth.decoder_config(32)).eval()
    unet.load_state_dict(synth.seeded_state_dict(unet, 1234))
    dec.load_state_dict(synth.seeded_state_dict(dec, 1235))
    usd = {k: v.float() for k, v in unet.state_dict().items()}
    dsd = {k: v.float() for k, v in dec.state_dict().items()}
    unet, dec = unet.to(device), dec.to(device)
    sampler, denoiser = build_models_sampler(device)
    noise, c, uc = synth.synthetic_conditioning(T_FRAMES, lat, lat, seed=23)
    mv = lambda d: {k: v.to(device) for k, v in d.items()}
    z = make_step_latents(OpenAIWrapper(unet), sampler, denoiser, noise.clone().to(device), mv(c), mv(uc), device)
    frames = dec(z * (1.0 / 0.18215), timesteps=T_FRAMES).float().cpu()
    ioi = torch.zeros(2, T_FRAMES)
    ucfg, dcfg = synth.unet_config(64), synth.decoder_config(32)
    t0 = time.time()
    nthr = torch.get_num_threads()
    torch.set_num_threads(min(nthr, 32))          # small-op workload: more threads make it slower
    z_ref = O.sample_edm(lambda x8, cn, ctx, vec: O.unet_forward(usd, ucfg, x8, cn, ctx, vec, T_FRAMES, ioi), noise.clone(), c, uc, STEPS, T_FRAMES, CFG, CFG, 700.0)
    f_ref = O.decode_first_stage(dsd, dcfg, z_ref, 0.18215, T_FRAMES)
    torch.set_num_threads(nthr)
    dt = time.time() - t0
    cos = torch.nn.functional.cosine_similarity(z.double().cpu().flatten(), z_ref.double().flatten(), dim=0).item()
    mse = ((frames.double() - f_ref.double()) ** 2).mean().item()
    peak = (f_ref.max() - f_ref.min()).item()
    out = {"config": f"width 64, T=18, {lat}x{lat} latent, 25 EulerEDM steps, cfg 4.5, decode to {8 * lat}x{8 * lat}", "latent_cosine": round(cos, 6),
           "frames_psnr_db": round(10.0 * math.log10(peak * peak / mse), 2) if mse > 0 else None, "oracle_seconds": round(dt, 1)}
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "r06_parity.json")))
        r = rec["rollout_25_steps_width64"]
        out["recorded_64x64"] = {"latent_cosine": r["latent_cosine"], "frames_psnr_db": r["frames_psnr_db"], "decoder_only_psnr_db": r.get("decoder_only_psnr_db"),
                                 "source": "profiles/r06_parity.json (written by tests/test_headline_parity_gpu.py on the GPU box)",
                                 # round 5: the headline configuration itself against the REFERENCE's own modules (tests/golden/v3d_full.pt)
                                 "headline_width320_vs_reference_modules": rec.get("rollout_25_steps_width320_vs_reference"),
                                 "headline_evals_vs_reference_modules": [rec[k] for k in ("headline_eval_vs_reference_call0", "headline_eval_vs_reference_call8",
                                                                                         "headline_eval_vs_reference_call14", "headline_eval_vs_reference_call20") if k in rec],
                                 "headline_unet_eval": rec.get("headline_unet_eval"),
                                 "headline_midschedule_evals": [rec[k] for k in ("headline_midschedule_eval_step8", "headline_midschedule_eval_step14",
                                                                                "headline_midschedule_eval_step20") if k in rec]}
    except Exception:
        pass
    return out


def shard_sim(args, device, unet, wrapped, dec, denoiser, noise, c, uc):
    """bench.py --shard-sim N: what ONE GPU of an N-GPU frame-sharded run (BASELINE.json configs[2] / [3]) COMPUTES, measured on this one GPU.
    A SimFrameShard (v3d_amd/dist.py) plays the rank with the most frames (rank 0) and the rank with the fewest (rank N - 1) with
    self-fed halos / K|V / statistics: the same kernels, tile counts and buffer sizes as inside the real run, no communication.  Reported per
    rank: ms per guided U-Net evaluation and per decode (from two sharded samples of 2 and 6 EDM steps), the GEMM-family launches whose
    tiles do not fill the CUs and their share of the GEMM time, the bytes and grouped point-to-point calls the real exchanges would carry; and
    against the unsharded sample on the same GPU: the compute-only strong-scaling ceiling.  NO RCCL TIMING EXISTS FOR THIS PATH (one GPU
    per build box): the ceiling assumes free communication."""
    from v3d_amd.dist import SimFrameShard, sharded_sample
    from v3d_amd.ops import get_ops
    from v3d_amd.sgm.modules.diffusionmodules.sampling import EulerEDMSampler
    N = args.shard_sim
    B_IN, E_STEPS = args.inputs, args.edm_steps          # (BASELINE.json configs[3]: --inputs 4 --edm-steps 50)
    ops = get_ops()
    cus = ops.cu_count

    def sampler_of(steps):
        return EulerEDMSampler(discretization_config={"target": P + "discretizer.EDMDiscretization", "params": {"sigma_max": 700.0}}, num_steps=steps,
                               guider_config={"target": P + "guiders.LinearPredictionGuider", "params": {"max_scale": CFG, "min_scale": CFG, "num_frames": T_FRAMES}},
                               device=device)

    def wall(fn, reps=2):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    def run(rank):
        sh = SimFrameShard(T_FRAMES, N, rank) if N > 1 else None
        s2, s6 = sampler_of(2), sampler_of(6)
        extra = {"image_only_indicator": torch.zeros(2 * B_IN, T_FRAMES, device=device), "num_video_frames": T_FRAMES}

        dec_local = (lambda z: dec(z * (1.0 / 0.18215), timesteps=sh.T_local)) if sh is not None else None      # (one object per rank: the graph cache keys on shapes)

        def sample(smp):
            if sh is None:
                z = smp(lambda i, sg, cc: denoiser(wrapped, i, sg, cc, **extra), noise.clone(), cond=c, uc=uc) * (1.0 / 0.18215)
                return torch.cat([dec(z[i:i + T_FRAMES], timesteps=T_FRAMES) for i in range(0, z.shape[0], T_FRAMES)], dim=0)      # (as make_step: one input at a time)
            return sharded_sample(sh, smp, denoiser, wrapped, dec_local, noise.clone(), c, uc, B=B_IN, gather=False, graph=bool(args.graph))

        t2, t6 = wall(lambda: sample(s2)), wall(lambda: sample(s6))
        ev = (t6 - t2) / 4.0
        decode = t2 - 2.0 * ev
        # one instrumented 2-step sample: GEMM-family launches by tile fill
        rec = []
        orig = ops.gemm

        def timed_gemm(g):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(g)
            e1.record()
            rec.append((ops.last_gemm_launch()["fill"], e0, e1))          # the library's own record of what it launched (tiles / CU slots)
            return r
        ops.gemm = timed_gemm
        b0, x0 = (sh.bytes_sent, sh.n_exchanges) if sh else (0, 0)
        try:
            sample(s2)
            torch.cuda.synchronize()
        finally:
            ops.__dict__.pop("gemm", None)
        tot = sum(a.elapsed_time(b) for _, a, b in rec)
        under = [ms for f, ms in ((f, a.elapsed_time(b)) for f, a, b in rec) if f < 0.75]
        half = [ms for f, ms in ((f, a.elapsed_time(b)) for f, a, b in rec) if f < 0.5]
        out = {"frames": sh.T_local if sh else T_FRAMES, "ms_per_unet_eval": round(ev, 2), "ms_per_decode": round(decode, 2),
               "ms_per_sample_compute_only": round(E_STEPS * ev + decode, 1), "edm_steps": E_STEPS, "inputs": B_IN,
               "gemm_launches_per_2_step_sample": len(rec), "launches_filling_under_75pct_of_the_cu_slots": len(under), "their_share_of_gemm_time": round(sum(under) / max(tot, 1e-9), 3),
               "launches_filling_under_50pct": len(half), "their_share_of_gemm_time_50pct": round(sum(half) / max(tot, 1e-9), 3)}
        if sh:
            nb, nx = sh.bytes_sent - b0, sh.n_exchanges - x0
            # the 2-step sample = 2 evaluations + 1 decode: per-evaluation figures from the U-Net part only would need a second counter; report the sample's
            out["exchange_MB_per_2_step_sample_incl_decode"] = round(nb / 1e6, 1)
            out["grouped_p2p_calls_per_2_step_sample_incl_decode"] = nx
        return out

    base = run(0) if N == 1 else None
    if N > 1:
        keep = args.shard_sim
        args.shard_sim = 1
        base = shard_sim(args, device, unet, wrapped, dec, denoiser, noise, c, uc)["unsharded"]
        args.shard_sim = keep
    if N == 1:
        return {"unsharded": base}
    r_first, r_last = run(0), run(N - 1)
    slow = max(r_first["ms_per_sample_compute_only"], r_last["ms_per_sample_compute_only"])
    return {"world": N, "cus": cus, "unsharded": base, "rank_0": r_first, f"rank_{N - 1}": r_last,
            "ideal_speedup_most_loaded_rank": round(T_FRAMES / r_first["frames"], 2),
            "compute_only_strong_scaling_ceiling": round(base["ms_per_sample_compute_only"] / slow, 2),
            "hip_graph_replay_of_the_rank": bool(args.graph),
            "note": "compute of one rank on one GPU with self-fed halos / K|V / statistics (v3d_amd/dist.py SimFrameShard); communication is NOT timed - "
                    "no RCCL run of this path exists (one GPU per build box)"}


def build_models_sampler(device):
    from v3d_amd.sgm.modules.diffusionmodules.denoiser import Denoiser
    from v3d_amd.sgm.modules.diffusionmodules.sampling import EulerEDMSampler
    sampler = EulerEDMSampler(
        discretization_config={"target": P + "discretizer.EDMDiscretization", "params": {"sigma_max": 700.0}}, num_steps=STEPS,
        guider_config={"target": P + "guiders.LinearPredictionGuider", "params": {"max_scale": CFG, "min_scale": CFG, "num_frames": T_FRAMES}}, device=device)
    return sampler, Denoiser({"target": P + "denoiser_scaling.VScalingWithEDMcNoise"})


def make_step_latents(wrapped, sampler, denoiser, noise, c, uc, device):
    extra = {"image_only_indicator": torch.zeros(2, T_FRAMES, device=device), "num_video_frames": T_FRAMES}
    return sampler(lambda i, s, cc: denoiser(wrapped, i, s, cc, **extra), noise, cond=c, uc=uc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--shard", choices=["replica", "frames", "hybrid"], default="replica",
                    help="multi-GPU mode: independent samples per rank (weak scaling, default), ONE sample with its frames sharded over the ranks, "
                         "or cfg-parallel x frame-shard (2 x N/2: the unconditional / conditional halves on two frame groups)")
    ap.add_argument("--workload", choices=["v3d512", "scene"], default="v3d512",
                    help="v3d512 = BASELINE.json configs[1] (the headline: 18 frames, 64 x 64 latents); scene = configs[4] (24 frames, 576 x 1024 -> "
                         "72 x 128 latents; --fp8 switches its spatial self-attention to the e4m3 MFMA kernel).  A scene line is never the headline number.")
    ap.add_argument("--fp8", action="store_true", help="scene workload: fp8 (OCP e4m3) spatial self-attention (V3D_ATTN_FP8=1)")
    ap.add_argument("--inputs", type=int, default=1, help="inputs per sample step (BASELINE.json configs[3]: batch-of-4 inputs); frames per step = inputs x frames")
    ap.add_argument("--edm-steps", dest="edm_steps", type=int, default=STEPS, help="EDM sampler steps (configs[3]: 50)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-parity-rollout", action="store_true")
    ap.add_argument("--shard-timeout", dest="shard_timeout", type=float, default=240.0,
                    help="N > 1 replica mode: seconds the secondary frame-sharded run may take before the line is printed without it")
    ap.add_argument("--shard-sim", dest="shard_sim", type=int, default=0,
                    help="N: measure on THIS GPU what rank 0 (most frames) and rank N-1 (fewest) of an N-GPU frame-sharded run compute per evaluation / decode "
                         "(self-fed exchanges, no communication) and the compute-only strong-scaling ceiling; prints its own JSON line and exits")
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
    scene = args.workload == "scene"
    if args.fp8:
        if not scene:
            raise SystemExit("--fp8 belongs to --workload scene (BASELINE.json configs[4]); the headline config is bf16")
        os.environ["V3D_ATTN_FP8"] = "1"
    T_FR, LH, LW = (24, 72, 128) if scene else (T_FRAMES, LAT, LAT)
    B_IN, E_STEPS = args.inputs, args.edm_steps
    headline = not scene and B_IN == 1 and E_STEPS == STEPS          # roofline / cpu_baseline / parity legs belong to the headline workload
    unet, wrapped, dec, sampler, denoiser = build_models(device, steps=E_STEPS, frames=T_FR)
    shard_mode = args.shard in ("frames", "hybrid") and world > 1
    # replica mode: every rank generates its own sample (different seed per rank); frame-shard mode: ONE sample, same inputs everywhere
    noise, c, uc = synth.synthetic_conditioning(T_FR, LH, LW, seed=23 + (0 if shard_mode else rank), device=device, batch=B_IN)
    if args.shard_sim:
        if world > 1 or scene:
            raise SystemExit("--shard-sim runs on one GPU at the headline shapes (--inputs / --edm-steps select BASELINE.json configs[3]'s batch of inputs)")
        res = shard_sim(args, device, unet, wrapped, dec, denoiser, noise, c, uc)
        print(json.dumps({"metric": f"frame-shard compute simulation, V3D_512 18-frame sample over {args.shard_sim} GPUs (one GPU measured)", "n_gpus": 1,
                          "dtype": "bf16", "data": "synthetic", "config": {"workload": "BASELINE.json configs[2]/[3] sub-problem of one rank: V3D_512, 18 frames "
                          f"sharded over {args.shard_sim} ranks, 64x64 latents, {B_IN} input(s) = guided batch of {2 * B_IN}, {E_STEPS} EulerEDM steps, 18-frame decode"}, "shard_sim": res}), flush=True)
        return
    shard = None
    if world > 1:
        from v3d_amd.dist import FrameShard, HybridShard
        shard = HybridShard(T_FR) if args.shard == "hybrid" else FrameShard(T_FR)
    if shard_mode:
        step = make_sharded_step(shard, wrapped, dec, sampler, denoiser, noise, c, uc, B=B_IN)
    else:
        step = make_step(wrapped, dec, sampler, denoiser, noise, c, uc, device, graph=args.graph, frames=T_FR, inputs=B_IN)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, warmup, steps):
        for _ in range(warmup):
            out = fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = fn()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return out, dt

    if shard is not None:
        shard.bytes_sent = shard.n_exchanges = shard.n_allreduce = 0
    out, dt = timed(step, args.warmup, args.steps)
    assert out.shape == (B_IN * T_FR, 3, LH * 8, LW * 8) and torch.isfinite(out).all()
    from v3d_amd.ops import get_ops as _get_ops
    _get_ops().check_health()                 # a stream-K hand-off that gave up would have produced a wrong tile silently: fail the run instead

    result = None
    samples = args.steps * (1 if shard_mode else world)
    ranks_seen = 1
    if world > 1:
        # how many ranks actually took part (an all-reduce of ones over RCCL): a SCALE record then shows N ranks were seen, not just requested
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
    if rank == 0:
        frames = samples * B_IN * T_FR
        par = f"frame-shard {shard.describe()}" if shard_mode else f"replica x{world}"
        if scene:
            wl = (f"BASELINE.json configs[4] shape: sparse-view scene config, random-init SVD-XT weights, {B_IN}x{T_FR}x4x{LH}x{LW} latent ({8 * LH}x{8 * LW} "
                  f"frames), {E_STEPS} EulerEDM steps, cfg 4.5, {T_FR}-frame VideoDecoder decode, spatial self-attention "
                  + ("fp8 e4m3 MFMA (V3D_ATTN_FP8=1)" if args.fp8 else "bf16") + " - NOT the headline workload")
        else:
            wl = (f"BASELINE.json configs[{1 if headline else 3}]: V3D_512 random-init SVD-XT weights, {B_IN}x{T_FR}x4x{LH}x{LW} latent, {E_STEPS} EulerEDM steps, "
                  f"cfg 4.5 (LinearPredictionGuider), {T_FR}-frame VideoDecoder decode to 512x512"
                  + (f", batch of {B_IN} inputs per step" if B_IN > 1 else "") + ", "
                  + ("ONE sample, frames sharded over the GPUs (configs[2]/[3])" if shard_mode else "one sample per GPU"))
        result = {
            "metric": "multi-view frames/sec, V3D_512 18-frame 25-step EDM" if headline else
                      f"multi-view frames/sec, {'scene 24-frame' if scene else 'V3D_512 18-frame'} {E_STEPS}-step EDM, {B_IN} input(s) per step",
            "value": round(frames / dt, 4), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "strong" if shard_mode else "weak", "vs_baseline": None,
            "dtype": "bf16 (fp8 e4m3 spatial attention)" if args.fp8 else "bf16", "data": "synthetic",
            "config": {"workload": wl, "frames": T_FR, "inputs_per_step": B_IN, "edm_steps": E_STEPS, "latent": [B_IN * T_FR, 4, LH, LW], "parallelism": par},
        }
        if world > 1:
            try:
                ver = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:      # noqa: BLE001
                ver = "unknown"
            result["collectives"] = {"backend": f"torch.distributed nccl = RCCL {ver}", "ranks_requested": world, "ranks_seen": ranks_seen,
                                     "data_path": ("grouped point-to-point (K|V all-gather, halo + fp64 statistics) between the frame-shard ranks" if shard_mode
                                                   else "none (independent samples per rank); barrier + max-over-ranks timing only")}
        if headline:
            sample_tflop = STEPS * F_UNET_TFLOP + T_FRAMES * F_VAE_TFLOP_PER_FRAME
            result["achieved_tflops_reference_graph"] = round(samples * sample_tflop / dt, 1)
            result["frac_of_bf16_peak_reference_graph"] = round(samples * sample_tflop / dt / PEAK_BF16_TFLOPS / world, 4)
        if shard_mode:
            n_eval = (args.steps + args.warmup) * E_STEPS
            cnt = shard.counters()
            result["frame_shard_exchanges"] = {
                "grouped_p2p_calls_per_evaluation_rank0": round(cnt["grouped_p2p_calls"] / n_eval, 1), "all_reduces_per_evaluation": cnt["all_reduces"] / n_eval,
                "sent_MB_per_sample_rank0": round(cnt["bytes_sent"] / (args.steps + args.warmup) / 1e6, 1),
                "note": "one grouped point-to-point launch per temporal norm + conv (raw halo + fp64 statistics) and per temporal attention (K|V); "
                        "NO RCCL timing of this path existed before this run: the build boxes have one GPU"}
    if rank == 0 and not args.no_roofline and headline:
        # per-launch HIP events need the launches to come from Python: the instrumented sample runs un-captured (and un-sharded)
        noise0, c0, uc0 = synth.synthetic_conditioning(T_FRAMES, LAT, LAT, seed=23, device=device)
        result["roofline"] = measure_rooflines(make_step(wrapped, dec, sampler, denoiser, noise0, c0, uc0, device, graph=False))
        result["config"]["hip_graph"] = bool(args.graph)
    elif rank == 0 and not args.no_roofline:
        # other workloads: the same live per-op-family table (its own bounds); the fp8 attention family is priced against the fp8 MFMA peak
        noise0, c0, uc0 = synth.synthetic_conditioning(T_FR, LH, LW, seed=23, device=device, batch=B_IN)
        rl = measure_rooflines(make_step(wrapped, dec, sampler, denoiser, noise0, c0, uc0, device, graph=False, frames=T_FR, inputs=B_IN))
        for k in rl["per_kernel"]:
            if k["kernel"] == "attn_spatial_fp8":
                k["frac"] = round(k["achieved"] / PEAK_FP8_TFLOPS, 4)
                k["peak"] = PEAK_FP8_TFLOPS
        rl.pop("traffic", None); rl.pop("traffic_profile", None); rl.pop("traffic_unit", None)
        result["roofline"] = rl
    if world > 1:
        dist.barrier()
    # ---- N > 1, replica mode: the frame-sharded (latency) mode of the same job, measured LAST and under a watchdog: it is a secondary
    #      number and its exchanges (grouped RCCL P2P between all ranks) must never cost the run its replica result - if it has not
    #      finished in time, rank 0 prints the line it has and every rank leaves without waiting for the others ----
    printed = [False]
    if world > 1 and not shard_mode and headline:
        import threading

        def bail():
            if rank == 0 and not printed[0]:
                result["frame_shard"] = {"value": None, "error": f"frame-sharded secondary run did not finish within {args.shard_timeout} s (abandoned)"}
                printed[0] = True
                print(json.dumps(result), flush=True)
            os._exit(0)

        watchdog = threading.Timer(args.shard_timeout, bail)
        watchdog.daemon = True
        watchdog.start()
        fs = None
        try:
            n_fs = max(1, min(args.steps, 3))
            noise0, c0, uc0 = synth.synthetic_conditioning(T_FRAMES, LAT, LAT, seed=23, device=device)
            sstep = make_sharded_step(shard, wrapped, dec, sampler, denoiser, noise0, c0, uc0)
            sent0 = shard.bytes_sent
            fout, fdt = timed(sstep, 1, n_fs)
            ok = bool(fout.shape == (T_FRAMES, 3, LAT * 8, LAT * 8) and torch.isfinite(fout).all())
            fs = {"value": round(n_fs * T_FRAMES / fdt, 4), "unit": "frames/s", "ms_per_sample": round(fdt / n_fs * 1e3, 2), "steps": n_fs, "warmup": 1,
                  "scaling": "strong", "parallelism": f"frame-shard {shard.describe()}", "finite": ok,
                  "sent_MB_per_sample_rank0": round((shard.bytes_sent - sent0) / (n_fs + 1) / 1e6, 1),
                  "note": "ONE sample, frames sharded over the ranks for all 25 steps + decode (v3d_amd/dist.py::sharded_sample); not part of `value`"}
        except Exception as e:   # never lose the replica number to the secondary measurement
            fs = {"value": None, "error": f"{type(e).__name__}: {e}"[:400]}
        if rank == 0 and not printed[0]:
            result["frame_shard"] = fs
            printed[0] = True
            print(json.dumps(result), flush=True)
        try:                      # (a rank that failed alone would wait here for ever: the watchdog is still armed)
            dist.barrier()
            dist.destroy_process_group()
        finally:
            watchdog.cancel()
        return
    if rank == 0 and world == 1 and not args.no_cpu_baseline and headline:
        try:
            result["cpu_baseline"] = cpu_baseline(unet, dec)
        except Exception as e:  # the baseline is informational; never lose the GPU number to a host-side failure
            result["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port", "sample": f"failed: {e}"}
        if not args.no_parity_rollout:
            try:
                del unet, wrapped, dec
                result["cpu_baseline"]["parity_rollout"] = parity_rollout(device)
            except Exception as e:
                result["cpu_baseline"]["parity_rollout"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
