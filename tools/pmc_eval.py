"""Workload for the rocprofv3 --pmc passes: N_EVAL full-size guided VideoUNet evaluations + a calibration copy of known size.
(rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python tools/pmc_eval.py ; again with WRITE_SIZE ; then tools/pmc_traffic.py)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.set_grad_enabled(False)
import bench
from v3d_amd import synth
dev = "cuda"
N_EVAL = 2
unet, wrapped, dec, sampler, denoiser = bench.build_models(dev)
noise, c, uc = synth.synthetic_conditioning(18, 64, 64, seed=23, device=dev)
x = torch.cat([noise, noise]); sig = torch.full((36,), 10.0, device=dev)
cond = {k: torch.cat([uc[k], c[k]]) for k in c}
extra = {"image_only_indicator": torch.zeros(2, 18, device=dev), "num_video_frames": 18}
# record the v3d_gemm / v3d_ff_fused calls of the evaluations in launch order (shape + algorithmic bytes): tools/pmc_per_launch.py joins
# them with the per-dispatch counters of the same run
from v3d_amd.ops import get_ops
_ops = get_ops()
_calls = []
_orig_gemm, _orig_ff, _orig_lnff = _ops.gemm, _ops.ff_fused, _ops.ln_ff_fused


def _rec_gemm(g):
    taps = {0: 1, 1: 9, 2: 3}[g.mode]
    nout = g.N // 2 if g.geglu else g.N
    rd = (g.A.shape[-2] * g.K * 2 + taps * g.N * g.K * 2) * g.batch + sum(g.M * nout * 2 for r in (g.res1, g.res2) if r is not None)
    wr = g.M * nout * g.out.element_size() * g.batch
    _calls.append({"kind": "gemm", "mode": g.mode, "M": g.M, "N": g.N, "K": g.K, "batch": g.batch, "geglu": bool(g.geglu),
                   "res": int(g.res1 is not None) + int(g.res2 is not None), "alg_read": rd, "alg_write": wr, "flop": 2.0 * g.M * g.N * g.K * taps * g.batch})
    return _orig_gemm(g)


def _rec_ff(xx, w1p, b1, w2p, b2, out, **kw):
    M, C, hidden = xx.shape[0], xx.shape[1], w2p.shape[-1]
    nres = sum(1 for k in ("res1", "res2") if kw.get(k) is not None)
    _calls.append({"kind": "ff_fused", "mode": 0, "M": M, "N": C, "K": hidden, "batch": 1, "geglu": True, "res": nres,
                   "alg_read": M * C * 2 + 3 * C * hidden * 2 + nres * M * C * 2, "alg_write": M * C * 2, "flop": 6.0 * M * C * hidden})
    return _orig_ff(xx, w1p, b1, w2p, b2, out, **kw)


def _rec_lnff(xx, eps, w1p, b1, w2p, b2, out, **kw):       # v3d_ln_ff_fused: the same kernel with the LayerNorm on its resident rows
    M, C, hidden = xx.shape[0], xx.shape[1], w2p.shape[-1]
    nres = sum(1 for k in ("res1", "res2") if kw.get(k) is not None)
    _calls.append({"kind": "ff_fused", "mode": 0, "M": M, "N": C, "K": hidden, "batch": 1, "geglu": True, "res": nres,
                   "alg_read": M * C * 2 + 3 * C * hidden * 2 + nres * M * C * 2, "alg_write": M * C * 2, "flop": 6.0 * M * C * hidden})
    return _orig_lnff(xx, eps, w1p, b1, w2p, b2, out, **kw)


_ops.gemm, _ops.ff_fused, _ops.ln_ff_fused = _rec_gemm, _rec_ff, _rec_lnff
# window marker (tools/pmc_traffic.py counts only dispatches behind the first copy2d): everything above - model build, packing, input
# synthesis, one warm-up evaluation that triggers the lazy packing - is setup
denoiser(wrapped, x, sig, cond, **extra)
_calls.clear()
torch.cuda.synchronize()
_mk = torch.zeros(16, 64, device=dev).bfloat16()
_ops.copy2d_bf16(_mk, torch.empty_like(_mk))
for _ in range(N_EVAL):
    denoiser(wrapped, x, sig, cond, **extra)
torch.cuda.synchronize()
_ops.__dict__.pop("gemm", None); _ops.__dict__.pop("ff_fused", None); _ops.__dict__.pop("ln_ff_fused", None)
import json
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"n_eval": N_EVAL, "calls": _calls}, open(os.path.join(ROOT, "gpurun_out", "pmc_eval_calls.json"), "w"))
# calibration: 4 x (256 MiB read + 256 MiB write) through the library's own 2-D copy kernel (16-byte lanes, coalesced)
from v3d_amd.ops import get_ops
ops = get_ops()
a = torch.randn(1 << 17, 1024, device=dev).bfloat16()   # 256 MiB
b = torch.empty_like(a)
for _ in range(4):
    ops.copy2d_bf16(a, b)
torch.cuda.synchronize()
print("pmc_eval done", N_EVAL)
