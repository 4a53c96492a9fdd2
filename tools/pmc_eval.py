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
for _ in range(N_EVAL):
    denoiser(wrapped, x, sig, cond, **extra)
torch.cuda.synchronize()
# calibration: 4 x (256 MiB read + 256 MiB write) through the library's own 2-D copy kernel (16-byte lanes, coalesced)
from v3d_amd.ops import get_ops
ops = get_ops()
a = torch.randn(1 << 17, 1024, device=dev).bfloat16()   # 256 MiB
b = torch.empty_like(a)
for _ in range(4):
    ops.copy2d_bf16(a, b)
torch.cuda.synchronize()
print("pmc_eval done", N_EVAL)
