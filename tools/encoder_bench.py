"""Time the full-width VAE Encoder (1.08 TFLOP per 512x512 image) on the HIP kernels."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from v3d_amd import synth
from v3d_amd.sgm.modules.diffusionmodules.model import Encoder
torch.set_grad_enabled(False)
with torch.device("cuda"):
    enc = Encoder(**synth.encoder_config(128)).eval()
synth.init_module_fast(enc, seed=3)
for n in (1, 4, 18):
    x = torch.rand(n, 3, 512, 512, device="cuda") * 2 - 1
    enc(x); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        enc(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"encoder n={n}: {dt * 1e3:.1f} ms  ({n / dt:.1f} images/s, {1.08 * n / dt:.0f} TFLOP/s algorithmic)")
