"""Per-block timing of one full-size VideoUNet evaluation + VAE decode (HIP events around every op family)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from v3d_amd import synth
from v3d_amd.ops import get_ops
from v3d_amd.engine import unet as U, blocks as B, vae as V
torch.set_grad_enabled(False)
dev = "cuda"
import bench
unet, wrapped, dec, sampler, denoiser = bench.build_models(dev)
noise, c, uc = synth.synthetic_conditioning(18, 64, 64, seed=23, device=dev)
ops = get_ops()
rec = collections.OrderedDict()
def wrap(mod, name, keyfn):
    orig = getattr(mod, name)
    def f(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = orig(*a, **k); e1.record()
        rec.setdefault(keyfn(*a, **k), []).append((e0, e1))
        return r
    setattr(mod, name, f)
wrap(U, "unet_resblock", lambda env, g, p, *a, **k: f"res  {g.H:3d}x{g.W:<3d} {p.cin:4d}->{p.cout:4d}")
wrap(U, "run_svt", lambda env, g, p, x: f"svt  {g.H:3d}x{g.W:<3d} C={p.C}")
wrap(V, "vae_resblock", lambda env, g, p, x: f"vres {g.H:3d}x{g.W:<3d} {p.cin:4d}->{p.cout:4d}")
wrap(V, "run_vae_attn", lambda env, g, p, x: f"vattn {g.H}x{g.W}")
x = torch.cat([noise, noise]); sig = torch.full((36,), 10.0, device=dev)
cond = {k: torch.cat([uc[k], c[k]]) for k in c}
extra = {"image_only_indicator": torch.zeros(2, 18, device=dev), "num_video_frames": 18}
for it in range(2):
    rec.clear()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record(); denoiser(wrapped, x, sig, cond, **extra); t1.record()
    d0, d1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    d0.record(); dec(noise, timesteps=18); d1.record()
    torch.cuda.synchronize()
tot = 0
for k, v in rec.items():
    ms = sum(a.elapsed_time(b) for a, b in v)
    tot += ms
    print(f"{k:32s} n={len(v):2d} total={ms:8.3f} ms  avg={ms/len(v):7.3f}")
print(f"unet eval {t0.elapsed_time(t1):.2f} ms, decode {d0.elapsed_time(d1):.2f} ms, sum of wrapped blocks {tot:.2f} ms")
