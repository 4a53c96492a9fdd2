"""Probe (round 6): do the two halves of the guided batch (unconditional / conditional sample: independent through the whole U-Net) run faster as two
CONCURRENT evaluations of 18 images on two HIP streams than as one evaluation of 36 images?  Every persistent kernel ends in a partial round and every
small kernel is latency-bound; a second stream's kernels can fill those holes.  Each stream gets its own backend object (the zeroed scratch arena and
the library's split-K / stream-K workspaces are per evaluation / per stream).  Prints ms per guided evaluation for both forms + max |diff|."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from v3d_amd.hip import HipOps
from v3d_amd.ops import use_backend
torch.set_grad_enabled(False)
dev = "cuda:0"
T = 18
unet, wrapped, dec, sampler, denoiser = bench.build_models(dev)
g = torch.Generator(device=dev).manual_seed(3)
x = torch.randn(2 * T, 8, 64, 64, device=dev, generator=g)
ts = torch.rand(2 * T, device=dev, generator=g) * 5 - 2
ctx = torch.randn(2 * T, 1, 1024, device=dev, generator=g)
y = torch.randn(2 * T, 768, device=dev, generator=g)
ioi = torch.zeros(2, T, device=dev)


def full():
    return unet(x, ts, context=ctx, y=y, num_video_frames=T, image_only_indicator=ioi)


s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
o0, o1 = HipOps(), HipOps()
halves = [(x[:T].contiguous(), ts[:T].contiguous(), ctx[:T].contiguous(), y[:T].contiguous(), ioi[:1].contiguous()),
          (x[T:].contiguous(), ts[T:].contiguous(), ctx[T:].contiguous(), y[T:].contiguous(), ioi[1:].contiguous())]


def split():
    outs = []
    cur = torch.cuda.current_stream()
    for s, o, h in ((s0, o0, halves[0]), (s1, o1, halves[1])):
        s.wait_stream(cur)
        with torch.cuda.stream(s), use_backend(o):
            outs.append(unet(h[0], h[1], context=h[2], y=h[3], num_video_frames=T, image_only_indicator=h[4]))
    cur.wait_stream(s0)
    cur.wait_stream(s1)
    return torch.cat(outs, dim=0)


def split_serial():      # the same two half evaluations back to back on ONE stream: what the halving alone costs
    outs = []
    for h in halves:
        outs.append(unet(h[0], h[1], context=h[2], y=h[3], num_video_frames=T, image_only_indicator=h[4]))
    return torch.cat(outs, dim=0)


def timeit(fn, n=6):
    for _ in range(2):
        out = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


for rnd in range(2):
    tf, of = timeit(full)
    t2, o2 = timeit(split)
    t1, o1_ = timeit(split_serial)
    d = float((of.float() - o2.float()).abs().max()), float(of.float().abs().max())
    print(f"round {rnd}: one evaluation of 36 images {tf:7.2f} ms | two of 18 on two streams {t2:7.2f} ms | two of 18 on one stream {t1:7.2f} ms | max |diff| {d[0]:.3e} of {d[1]:.3e}", flush=True)
