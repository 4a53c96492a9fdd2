"""Per-op timing inside one full-size VideoDecoder decode (18 frames -> 512x512): every primitive launch with its shape."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from v3d_amd.ops import get_ops
torch.set_grad_enabled(False)
dev = "cuda"
import bench
unet, wrapped, dec, sampler, denoiser = bench.build_models(dev)
ops = get_ops()
rec = []
names = [n for n in dir(ops) if not n.startswith("_") and callable(getattr(ops, n)) and n not in ("name",)]
def wrap(name):
    orig = getattr(ops, name)
    def f(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = orig(*a, **k); e1.record()
        key = name
        if name == "gemm":
            g = a[0]
            key = f"gemm m{g.mode} M={g.M} N={g.N} K={g.K} b={g.batch}"
            fl = 2.0 * g.M * g.N * g.K * {0: 1, 1: 9, 2: 3}[g.mode] * g.batch
        else:
            shp = [tuple(t.shape) for t in a if isinstance(t, torch.Tensor)][:1]
            key = f"{name} {shp}"
            fl = 0
        rec.append((key, e0, e1, fl))
        return r
    setattr(ops, name, f)
for n in ("gemm", "groupnorm_stats", "groupnorm_apply", "layernorm", "attn_spatial", "attn_temporal", "softmax_rows", "tmix_small", "copy2d_bf16",
          "nchw_to_nhwc_bf16", "silu_add", "axpb_f32"):
    if hasattr(ops, n):
        wrap(n)
z = torch.randn(18, 4, 64, 64, device=dev)
for it in range(2):
    rec.clear()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); dec(z, timesteps=18); t1.record()
    torch.cuda.synchronize()
agg = collections.OrderedDict()
for k, a, b, fl in rec:
    d = agg.setdefault(k, [0, 0.0, 0.0]); d[0] += 1; d[1] += a.elapsed_time(b); d[2] += fl
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    tf = f"{v[2] / v[1] / 1e9:7.0f} TF/s" if v[2] else ""
    print(f"{v[1]:8.3f} ms {v[1] / tot * 100:5.1f}% n={v[0]:3d} avg={v[1] / v[0] * 1e3:8.1f}us {tf}  {k}")
print(f"ops total {tot:.2f} ms over {len(rec)} launches; decode wall {t0.elapsed_time(t1):.2f} ms")
