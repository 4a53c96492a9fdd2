"""Per-op timing inside one full-size VideoUNet evaluation: every primitive launch with its shape (HIP events)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from v3d_amd import synth
from v3d_amd.ops import get_ops
torch.set_grad_enabled(False)
dev = "cuda"
import bench
unet, wrapped, dec, sampler, denoiser = bench.build_models(dev)
noise, c, uc = synth.synthetic_conditioning(18, 64, 64, seed=23, device=dev)
ops = get_ops()
rec = []
def wrap(name, keyfn):
    orig = getattr(ops, name)
    def f(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = orig(*a, **k); e1.record()
        key = keyfn(*a, **k)
        if name == "gemm":       # + the library's own record of the kernel that took the launch (family, tile, tiles, fill)
            li = ops.last_gemm_launch()
            fam = {1: "v1", 2: "v2", 3: "v3", 5: "halo", 6: "v6"}.get(li["family"], "?") + (f"/sk{li['splitk']}" if li["splitk"] > 1 else "") + ("/tail" if li["streamk_tail"] else "")
            key = (key[0] + f"  <{fam} {li['bm']}x{li['bn']} tiles={li['tiles']} x{li['blocks_per_cu']}/CU fill={li['fill']:.2f}>", key[1])
        rec.append((key, e0, e1))
        return r
    setattr(ops, name, f)
def gk(g):
    taps = {0: 1, 1: 9, 2: 3}[g.mode]
    ep = ("N" if g.gn_in is not None else "") + ("S" if g.gn_stats is not None else "") + ("G" if g.geglu else "") + ("b" if g.bias is not None else "") + ("a" if g.add is not None else "") + ("r" if g.res1 is not None else "") + ("R" if g.res2 is not None else "") + ("c" if g.coef is not None else "") + ("f" if g.out.dtype == torch.float32 else "")
    return f"gemm m{g.mode} M={g.M} N={g.N} K={g.K} b={g.batch} [{ep}]", 2.0 * g.M * g.N * g.K * taps * g.batch
wrap("gemm", gk)
wrap("groupnorm_stats", lambda x1, x2, st, n, S, g, ips: (f"gn_stats n={n} S={S} C={x1.shape[-1] + (x2.shape[-1] if x2 is not None else 0)} ips={ips}", 0))
if hasattr(ops, "groupnorm_stats_table"):
    wrap("groupnorm_stats_table", lambda x1, x2, st, tk, n, S, g, ips, *a: (f"gn_stats+table n={n} S={S} C={x1.shape[-1] + (x2.shape[-1] if x2 is not None else 0)} ips={ips}", 0))
wrap("groupnorm_apply", lambda x1, x2, *a: (f"gn_apply rows={x1.shape[0]} C={x1.shape[-1] + (x2.shape[-1] if x2 is not None else 0)}", 0))
wrap("groupnorm_finalize", lambda st, sums, *a: (f"gn_finalize n_stat={st.shape[0]} slots={st.shape[1]}", 0))
for nm in ("ff_fused", "ln_ff_fused", "ln_proj"):
    if hasattr(ops, nm):
        wrap(nm, (lambda nm_: lambda *a, **k: (f"{nm_} rows={a[0].shape[0]}", 0))(nm))
wrap("layernorm", lambda x, *a, **k: (f"ln rows={x.shape[0]} C={x.shape[-1]} add={'add' in k}", 0))
wrap("attn_spatial", lambda q, k, vT, out, n, S, h, sc: (f"attn_spatial n={n} S={S} h={h}", 4.0 * n * h * S * S * 64))
wrap("attn_temporal", lambda q, k, v, out, h, sc: (f"attn_temporal {tuple(q.shape)}", 0))
x = torch.cat([noise, noise]); sig = torch.full((36,), 10.0, device=dev)
cond = {k: torch.cat([uc[k], c[k]]) for k in c}
extra = {"image_only_indicator": torch.zeros(2, 18, device=dev), "num_video_frames": 18}
for it in range(2):
    rec.clear()
    denoiser(wrapped, x, sig, cond, **extra)
    torch.cuda.synchronize()
agg = collections.OrderedDict()
for (k, fl), a, b in rec:
    d = agg.setdefault(k, [0, 0.0, 0.0]); d[0] += 1; d[1] += a.elapsed_time(b); d[2] += fl
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:400]:
    tf = f"{v[2] / v[1] / 1e9:7.0f} TF/s" if v[2] else ""
    print(f"{v[1]:8.3f} ms {v[1] / tot * 100:5.1f}% n={v[0]:3d} avg={v[1] / v[0] * 1e3:8.1f}us {tf}  {k}")
print(f"total {tot:.2f} ms over {len(rec)} launches")
# GEMM-family time by kernel family (what the dispatcher of gemm.hip sent where)
fam = collections.OrderedDict()
for k, v in agg.items():
    if k.startswith("gemm "):
        f = k[k.index("<") + 1:].split()[0]
        d = fam.setdefault(f, [0, 0.0, 0.0]); d[0] += v[0]; d[1] += v[1]; d[2] += v[2]
for f, v in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f"family {f:10s} {v[1]:8.3f} ms  n={v[0]:4d}  {v[2] / v[1] / 1e9:7.0f} TF/s")
