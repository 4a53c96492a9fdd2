"""Interleaved same-process A/B of several builds of libv3d_hip.so on the spatial self-attention launches of the V3D_512 evaluation
(tools/mainloop_ab.py for the attention kernels): python tools/attn_ab.py base=path new=path ... -> median us, TF/s, max |diff| vs the first."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from v3d_amd.hip import HipOps  # noqa: E402

libs = [a.split("=", 1) for a in sys.argv[1:] if "=" in a and not a.startswith("--")]
hips = [(t, HipOps(lib_path=os.path.join(ROOT, p) if not os.path.isabs(p) else p)) for t, p in libs]
BF = torch.bfloat16
print(f"{'case':22s}" + "".join(f"{t:>10s}" for t, _ in hips))
for n, S, h in ((36, 4096, 5), (36, 1024, 10), (36, 256, 20)):
    C = h * 64
    g = torch.Generator(device="cuda").manual_seed(1)
    q = torch.randn(n * S, C, device="cuda", generator=g).to(BF)
    k = torch.randn(n * S, C, device="cuda", generator=g).to(BF)
    vT = torch.randn(n, C, S, device="cuda", generator=g).to(BF)
    outs = [torch.zeros(n * S, C, dtype=BF, device="cuda") for _ in hips]
    for (t, hp), o in zip(hips, outs):
        hp.attn_spatial(q, k, vT, o, n, S, h, 0.125)
        hp.attn_spatial(q, k, vT, o, n, S, h, 0.125)
    torch.cuda.synchronize()
    diffs = [float((o.float() - outs[0].float()).abs().max()) for o in outs]
    times = [[] for _ in hips]
    for r in range(7):
        order = list(range(len(hips)))
        if r % 2:
            order.reverse()
        for i in order:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                hips[i][1].attn_spatial(q, k, vT, outs[i], n, S, h, 0.125)
            e1.record()
            torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1) / 4 * 1e3)
    med = [statistics.median(t) for t in times]
    fl = 4.0 * n * h * S * S * 64
    print(f"n={n} S={S} h={h}".ljust(22) + "".join(f"{m:10.1f}" for m in med) + f"   {fl / med[0] / 1e6:6.0f} TF/s  diff " + " ".join(f"{d:.3g}" for d in diffs), flush=True)
