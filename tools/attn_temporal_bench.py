"""Temporal attention at the V3D_512 shapes: time per launch and effective HBM rate (q, k, v read + out written once).
V3D_ATTN_TEMPORAL_IMPL=1 (VALU dot2 kernel of rounds 1-4) / 2 (MFMA kernel, default) is read once per process: run twice for the A/B."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from v3d_amd.ops import get_ops
torch.set_grad_enabled(False)
ops = get_ops()
dev = "cuda"
print("impl", os.environ.get("V3D_ATTN_TEMPORAL_IMPL", "2 (default)"))
for (B, T, S, C) in [(2, 18, 4096, 320), (2, 18, 1024, 640), (2, 18, 256, 1280), (2, 18, 64, 1280), (8, 18, 4096, 320)]:
    heads = C // 64
    g = torch.Generator(device=dev).manual_seed(1)
    qkv = torch.randn(B, T, S, 3 * C, device=dev, generator=g).to(torch.bfloat16)
    out = torch.empty(B, T, S, C, device=dev, dtype=torch.bfloat16)
    f = lambda: ops.attn_temporal(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], out, heads, 0.125)
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    byts = B * T * S * C * 2 * 4
    print(f"attn_temporal B={B} T={T} S={S:5d} C={C:5d}  {us:8.1f} us  {byts / us / 1e6:6.2f} TB/s  checksum {out.float().abs().mean().item():.6f}")
