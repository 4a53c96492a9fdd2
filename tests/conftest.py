import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return torch.load(os.path.join(ROOT, "tests", "golden", "v3d_tiny.pt"))


@pytest.fixture(scope="session")
def hip_ops():
    """The product backend; fails loudly (no skip) when the extension or the GPU is missing on a -m gpu run."""
    from v3d_amd.hip import HipOps
    return HipOps()


def rel_cos(a: torch.Tensor, b: torch.Tensor):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()      # (fp32 dot products of 10^7 elements drift past 1.0)
    rel = ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()
    cos = torch.nn.functional.cosine_similarity(a, b, dim=0).item()
    return rel, cos


@pytest.fixture(scope="session")
def full_unet():
    """The BASELINE.json configs[1] network (width 320, random-init SVD-XT architecture) on the GPU, built once per session."""
    from v3d_amd import synth
    from v3d_amd.sgm.modules.diffusionmodules.video_model import VideoUNet
    with torch.device("cuda"):
        net = VideoUNet(**synth.unet_config(320)).eval()
    synth.init_module_fast(net, seed=1)
    return net


def full_inputs(n, seed, H=64, W=64):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(n, 8, H, W, generator=g), torch.randn(n, generator=g), torch.randn(n, 1, 1024, generator=g),
            torch.randn(n, 768, generator=g))


def psnr(a: torch.Tensor, b: torch.Tensor):
    """PSNR in dB of `a` against the reference `b`, peak = value range of the reference."""
    a, b = a.float().cpu(), b.float().cpu()
    mse = ((a - b) ** 2).mean().item()
    peak = (b.max() - b.min()).item()
    import math
    return float("inf") if mse == 0 else 10.0 * math.log10(peak * peak / mse)


def record_parity(name: str, values: dict):
    """Measured parity numbers of the -m gpu run, merged into gpurun_out/parity.json (copied to profiles/ when committed)."""
    import json
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, "parity.json")
    data = {}
    if os.path.exists(path):
        try:
            data = json.load(open(path))
        except Exception:
            data = {}
    data[name] = values
    json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    print(f"[parity] {name}: {values}")
