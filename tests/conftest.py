import contextlib
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


# ---- the fp32 oracle executed on the GPU ---------------------------------------------------------------------------------------------
# oracle/sgm_oracle.py is a pure functional torch restatement: given tensors on a device it runs there.  The full-width (-m gpu) parity
# tests hand it cuda tensors, so the CHECKER runs as fp32 ATen kernels on the GPU (rocBLAS fp32 GEMMs, the native im2col convolutions -
# MIOpen is switched off: this image carries no gfx950 find-db and every first convolution would JIT-compile -, the math SDPA backend,
# TF32 off) instead of ~15 minutes of CPU time per suite run (GPUTEST_r04: the driver's 1200 s limit hit after 144 of 168 tests, ~900 s of
# them the CPU oracle).  The device-executed oracle is itself pinned: tests/test_headline_parity_gpu.py::test_device_oracle_is_pinned holds
# it to the reference-generated fixtures (tests/golden/v3d_tiny.pt) and to the CPU-executed oracle at rtol 1e-4.
# V3D_ORACLE_DEVICE=cpu restores the CPU-executed checker everywhere.
ORACLE_DEVICE = os.environ.get("V3D_ORACLE_DEVICE", "cuda")


@contextlib.contextmanager
def device_oracle():
    """with device_oracle() as dev: ref = O.unet_forward(odev(sd, dev), cfg, *odev(inputs, dev)).cpu()"""
    if ORACLE_DEVICE == "cpu":
        yield "cpu"
        return
    from torch.nn.attention import SDPBackend, sdpa_kernel
    old_mm, old_cd = torch.backends.cuda.matmul.allow_tf32, torch.backends.cudnn.allow_tf32
    old_prec = torch.get_float32_matmul_precision()
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    torch.set_float32_matmul_precision("highest")
    try:
        with torch.backends.cudnn.flags(enabled=False), sdpa_kernel(SDPBackend.MATH):
            yield ORACLE_DEVICE
    finally:
        torch.backends.cuda.matmul.allow_tf32, torch.backends.cudnn.allow_tf32 = old_mm, old_cd
        torch.set_float32_matmul_precision(old_prec)
        torch.cuda.empty_cache()


def odev(obj, dev):
    """Tensors (also inside dicts / lists / tuples) as fp32 on the oracle's device; everything else unchanged."""
    if torch.is_tensor(obj):
        return obj.detach().to(device=dev, dtype=torch.float32) if obj.is_floating_point() else obj.detach().to(dev)
    if isinstance(obj, dict):
        return {k: odev(v, dev) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(odev(v, dev) for v in obj)
    return obj


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
