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
    a, b = a.float().flatten().cpu(), b.float().flatten().cpu()
    rel = ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()
    cos = torch.nn.functional.cosine_similarity(a, b, dim=0).item()
    return rel, cos
