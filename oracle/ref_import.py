"""TEST INFRASTRUCTURE — import the reference's own hot-path modules from /root/reference, unmodified.

Only usable in the build container (the GPU box has no /root/reference).  Used by oracle/gen_golden.py to pin
the oracle restatement and to emit tests/golden/*.pt.  Recipe: SURVEY.md Appendix C — stub the two missing
third-party modules and pre-register bare namespace packages so `sgm/__init__.py` (pytorch_lightning, kornia,
open_clip) is never executed.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

REF_ROOT = os.environ.get("V3D_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "sgm"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m


_loaded = {}


def load():
    """Returns a dict of reference modules: video_model, sampling, denoiser, wrappers, temporal_ae, guiders, ..."""
    if _loaded:
        return _loaded
    if not available():
        raise RuntimeError(f"reference tree not found at {REF_ROOT}")
    if "mediapy" not in sys.modules:
        _stub("mediapy", write_image=lambda *a, **k: None, write_video=lambda *a, **k: None)
    if "omegaconf" not in sys.modules:
        _stub("omegaconf", ListConfig=type("ListConfig", (list,), {}), OmegaConf=type("OmegaConf", (dict,), {}))
    for pkg, path in [("sgm", "sgm"), ("sgm.modules", "sgm/modules"),
                      ("sgm.modules.diffusionmodules", "sgm/modules/diffusionmodules"),
                      ("sgm.modules.autoencoding", "sgm/modules/autoencoding")]:
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = [os.path.join(REF_ROOT, path)]
            sys.modules[pkg] = m
    names = {
        "util": "sgm.util",
        "video_model": "sgm.modules.diffusionmodules.video_model",
        "openaimodel": "sgm.modules.diffusionmodules.openaimodel",
        "sampling": "sgm.modules.diffusionmodules.sampling",
        "guiders": "sgm.modules.diffusionmodules.guiders",
        "discretizer": "sgm.modules.diffusionmodules.discretizer",
        "denoiser": "sgm.modules.diffusionmodules.denoiser",
        "denoiser_scaling": "sgm.modules.diffusionmodules.denoiser_scaling",
        "wrappers": "sgm.modules.diffusionmodules.wrappers",
        "model": "sgm.modules.diffusionmodules.model",
        "attention": "sgm.modules.attention",
        "video_attention": "sgm.modules.video_attention",
        "temporal_ae": "sgm.modules.autoencoding.temporal_ae",
    }
    for k, v in names.items():
        _loaded[k] = importlib.import_module(v)
    return _loaded
