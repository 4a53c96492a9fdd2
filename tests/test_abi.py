"""C-ABI contract checks that need no GPU: the library builds/loads, exports every symbol include/v3d_hip.h declares,
the ctypes mirror of v3d_gemm_args has the compiled size, and the product refuses to run without a GPU."""
import os
import re

import pytest
import torch

from v3d_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "v3d_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(v3d_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(hip.LIB_PATH):
        from v3d_amd.build import build
        build(verbose=False)
    return hip.load_library()


def test_every_declared_symbol_is_exported_and_bound(lib):
    syms = _declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/v3d_hip.h but not exported"
        assert s in hip.SIGNATURES, f"{s} has no ctypes signature in v3d_amd/hip.py"
    assert set(hip.SIGNATURES) == set(syms)


def test_abi_version_and_struct_size(lib):
    import ctypes
    assert lib.v3d_abi_version() == hip.ABI_VERSION
    assert lib.v3d_sizeof_gemm_args() == ctypes.sizeof(hip._GemmArgs)


def test_argument_errors_without_gpu(lib):
    # argument validation happens before any launch: a null args pointer must fail with V3D_ERR_ARG and a message
    assert lib.v3d_gemm(None, None) == -1
    assert b"null" in lib.v3d_last_error()


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful on a machine without a GPU")
def test_product_fails_loudly_without_gpu():
    from v3d_amd import ops
    assert ops._ACTIVE is None
    with pytest.raises(RuntimeError, match="no HIP device|not found"):
        hip.HipOps()
    from tiny import TINY, build_unet, tiny_unet_inputs
    net = build_unet()
    _, _, _, x8, ts, ctx, y = tiny_unet_inputs(TINY["T"], TINY["H"], TINY["W"], TINY["seed"])
    with pytest.raises(RuntimeError):
        net(x8, ts, context=ctx, y=y, num_video_frames=TINY["T"], image_only_indicator=torch.zeros(2, TINY["T"]))
