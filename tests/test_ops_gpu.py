"""-m gpu: every HIP kernel, called through the C ABI, against the torch restatement of the same op (oracle/ops_emul.py)
on identical seeded inputs: exact V3D_512 shapes plus ragged / edge shapes (tests/op_cases.py)."""
import pytest
import torch

import op_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def emu():
    from oracle.ops_emul import EmulOps
    return EmulOps("cuda")


@pytest.mark.parametrize("name,fn,kwargs,tol", op_cases.all_cases(full=True), ids=[c[0] for c in op_cases.all_cases(full=True)])
def test_op(hip_ops, emu, name, fn, kwargs, tol):
    rel, cos, ok = op_cases.run_case(hip_ops, emu, "cuda", name, fn, kwargs, tol)
    assert ok, f"{name}: rel={rel:.3e} (tol {tol}) cos={cos:.6f}"


def test_elementwise(hip_ops, emu):
    bf16_out = {"timestep_embedding", "timestep_embedding_odd", "silu_add", "silu", "pack_input", "pack_input_pad", "nchw_to_nhwc", "copy2d", "gelu",
                "pack_input_im2col", "pack_input_im2col_nocond", "conv_in_as_gemm", "conv_out_as_gemm"}
    for k, (rel, cos) in op_cases.case_elementwise(hip_ops, emu, "cuda").items():
        tol = op_cases.TOL_BF16 if k in bf16_out else 1e-4
        assert rel <= tol, f"{k}: rel={rel:.3e}"


def test_bad_arguments_raise(hip_ops):
    x = torch.zeros(8, 12, dtype=torch.bfloat16, device="cuda")   # K = 12 is not a multiple of 8
    w = torch.zeros(16, 12, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(RuntimeError, match="multiples of 8"):
        hip_ops.linear(x, w)
    with pytest.raises(RuntimeError, match="HIP ops need device memory"):
        hip_ops.linear(torch.zeros(8, 16, dtype=torch.bfloat16), torch.zeros(16, 16, dtype=torch.bfloat16))
