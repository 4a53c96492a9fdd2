"""Every v3d_gemm main loop against the oracle emulator on the same cases: the default heuristic only sends launches whose
tiles fill the CUs to the persistent v3 kernels, so the (smaller) parity shapes are re-run with each implementation forced
(V3D_GEMM_IMPL is read once per process -> one subprocess per implementation)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("impl,extra", [("1", {}), ("2", {}), ("3", {}), ("2", {"V3D_GEMM_SPLITK": "3"}), ("3", {"V3D_GEMM_V3S": "0"}), ("", {"V3D_GEMM_V6": "2"})])
def test_gemm_parity_with_forced_impl(impl, extra):
    """("", V3D_GEMM_V6=2): every linear launch the two-blocks-per-CU kernel can take (M % 192 == 0, N % 160 == 0, >= 2 tiles per CU) goes to it"""
    env = dict(os.environ, **({"V3D_GEMM_IMPL": impl} if impl else {}), **extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gemm_sweep.py"), "--check", "--only=__none__"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "gemm parity failures: 0" in r.stdout, r.stdout + r.stderr
    if impl == "3":      # the gn_epi_* cases must have gone through the v3 <GN> epilogue (not the stand-alone fallback inside v3d_gemm)
        import re
        m = re.search(r"gn epilogue launches: (\d+)", r.stdout)
        assert m and int(m.group(1)) >= 5, r.stdout


@pytest.mark.gpu
def test_counted_waits_hold_under_memory_contention():
    """The hand-managed epilogue (gemm_common.h e4_*) paces residual loads and row stores with COUNTED s_waitcnt and, since round 6, reads its staging rows with
    inline asm so that the compiler adds no vmcnt(0) of its own: a wrong count now shows as a rare wrong row.  tools/e4_stress.py launches the persistent
    LINEAR / two-blocks-per-CU / LDS-haloed kernels 60 times each into a poisoned output, every other launch against a 0.5-GB copy on a second stream, and
    compares every result bit for bit with the first."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "e4_stress.py"), "60"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("0 differ from the first") >= 10, r.stdout + r.stderr
