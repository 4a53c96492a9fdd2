"""Build libv3d_hip.so (gfx950) from v3d_amd/csrc/*.hip with hipcc; in-tree, no JIT cache.

`python -m v3d_amd.build` or `v3d_amd.build.build()`.  hipcc cross-compiles without a GPU present.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libv3d_hip.so")
ARCH = "gfx950"
# (-fno-slp-vectorize was tried against the v_pk_*_f32 forms the SLP vectoriser puts into the GEGLU epilogue: no change,
# 537 vs 545 TF/s on the 64x64 feed-forward projection, 10.55 vs 10.67 frames/s)
FLAGS = ["-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-ffast-math", "-fno-finite-math-only",
         "-Wall", "-Wno-unused-function"]
# ff.hip lays its instruction stream out by hand in fenced issue slots; the SLP vectoriser would pair up GELU pieces that belong to
# different slots into v_pk_*_f32 (which also cost more than they save beside MFMAs)
# attn.hip: the compiler SLP-packs the softmax's adjacent fp32 multiplies / adds into v_pk_*_f32, which beside MFMAs cost more than the scalar
# pairs (same-process A/B, profiles/r05_attn_ab.txt: 917 -> 914 us alone, 895 -> 874 us together with the deferred maximum)
FILE_FLAGS = {"ff.hip": ["-fno-slp-vectorize"], "attn.hip": ["-fno-slp-vectorize"]}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 to build the gfx950 kernels)")


LAB = os.path.join(os.path.dirname(HERE), "tools", "lab")     # lab-only kernels (round-4 gemm4.hip): linked into the experiments library only


def sources(experiments: bool = False):
    src = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    if experiments and os.path.isdir(LAB):
        src += sorted(os.path.join(LAB, f) for f in os.listdir(LAB) if f.endswith(".hip"))
    return src


def _stale(out: str, deps) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, experiments: bool = False) -> str:
    """experiments=True: a second library (lib_exp/libv3d_hip_exp.so, -DV3D_EXPERIMENTS) whose GEMM / convolution kernels honour the
    V3D_GEMM_ABLATE timing switches (skip MFMAs / DMA / fragment reads / barriers: results are garbage by design).  Select it with
    V3D_HIP_LIB=<path>; the product library never contains those switches."""
    if experiments:
        return _build(force, verbose, os.path.join(HERE, "lib_exp"), "libv3d_hip_exp.so", ["-DV3D_EXPERIMENTS"])
    return _build(force, verbose, LIBDIR, "libv3d_hip.so", [])


def _build(force, verbose, LIBDIR, libname, extra) -> str:
    LIB = os.path.join(LIBDIR, libname)
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "v3d_hip.h"))
    objs, jobs = [], []
    for src in sources(experiments=bool(extra)):
        obj = os.path.join(LIBDIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc, *FLAGS, *extra, *FILE_FLAGS.get(os.path.basename(src), []), "-I", CSRC, "-c", src, "-o", obj]
        if verbose:
            print("[v3d_amd.build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    if jobs or force or _stale(LIB, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB]
        if verbose:
            print("[v3d_amd.build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if not extra:
        # libv3d_comm.so is OPTIONAL: the Python product path never loads it (dist.py runs the same schedule on torch.distributed), and a box
        # without rccl.h / librccl under the ROCm tree must still get the kernel library
        try:
            build_comm(force=force, verbose=verbose)
        except Exception as e:      # noqa: BLE001
            print(f"[v3d_amd.build] WARNING: libv3d_comm.so not built ({str(e).splitlines()[0]}); the kernel library is unaffected", file=sys.stderr)
    return LIB


COMM_SRC = os.path.join(HERE, "csrc_comm", "comm.hip")
COMM_LIB = os.path.join(LIBDIR, "libv3d_comm.so")


def build_comm(force: bool = False, verbose: bool = True) -> str:
    """libv3d_comm.so (include/v3d_comm.h): the RCCL wrappers of the frame-axis exchanges for hosts without torch.distributed - its own
    library (links librccl; the kernel library does not)."""
    hdr = os.path.join(os.path.dirname(HERE), "include", "v3d_comm.h")
    if force or _stale(COMM_LIB, [COMM_SRC, hdr]):
        rocm_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(_hipcc()))), "lib")
        # rpath: a host that has not loaded torch (whose bundled librccl would otherwise satisfy the dependency) resolves librccl from the ROCm tree
        cmd = [_hipcc(), "-O2", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-shared", COMM_SRC, "-o", COMM_LIB, f"-L{rocm_lib}", "-lrccl",
               f"-Wl,-rpath,{rocm_lib}"]
        if verbose:
            print("[v3d_amd.build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {COMM_SRC}:\n{r.stdout}\n{r.stderr}")
    return COMM_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, experiments="--experiments" in sys.argv))
