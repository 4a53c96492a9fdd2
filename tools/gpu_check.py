"""Run every per-op parity case (tests/op_cases.py) on the GPU without stopping at the first failure, then a
micro-benchmark of the hot kernels at the exact V3D_512 shapes.  Writes gpurun_out/op_check.json and
gpurun_out/op_bench.json.  Usage: python tools/gpu_check.py [--no-bench] [--no-check]
"""
from __future__ import annotations

import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

from v3d_amd.ops import GEMM_CONV3X3, GEMM_CONVT3, GEMM_LINEAR, GemmCall  # noqa: E402

BF, F32 = torch.bfloat16, torch.float32


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters  # ms


def bench(hip):
    dev = "cuda"
    out = []

    only = [a.split("=", 1)[1].split(",") for a in sys.argv if a.startswith("--only=")]
    only = only[0] if only else None

    def want(name):
        return only is None or any(o in name for o in only)

    def gemm_case(name, M, N, K, mode=GEMM_LINEAR, geglu=False, conv=None, convt=None, res=False):
        if not want(name):
            return
        kw = {}
        taps = 1
        a_rows = M
        if mode == GEMM_CONV3X3:
            n_img, Hin, Win, stride, up = conv
            Hout, Wout = (Hin * up - 1) // stride + 1, (Win * up - 1) // stride + 1
            M = n_img * Hout * Wout
            a_rows = n_img * Hin * Win
            taps = 9
            kw.update(Hin=Hin, Win=Win, Hout=Hout, Wout=Wout, stride=stride, up=up)
        elif mode == GEMM_CONVT3:
            B, T, S = convt
            M = a_rows = B * T * S
            taps = 3
            kw.update(T=T, S=S, tmin=0, tmax=T - 1)
        A = torch.randn(a_rows, K, device=dev).to(BF)
        W = (torch.randn(taps, N, K, device=dev) / (K * taps) ** 0.5).to(BF)
        n_out = N // 2 if geglu else N
        o = torch.empty(M, n_out, dtype=BF, device=dev)
        bias = torch.randn(N, device=dev)
        if res:
            kw.update(res1=torch.randn(M, n_out, device=dev).to(BF))
        call = GemmCall(A=A, W=W, out=o, M=M, N=N, K=K, bias=bias, mode=mode, geglu=geglu, **kw)
        ms = timeit(lambda: hip.gemm(call))
        flop = 2.0 * M * N * K * taps
        out.append(dict(name=name, ms=ms, tflops=flop / ms / 1e9, M=M, N=N, K=K, taps=taps))
        print(f"{name:34s} M={M:7d} N={N:5d} K={K:5d} taps={taps}  {ms:8.3f} ms  {flop / ms / 1e9:8.1f} TF/s", flush=True)

    # [V3D] shapes, batch 36
    gemm_case("lin_L0_320x320", 36 * 4096, 320, 320)
    gemm_case("lin_L0_qk_640", 36 * 4096, 640, 320)
    gemm_case("lin_L0_ff1_geglu", 36 * 4096, 2560, 320, geglu=True)
    gemm_case("lin_L0_ff2", 36 * 4096, 320, 1280, res=True)
    gemm_case("lin_L1_640x640", 36 * 1024, 640, 640)
    gemm_case("lin_L1_ff1_geglu", 36 * 1024, 5120, 640, geglu=True)
    gemm_case("lin_L1_ff2", 36 * 1024, 640, 2560, res=True)
    gemm_case("lin_L2_1280x1280", 36 * 256, 1280, 1280)
    gemm_case("lin_L2_ff1_geglu", 36 * 256, 10240, 1280, geglu=True)
    gemm_case("lin_L2_ff2", 36 * 256, 1280, 5120, res=True)
    gemm_case("lin_L2_tqkv_3840", 36 * 256, 3840, 1280)
    gemm_case("lin_L1_tqkv_1920", 36 * 1024, 1920, 640)
    gemm_case("lin_sq_4096", 4096, 4096, 4096)
    gemm_case("conv_L0_640to320", 0, 320, 640, GEMM_CONV3X3, conv=(36, 64, 64, 1, 1))
    gemm_case("conv_L1_1920to640", 0, 640, 1920, GEMM_CONV3X3, conv=(36, 32, 32, 1, 1))
    gemm_case("conv_L1_1280sq", 0, 1280, 1280, GEMM_CONV3X3, conv=(36, 32, 32, 1, 1))
    gemm_case("conv_L2_2560to1280", 0, 1280, 2560, GEMM_CONV3X3, conv=(36, 16, 16, 1, 1))
    gemm_case("conv_L0_320", 0, 320, 320, GEMM_CONV3X3, conv=(36, 64, 64, 1, 1))
    gemm_case("conv_L0_960to320", 0, 320, 960, GEMM_CONV3X3, conv=(36, 64, 64, 1, 1))
    gemm_case("conv_L1_640", 0, 640, 640, GEMM_CONV3X3, conv=(36, 32, 32, 1, 1))
    gemm_case("conv_L2_1280", 0, 1280, 1280, GEMM_CONV3X3, conv=(36, 16, 16, 1, 1))
    gemm_case("conv_L3_1280", 0, 1280, 1280, GEMM_CONV3X3, conv=(36, 8, 8, 1, 1))
    gemm_case("conv_L3_2560to1280", 0, 1280, 2560, GEMM_CONV3X3, conv=(36, 8, 8, 1, 1))
    gemm_case("convt_L0_320", 0, 320, 320, GEMM_CONVT3, convt=(2, 18, 4096))
    gemm_case("convt_L2_1280", 0, 1280, 1280, GEMM_CONVT3, convt=(2, 18, 256))
    gemm_case("vae_conv_512sq_128", 0, 128, 128, GEMM_CONV3X3, conv=(18, 512, 512, 1, 1))
    gemm_case("vae_conv_256sq_256", 0, 256, 256, GEMM_CONV3X3, conv=(18, 256, 256, 1, 1))
    gemm_case("vae_conv_64sq_512", 0, 512, 512, GEMM_CONV3X3, conv=(18, 64, 64, 1, 1))

    def attn_case(name, n_img, S, heads):
        if not want(name):
            return
        C = heads * 64
        qk = torch.randn(n_img * S, 2 * C, device=dev).to(BF)
        vT = torch.randn(n_img, C, S, device=dev).to(BF)
        o = torch.empty(n_img * S, C, dtype=BF, device=dev)
        ms = timeit(lambda: hip.attn_spatial(qk[:, :C], qk[:, C:], vT, o, n_img, S, heads, 0.125))
        flop = 4.0 * n_img * heads * S * S * 64
        out.append(dict(name=name, ms=ms, tflops=flop / ms / 1e9))
        print(f"{name:34s} n={n_img} S={S} h={heads}  {ms:8.3f} ms  {flop / ms / 1e9:8.1f} TF/s", flush=True)

    attn_case("attn_L0", 36, 4096, 5)
    attn_case("attn_L1", 36, 1024, 10)
    attn_case("attn_L2", 36, 256, 20)
    attn_case("attn_L3", 36, 64, 20)

    def tattn_case(name, B, T, S, heads):
        if not want(name):
            return
        C = heads * 64
        qkv = torch.randn(B, T, S, 3 * C, device=dev).to(BF)
        o = torch.empty(B, T, S, C, dtype=BF, device=dev)
        ms = timeit(lambda: hip.attn_temporal(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], o, heads, 0.125))
        byt = qkv.numel() * 2 + o.numel() * 2
        out.append(dict(name=name, ms=ms, gbps=byt / ms / 1e6))
        print(f"{name:34s} B={B} T={T} S={S} h={heads}  {ms:8.3f} ms  {byt / ms / 1e6:8.1f} GB/s", flush=True)

    tattn_case("tattn_L0", 2, 18, 4096, 5)
    tattn_case("tattn_L1", 2, 18, 1024, 10)
    tattn_case("tattn_L2", 2, 18, 256, 20)

    def gn_case(name, n_img, S, C, ips=1):
        if not want(name):
            return
        x = torch.randn(n_img * S, C, device=dev).to(BF)
        ga, be = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        ms = timeit(lambda: hip.groupnorm(x, None, ga, be, n_img, S, eps=1e-5, silu=True, imgs_per_stat=ips))
        byt = x.numel() * 2 * 3
        out.append(dict(name=name, ms=ms, gbps=byt / ms / 1e6))
        print(f"{name:34s} n={n_img} S={S} C={C}  {ms:8.3f} ms  {byt / ms / 1e6:8.1f} GB/s (stats+apply, 3 passes)", flush=True)

    gn_case("gn_L0_320", 36, 4096, 320)
    gn_case("gn_L0_960", 36, 4096, 960)
    gn_case("gn3d_L0_320", 36, 4096, 320, 18)
    gn_case("gn_L2_1280", 36, 256, 1280)
    gn_case("gn_vae_512sq_128", 18, 512 * 512, 128)

    def ln_case(name, M, C):
        if not want(name):
            return
        x = torch.randn(M, C, device=dev).to(BF)
        ga, be = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        o = torch.empty_like(x)
        ms = timeit(lambda: hip.layernorm(x, ga, be, o, 1e-5))
        byt = x.numel() * 2 * 2
        out.append(dict(name=name, ms=ms, gbps=byt / ms / 1e6))
        print(f"{name:34s} M={M} C={C}  {ms:8.3f} ms  {byt / ms / 1e6:8.1f} GB/s", flush=True)

    def lnproj_case(name, M, N, n_rm, S):
        if not want(name):
            return
        from v3d_amd.engine.packing import ln_proj_pack
        C = 320
        x = torch.randn(M, C, device=dev).to(BF)
        w = (torch.randn(N, C, device=dev) / C ** 0.5).to(BF)
        wp, pbias = ln_proj_pack(w, torch.ones(C, device=dev), torch.zeros(C, device=dev))
        ga, be = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        ms = timeit(lambda: hip.ln_proj(x, 1e-5, wp, pbias, n_rm, S))
        n1 = torch.empty_like(x)
        o2 = torch.empty(M, N, dtype=BF, device=dev)

        def unfused():
            hip.layernorm(x, ga, be, n1, 1e-5)
            hip.gemm(GemmCall(A=n1, W=w, out=o2, M=M, N=N, K=C))
        ms2 = timeit(unfused)
        out.append(dict(name=name, ms=ms, unfused_ms=ms2, tflops=2.0 * M * N * C / ms / 1e9))
        print(f"{name:34s} M={M} N={N} n_rm={n_rm}  {ms * 1e3:8.1f} us fused   vs {ms2 * 1e3:8.1f} us layernorm + one v3d_gemm (N = {N})", flush=True)

    lnproj_case("lnproj_L0_spatial_qkv", 36 * 4096, 960, 640, 4096)
    lnproj_case("lnproj_L0_temporal_qkv", 36 * 4096, 960, 960, 4096)

    def vattn_case(name, n_img, S, C=512):
        if not want(name):
            return
        q = torch.randn(n_img * S, C, device=dev).to(BF)
        k = torch.randn(n_img * S, C, device=dev).to(BF)
        vT = torch.randn(n_img, C, S, device=dev).to(BF)
        o = torch.empty(n_img * S, C, dtype=BF, device=dev)
        ms = timeit(lambda: hip.attn_vae(q, k, vT, None, o, n_img, S, C, C ** -0.5))
        flop = 4.0 * n_img * S * S * C
        out.append(dict(name=name, ms=ms, tflops=flop / ms / 1e9))
        print(f"{name:34s} n={n_img} S={S} C={C}  {ms:8.3f} ms  {flop / ms / 1e9:8.1f} TF/s (algorithmic 4 S^2 C)", flush=True)

    vattn_case("attn_vae_18x4096", 18, 4096)
    vattn_case("attn_vae_scene_24x9216", 24, 9216)

    ln_case("ln_L0", 36 * 4096, 320)
    ln_case("ln_L1", 36 * 1024, 640)
    ln_case("ln_L2", 36 * 256, 1280)
    return out


def main():
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    from v3d_amd.hip import HipOps
    from oracle.ops_emul import EmulOps
    import op_cases

    hip = HipOps()
    emu = EmulOps("cuda")
    print(f"device: CUs={hip.cu_count} wave={hip.wave_size} arch=gfx{hip.arch} name={torch.cuda.get_device_name(0)}", flush=True)
    results = []
    if "--no-check" not in sys.argv:
        nfail = 0
        only = [a.split("=", 1)[1].split(",") for a in sys.argv if a.startswith("--only=")]
        only = only[0] if only else None
        for name, fn, kw, tol in op_cases.all_cases(full=True):
            if only is not None and not any(o in name for o in only):
                continue
            t0 = time.time()
            try:
                if fn is op_cases.case_elementwise:
                    raise RuntimeError("unexpected")
                rel, cos, ok = op_cases.run_case(hip, emu, "cuda", name, fn, kw, tol)
                err = ""
            except Exception as e:  # keep going: one broken kernel must not hide the others
                rel, cos, ok, err = float("nan"), float("nan"), False, f"{type(e).__name__}: {e}"
                traceback.print_exc()
            torch.cuda.synchronize()
            nfail += (not ok)
            results.append(dict(name=name, rel=rel, cos=cos, ok=ok, err=err))
            print(f"{'PASS' if ok else 'FAIL'} {name:34s} rel={rel:.3e} cos={cos:.6f} {time.time() - t0:.2f}s {err}", flush=True)
        try:
            for k, (rel, cos) in (op_cases.case_elementwise(hip, emu, "cuda").items() if only is None else ()):
                tol = op_cases.TOL_BF16 if k in ("timestep_embedding", "timestep_embedding_odd", "silu_add", "silu", "pack_input",
                                                 "pack_input_pad", "nchw_to_nhwc", "copy2d") else 1e-4
                ok = rel <= tol
                nfail += (not ok)
                results.append(dict(name="elem_" + k, rel=rel, cos=cos, ok=ok, err=""))
                print(f"{'PASS' if ok else 'FAIL'} elem_{k:29s} rel={rel:.3e} cos={cos:.6f}", flush=True)
        except Exception as e:
            traceback.print_exc()
            nfail += 1
            results.append(dict(name="elementwise", rel=float("nan"), cos=float("nan"), ok=False, err=str(e)))
        print(f"op parity: {len(results) - nfail}/{len(results)} passed", flush=True)
        with open(os.path.join(ROOT, "gpurun_out", "op_check.json"), "w") as f:
            json.dump(results, f, indent=1)
    if "--no-bench" not in sys.argv:
        try:
            b = bench(hip)
            with open(os.path.join(ROOT, "gpurun_out", "op_bench.json"), "w") as f:
                json.dump(b, f, indent=1)
        except Exception:
            traceback.print_exc()


if __name__ == "__main__":
    main()
