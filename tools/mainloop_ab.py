"""Interleaved same-process A/B of several builds of libv3d_hip.so on the GEMM-family launches that carry the V3D_512 evaluation
(cdna_hip_programming.md rule 24: N variants x M rounds in one process; median and min per variant).  Every build's output is also compared
with the first build's (max abs difference: pure scheduling variants must agree bit for bit).

  python tools/mainloop_ab.py base=v3d_amd/lib/libv3d_hip.so m1=v3d_amd/lib_exp/libv3d_m1.so ... [--only=substr,...] [--rounds=5]
-> gpurun_out/mainloop_ab.json + a table on stdout (us per launch, TF/s of the first build)."""
from __future__ import annotations

import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from v3d_amd.hip import HipOps  # noqa: E402
from v3d_amd.ops import GEMM_CONV3X3, GEMM_CONVT3, GEMM_LINEAR, GemmCall, OpsBase  # noqa: E402

BF, F32 = torch.bfloat16, torch.float32
dev = "cuda"


def build_cases(hip0, only):
    """name -> (make_call(out) -> GemmCall, flop, out_shape).  Shapes / epilogue flags: profiles/r03_op_times.log."""
    cases = []

    def want(name):
        return only is None or any(o in name for o in only)

    def lin(name, M, N, K, bias=True, add=False, res=False, geglu=False, rpg=4096):
        if not want(name):
            return
        A = torch.randn(M, K, device=dev).to(BF)
        W = (torch.randn(1, N, K, device=dev) / K ** 0.5).to(BF)
        n_out = N // 2 if geglu else N
        kw = {}
        if bias:
            kw["bias"] = torch.randn(N, device=dev)
        if add:
            kw.update(add=torch.randn(M // rpg, N, device=dev), add_rpg=rpg, add_ld=N)
        if res:
            kw["res1"] = torch.randn(M, n_out, device=dev).to(BF)
        mk = lambda out: GemmCall(A=A, W=W, out=out, M=M, N=N, K=K, mode=GEMM_LINEAR, geglu=geglu, **kw)
        cases.append((name, mk, 2.0 * M * N * K, (M, n_out)))

    def conv(name, N, C1, C2=0, conv=None, convt=None, res=False, add=False, gn_out=True, gn_in=True):
        if not want(name):
            return
        K = C1 + C2
        if conv is not None:
            n_img, H, W_ = conv
            S, ips, taps = H * W_, 1, 9
            kw = dict(mode=GEMM_CONV3X3, Hin=H, Win=W_, Hout=H, Wout=W_)
        else:
            B, T, S = convt
            n_img, ips, taps = B * T, T, 3
            kw = dict(mode=GEMM_CONVT3, T=T, S=S, tmin=0, tmax=T - 1)
        M = n_img * S
        x1 = torch.randn(M, C1, device=dev).to(BF)
        x2 = torch.randn(M, C2, device=dev).to(BF) if C2 else None
        w = (torch.randn(taps, N, K, device=dev) / (K * taps) ** 0.5).to(BF)
        kw.update(bias=torch.randn(N, device=dev))
        if res:
            kw.update(res1=torch.randn(M, N, device=dev).to(BF))
        if add:
            kw.update(add=torch.randn(n_img, N, device=dev), add_rpg=S, add_ld=N)
        if gn_in:
            ga, be = torch.ones(K, device=dev), torch.zeros(K, device=dev)
            table = hip0.groupnorm_table(x1, x2, ga, be, n_img, S, eps=1e-5, imgs_per_stat=ips)
            kw.update(gn_in=table, gn_in_rps=ips * S, gn_in_silu=True, A2=x2)
        rps = ips * S
        nslots = OpsBase.gn_nslots(rps, ips)

        def mk(out):
            skw = {}
            if gn_out:
                skw = dict(gn_stats=torch.zeros((M // rps, nslots, 32, 2), dtype=F32, device=dev), gn_rps=rps, gn_cpg=N // 32)
            return GemmCall(A=x1, W=w, out=out, M=M, N=N, K=K, **kw, **skw)
        cases.append((name, mk, 2.0 * M * N * K * taps, (M, N)))

    # GroupNorm -> SiLU -> conv3x3 (conv_halo_kernel), ResBlock in / out convolutions of the three upper levels
    conv("c3_L0_320_in", 320, 320, conv=(36, 64, 64), add=True)
    conv("c3_L0_320_out", 320, 320, conv=(36, 64, 64), res=True)
    conv("c3_L0_concat640", 320, 320, 320, conv=(36, 64, 64), add=True)
    conv("c3_L1_640_out", 640, 640, conv=(36, 32, 32), res=True)
    conv("c3_L1_concat1920", 640, 1280, 640, conv=(36, 32, 32), add=True)
    conv("c3_L2_1280_out", 1280, 1280, conv=(36, 16, 16), res=True)
    conv("c3_L2_concat2560", 1280, 1280, 1280, conv=(36, 16, 16), add=True)
    conv("c3_L3_1280_out", 1280, 1280, conv=(36, 8, 8), res=True)                   # W = 8 haloed kernel: 48 tiles, stream-K over 5-6 blocks per tile
    conv("c3_L3_concat2560", 1280, 1280, 1280, conv=(36, 8, 8), add=True)
    conv("ct_L0_320_gnin", 320, 320, convt=(2, 18, 4096), res=True, gn_out=False)
    conv("ct_L0_320_plain", 320, 320, convt=(2, 18, 4096), add=True, gn_out=False, gn_in=False)     # v3 CONVT3
    conv("ct_L1_640_plain", 640, 640, convt=(2, 18, 1024), res=True, gn_out=False, gn_in=False)
    conv("ct_L2_1280_plain", 1280, 1280, convt=(2, 18, 256), res=True, gn_out=False, gn_in=False)
    conv("c3_L1_plain_640", 640, 640, conv=(36, 32, 32), gn_out=False, gn_in=False)                   # v3 CONV3X3 (VAE / strided family)
    # linears (v3 192 x 320 / v2), attention + feed-forward projections
    lin("lin_L0_320_bar", 36 * 4096, 320, 320, add=True, res=True)
    lin("lin_L0_320_b", 36 * 4096, 320, 320)
    lin("lin_L0_320_r", 36 * 4096, 320, 320, res=True)
    lin("lin_L0_skip640_b", 36 * 4096, 320, 640)                               # K = 640 -> 320 (skip connection of the concat ResBlocks)
    lin("lin_L0_skip640_bar", 36 * 4096, 320, 640, add=True, res=True)
    lin("lin_L1_640_bar", 36 * 1024, 640, 640, add=True, res=True, rpg=1024)
    lin("lin_L1_ff2_br", 36 * 1024, 640, 2560, res=True)
    lin("lin_L1_qkv", 36 * 1024, 1920, 640, bias=False)
    lin("lin_L2_1280_bar", 36 * 256, 1280, 1280, add=True, res=True, rpg=256)
    lin("lin_L2_ff2_br", 36 * 256, 1280, 5120, res=True)
    lin("lin_L2_qkv", 36 * 256, 3840, 1280, bias=False)
    lin("lin_L3_1280_b", 36 * 64, 1280, 1280)
    lin("lin_L3_1280_bar", 36 * 64, 1280, 1280, add=True, res=True, rpg=64)
    lin("lin_L3_ff2_br", 36 * 64, 1280, 5120, res=True)
    # one rank of an 8-way frame shard (3 frames x 2 samples = 6 images): BASELINE.json configs[2] / [3]
    lin("shard6_L0_320_bar", 6 * 4096, 320, 320, add=True, res=True)
    lin("shard6_L1_640_bar", 6 * 1024, 640, 640, add=True, res=True, rpg=1024)
    lin("shard6_L1_ff2_br", 6 * 1024, 640, 2560, res=True)
    lin("shard6_L2_1280_bar", 6 * 256, 1280, 1280, add=True, res=True, rpg=256)
    lin("shard6_L2_ff2_br", 6 * 256, 1280, 5120, res=True)
    lin("geglu_L1", 36 * 1024, 5120, 640, geglu=True)
    lin("geglu_L2", 36 * 256, 10240, 1280, geglu=True)
    lin("lin_sq_4096", 4096, 4096, 4096)
    # the plain q | k | v projection shapes of the vendor-library probe (library 1022-1054 TF/s)
    lin("plain_9216_2560_1280", 9216, 2560, 1280, bias=False)
    lin("plain_9216_3840_1280", 9216, 3840, 1280, bias=False)
    lin("plain_36864_1280_640", 36864, 1280, 640, bias=False)
    lin("plain_36864_1280_640_br", 36864, 1280, 640, res=True)
    lin("plain_sq_8192", 8192, 8192, 8192, bias=False)
    return cases


def main():
    libs, only, rounds = [], None, 5
    for a in sys.argv[1:]:
        if a.startswith("--only="):
            only = a.split("=", 1)[1].split(",")
        elif a.startswith("--rounds="):
            rounds = int(a.split("=", 1)[1])
        elif "=" in a:
            tag, path = a.split("=", 1)
            libs.append((tag, os.path.join(ROOT, path) if not os.path.isabs(path) else path))
    assert libs, "usage: mainloop_ab.py tag=path [tag=path ...]"
    hips = [(tag, HipOps(lib_path=path)) for tag, path in libs]
    cases = build_cases(hips[0][1], only)
    rows = []
    print(f"{'case':20s}" + "".join(f"{t:>11s}" for t, _ in hips) + "   (median us; min in the json)   TF/s of the first, max |diff| vs the first")
    for name, mk, flop, oshape in cases:
        outs = [torch.zeros(oshape, dtype=BF, device=dev) for _ in hips]
        calls = [mk(o) for o in outs]
        times = [[] for _ in hips]
        for (tag, h), c in zip(hips, calls):          # warm-up + the outputs that are compared
            h.gemm(c)
            h.gemm(c)
        torch.cuda.synchronize()
        diffs = [float((o.float() - outs[0].float()).abs().max()) for o in outs]
        for r in range(rounds):
            order = list(range(len(hips)))
            if r % 2:
                order.reverse()
            for i in order:
                h, c = hips[i][1], calls[i]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    h.gemm(c)
                e1.record()
                torch.cuda.synchronize()
                times[i].append(e0.elapsed_time(e1) / 4 * 1e3)
        med = [statistics.median(t) for t in times]
        mn = [min(t) for t in times]
        rows.append(dict(case=name, flop=flop, tags=[t for t, _ in hips], median_us=med, min_us=mn, max_abs_diff_vs_first=diffs))
        print(f"{name:20s}" + "".join(f"{m:11.1f}" for m in med) + f"   {flop / med[0] / 1e6:6.0f} TF/s   diff " + " ".join(f"{d:.3g}" for d in diffs), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "mainloop_ab.json"), "w") as f:
        json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
