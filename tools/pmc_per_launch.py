"""Per-launch HBM-side traffic of the GEMM family: joins the per-dispatch FETCH_SIZE / WRITE_SIZE counters of the two rocprofv3 --pmc passes
over tools/pmc_eval.py with the v3d_gemm / v3d_ff_fused call list that run recorded (gpurun_out/pmc_eval_calls.json: shapes and algorithmic
bytes, in launch order).  Prints the shape classes ranked by measured bytes with their measured / algorithmic ratio - which launches carry
the re-read traffic.  usage: pmc_per_launch.py FETCH.db WRITE.db CALLS.json [FETCH_FACTOR WRITE_FACTOR]"""
import collections
import json
import sqlite3
import sys


def per_dispatch(path, counter):
    db = sqlite3.connect(path)
    cols = [d[1] for d in db.execute("pragma table_info(pmc_events)")]
    ix = {c: i for i, c in enumerate(cols)}
    name_c = "name" if "name" in ix else "kernel_name"
    cn_c = "counter_name" if "counter_name" in ix else "pmc_name"
    val_c = "value" if "value" in ix else "counter_value"
    disp_c = "dispatch_id"
    agg = collections.OrderedDict()
    for r in db.execute(f"select * from pmc_events order by {disp_c}"):
        if r[ix[cn_c]] != counter:
            continue
        k = r[ix[disp_c]]
        if k not in agg:
            agg[k] = [r[ix[name_c]], 0.0]
        agg[k][1] += r[ix[val_c]]
    return [(k, v[0], v[1]) for k, v in agg.items()]


def main():
    fetch = per_dispatch(sys.argv[1], "FETCH_SIZE")
    write = per_dispatch(sys.argv[2], "WRITE_SIZE")
    calls = json.load(open(sys.argv[3]))["calls"]
    ff = float(sys.argv[4]) if len(sys.argv) > 4 else 2.0
    fw = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
    is_main = lambda n: ("gemm_kernel_v" in n) or ("ff_fused_kernel" in n) or ("conv_halo_kernel" in n)
    fg = [d for d in fetch if is_main(d[1])]
    wg = [d for d in write if is_main(d[1])]
    print(f"{len(calls)} recorded calls, {len(fg)} / {len(wg)} GEMM-family dispatches in the FETCH / WRITE passes")
    if len(fg) != len(calls) or len(wg) != len(calls):
        print("WARNING: dispatch count differs from the call list (warm-up launches?): aligning from the END of both lists")
    n = min(len(calls), len(fg), len(wg))
    calls, fg, wg = calls[-n:], fg[-n:], wg[-n:]
    cls = collections.OrderedDict()
    for c, f, w in zip(calls, fg, wg):
        kern = f[1].split("(")[0].replace("void (anonymous namespace)::", "")[:58]
        key = (c["kind"], c["mode"], c["M"], c["N"], c["K"], c["batch"], c["geglu"], c["res"], kern)
        a = cls.setdefault(key, [0, 0.0, 0.0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += f[2] * ff * 1024
        a[2] += w[2] * fw * 1024
        a[3] += c["alg_read"]
        a[4] += c["alg_write"]
        a[5] += c["flop"]
    rows = sorted(cls.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))
    tot_m = sum(v[1] + v[2] for v in cls.values())
    tot_a = sum(v[3] + v[4] for v in cls.values())
    print(f"GEMM family total: measured {tot_m / 1e9:.1f} GB vs algorithmic {tot_a / 1e9:.1f} GB = {tot_m / tot_a:.2f}x (all recorded evaluations)")
    print(f"{'launches':>8s} {'meas MB/launch':>14s} {'alg MB/launch':>13s} {'ratio':>6s} {'read x':>7s} {'write x':>7s} {'flop/B alg':>10s}  shape / kernel")
    for key, v in rows[:28]:
        kind, mode, M, N, K, b, geglu, res, kern = key
        n_ = v[0]
        meas, alg = (v[1] + v[2]) / n_, (v[3] + v[4]) / n_
        print(f"{n_:8d} {meas / 1e6:14.1f} {alg / 1e6:13.1f} {meas / alg:6.2f} {v[1] / max(v[3], 1):7.2f} {v[2] / max(v[4], 1):7.2f} {v[5] / (v[3] + v[4]):10.0f}  "
              f"{kind} m{mode} M={M} N={N} K={K} b={b}{' geglu' if geglu else ''} res={res} | {kern}")


if __name__ == "__main__":
    main()
