"""MFMA-pipe utilisation per kernel family from one rocprofv3 --kernel-trace --pmc pass over tools/pmc_eval.py
(counters SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE).   usage: pmc_mfma.py PASS.db
busy fraction of a family = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (sum(GRBM_GUI_ACTIVE) / XCDS x SIMDS), SIMDS = 4 x 256, XCDS = 8: rocprofv3 sums
GRBM_GUI_ACTIVE (a dispatch's active shader-clock count) over the 8 XCDs (checked against kernel durations: 6.76e6 per 400-us ff_fused launch
= 8 x 2.1 GHz).  The ratio is relative to the clock the box actually ran at (~2.1 GHz under this load, 2.4 nominal).  The raw sums are printed too:
the aggregation of a counter over XCDs differs between rocprofv3 builds, so the ff_fused row (whose MFMA count is known: 6 M C hidden /
32768 flop per 32x32x16 instruction x 32 cycles) is used as the calibration of the MFMA counter."""
import collections, sqlite3, sys

db = sqlite3.connect(sys.argv[1])
cols = [d[1] for d in db.execute("pragma table_info(pmc_events)")]
ix = {c: i for i, c in enumerate(cols)}
name_c = "name" if "name" in ix else "kernel_name"
cn_c = "counter_name" if "counter_name" in ix else "pmc_name"
val_c = "value" if "value" in ix else "counter_value"
disp_c = "dispatch_id" if "dispatch_id" in ix else None


def family(name):
    for key, fam in (("ff_fused_kernel", "ff_fused"), ("ln_proj_kernel", "ln_proj"), ("conv_halo_kernel<0", "conv_halo_3x3"), ("conv_halo_kernel<1", "conv_halo_temporal"), ("conv_halo_kernel", "conv_halo"), ("gemm_kernel_v3<192, 320", "gemm_v3_192x320"), ("gemm_kernel_v3<256, 128", "gemm_v3_256x128_geglu"),
                     ("gemm_kernel_v3", "gemm_v3_256x256"), ("gemm_kernel_v2", "gemm_v2"), ("gemm_kernel_v1", "gemm_v1"), ("attn_spatial", "attn_spatial"),
                     ("attn_vae", "attn_vae")):
        if key in name:
            return fam
    return None


agg = collections.defaultdict(lambda: collections.defaultdict(float))
ndisp = collections.defaultdict(set)
for r in db.execute("select * from pmc_events"):
    f = family(r[ix[name_c]])
    if f is None:
        continue
    agg[f][r[ix[cn_c]]] += r[ix[val_c]]
    if disp_c:
        ndisp[f].add(r[ix[disp_c]])
SIMDS, XCDS = 4 * 256, 8
print(f"{'family':24s} {'dispatches':>10s} {'MFMA_BUSY':>14s} {'SQ_BUSY':>14s} {'GUI_ACTIVE':>14s} {'MFMA busy':>10s}")
for f, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0)):
    mf, sq, gui = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("SQ_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
    frac = mf / (gui / XCDS * SIMDS) if gui else float("nan")
    print(f"{f:24s} {len(ndisp[f]):10d} {mf:14.4g} {sq:14.4g} {gui:14.4g} {frac:10.3f}")
# calibration: 15 ff_fused launches per evaluation, 2 evaluations, M = 147456, C = 320, hidden = 1280
n_ff = len(ndisp.get("ff_fused", ()))
if n_ff:
    expect = n_ff * 6.0 * 147456 * 320 * 1280 / 32768 * 32
    print(f"calibration: ff_fused MFMA_BUSY expected {expect:.4g} (known instruction count x 32 cycles), measured {agg['ff_fused'].get('SQ_VALU_MFMA_BUSY_CYCLES', 0):.4g}"
          f" -> counter factor {agg['ff_fused'].get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / expect:.3f}")
