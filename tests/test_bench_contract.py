"""The bench.py output contract (driver prompt, Measurement section), checked on the committed GPU run logs under profiles/ - a
CPU-side guard that the JSON line keeps every field the driver and the judge read."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json_line(path):
    lines = [ln for ln in open(path).read().splitlines() if ln.startswith("{")]
    assert lines, f"no JSON line in {path}"
    return json.loads(lines[-1])


def test_committed_bench_lines_keep_the_contract():
    import re
    logs = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r0*_bench.log")) if re.fullmatch(r"r0\d[a-z]?_bench\.log", os.path.basename(f)))
    assert logs, "no committed bench log"
    for path in logs[-2:]:
        d = _last_json_line(path)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                  "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert k in d, (path, k)
        assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["n_gpus"] == 1
        assert d["vs_baseline"] is None                      # BASELINE.md has no published number for this metric
        assert "workload" in d["config"] and "model" not in d["config"]
        # value = whole-job frames / wall time
        assert abs(d["value"] - d["config"]["frames"] * 1e3 / d["ms_per_step"]) / d["value"] < 1e-3
        r = d["roofline"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in r, (path, k)
        assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
        if os.path.basename(path) >= "r02":
            assert r["per_kernel"] and all(k["bound"] in ("hbm", "mfma") and 0 < k["frac"] <= 1.0 for k in r["per_kernel"]), path
        c = d["cpu_baseline"]
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in c, (path, k)
        assert c["kind"] in ("reference", "port") and c["unit"] == d["unit"]


def test_pmc_profile_bench_reads_is_committed_and_not_older_than_the_kernels():
    """roofline.traffic comes from the rocprofv3 PMC passes committed under profiles/ (counters cannot be collected inside bench.py).
    The profile records the content hash of v3d_amd/csrc it was taken on: a kernel change after the last profile run fails here until
    tools/profile.sh has been re-run on the GPU and its summaries copied to profiles/."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    path = os.path.join(ROOT, "profiles", bench.PMC_PROFILE)
    assert os.path.isfile(path), f"{path} missing: run tools/profile.sh on the GPU box and commit its summaries"
    prof = json.load(open(path))
    assert prof.get("csrc_sha256_16") == bench.csrc_digest(), \
        f"profiles/{bench.PMC_PROFILE} was taken on kernel sources {prof.get('csrc_sha256_16')}, the tree is at {bench.csrc_digest()}: re-profile"
    tag = bench.PMC_PROFILE.split("_")[0]
    assert glob.glob(os.path.join(ROOT, "profiles", f"{tag}*_kernel_stats*.txt")), "kernel-trace summary of the same round missing"
