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
    logs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r01*_bench.log")))
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
        c = d["cpu_baseline"]
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in c, (path, k)
        assert c["kind"] in ("reference", "port") and c["unit"] == d["unit"]


def test_profiles_hold_the_rocprof_summary_bench_refers_to():
    src = open(os.path.join(ROOT, "bench.py")).read()
    import re
    m = re.search(r'"profiles", "(r01\w+_pmc_traffic\.json)"', src)
    assert m, "bench.py no longer reads a committed PMC traffic file"
    assert os.path.isfile(os.path.join(ROOT, "profiles", m.group(1)))
    tag = m.group(1).split("_")[0]
    assert glob.glob(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats*.txt")), "kernel-trace summary of the same round missing"
