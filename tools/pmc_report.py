"""Summarise a rocprofv3 --pmc run (rocpd sqlite) per kernel: mean counter values per dispatch."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [d[1] for d in cur.execute("pragma table_info(pmc_events)")]
rows = cur.execute("select * from pmc_events").fetchall()
ix = {c: i for i, c in enumerate(cols)}
if "--cols" in sys.argv:
    print(cols); print(rows[:3])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r[ix["name"]] if "name" in ix else r[ix["kernel_name"]]
    key = (name[:70], r[ix["grid_size"]] if "grid_size" in ix else 0)
    agg[key][r[ix["counter_name"]] if "counter_name" in ix else r[ix["pmc_name"]]].append(r[ix["value"]] if "value" in ix else r[ix["counter_value"]])
for key, d in agg.items():
    print(key)
    for c, v in sorted(d.items()):
        print(f"    {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
