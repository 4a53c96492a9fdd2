"""Per-kernel summary of a rocprofv3 --kernel-trace rocpd database.  usage: prof_summary.py DB [OUT.txt] [title]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
lines = [f"{r[2]/1e6:9.2f} ms {r[2]/tot*100:5.1f}% calls={r[1]:6d} avg={r[3]/1e3:9.1f}us min={r[4]/1e3:8.1f} max={r[5]/1e3:9.1f}  {r[0][:100]}" for r in rows[:26]]
lines.append(f"total kernel time {tot/1e6:.1f} ms over {sum(r[1] for r in rows)} dispatches, {len(rows)} distinct kernels")
print("\n".join(lines))
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write((sys.argv[3] if len(sys.argv) > 3 else "") + "\n" + "\n".join(lines) + "\n")
