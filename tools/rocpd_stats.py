"""Kernel summary (launches/step, ms/step, avg us) from a rocprofv3 rocpd sqlite database.
usage: python tools/rocpd_stats.py results.db STEPS_INCL_WARMUP [out.md]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); steps = int(sys.argv[2])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else cols[0]
rows = db.execute(f"select {name}, count(*), sum(end-start) from kernels group by {name} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
lines = ["| kernel | launches/step | ms/step | avg us |", "|---|---|---|---|"]
for n, c, ns in rows[:40]:
    lines.append(f"| `{n[:70]}` | {c/steps:.1f} | {ns/steps/1e6:.3f} | {ns/c/1e3:.1f} |")
lines.append(f"\nGPU-busy total: {tot/steps/1e6:.3f} ms/step")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(out + "\n")
