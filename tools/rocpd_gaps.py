"""Idle time between consecutive kernels of the replayed step, from a rocprofv3 rocpd sqlite database (kernel-trace).
usage: python tools/rocpd_gaps.py results.db [out.md [sequence.txt]]
Steps are delimited by adam_kernel (one launch per step); the last 10 whole steps are analysed."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else cols[0]
rows = db.execute(f"select {name}, start, end from kernels order by start").fetchall()
ends = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
# bench.py ends with three eager, event-timed steps (launch gaps of the host, torch.cat of the inputs): DM_GAPS_SKIP_LAST=n leaves
# the last n steps out, so that the analysed ones are graph replays
import os
skip = int(os.environ.get("DM_GAPS_SKIP_LAST", "4"))
ends = ends[:len(ends) - skip] if skip > 0 else ends
ends = ends[-11:]
out = []
tot_busy = tot_gap = tot_win = 0
by_prev = collections.defaultdict(lambda: [0, 0])
hist = collections.Counter()
nk = 0
for a, b in zip(ends[:-1], ends[1:]):
    seg = rows[a + 1:b + 1]
    nk += len(seg)
    win = seg[-1][2] - rows[a][2]
    busy = sum(e - s for _, s, e in seg)
    tot_busy += busy; tot_win += win
    prev_end, prev_name = rows[a][2], rows[a][0]
    for n, s, e in seg:
        g = s - prev_end
        tot_gap += max(g, 0)
        by_prev[prev_name[:60]][0] += 1; by_prev[prev_name[:60]][1] += g
        hist[min(int(max(g, 0) / 1000), 10)] += 1
        prev_end, prev_name = max(prev_end, e), n
n = len(ends) - 1
out.append(f"steps analysed: {n}; kernels per step {nk / n:.0f}")
out.append(f"step window {tot_win / n / 1e6:.3f} ms; kernel time {tot_busy / n / 1e6:.3f} ms; idle between kernels {tot_gap / n / 1e6:.3f} ms")
out.append("gap histogram (us bucket: launches per step): " + ", ".join(f"{k}{'+' if k == 10 else ''}: {v / n:.1f}" for k, v in sorted(hist.items())))
out.append("| predecessor kernel | gaps/step | avg gap us | idle ms/step |"); out.append("|---|---|---|---|")
for k, (c, g) in sorted(by_prev.items(), key=lambda kv: -kv[1][1])[:25]:
    out.append(f"| `{k}` | {c / n:.1f} | {g / c / 1e3:.2f} | {g / n / 1e6:.3f} |")
if len(sys.argv) > 3:       # one step's launch sequence: gap before, duration, name
    a, b = ends[-2], ends[-1]
    prev_end = rows[a][2]
    with open(sys.argv[3], "w") as f:
        for nme, st, en in rows[a + 1:b + 1]:
            f.write(f"{(st - prev_end) / 1e3:8.2f} {(en - st) / 1e3:8.2f}  {nme[:110]}\n")
            prev_end = max(prev_end, en)
txt = "\n".join(out)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
