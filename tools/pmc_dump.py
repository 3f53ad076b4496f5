"""Per-kernel averages of the counters in a rocprofv3 --pmc output directory (sqlite .db files).
usage: python tools/pmc_dump.py DIR [name-substring]"""
import glob, os, sqlite3, sys
from collections import defaultdict
d = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
for path in sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    view = "pmc_events" if "pmc_events" in tabs else next(t for t in tabs if "pmc_event" in t)
    tot, cnt = defaultdict(float), defaultdict(int)
    for name, cname, val in db.execute(f"select name, counter_name, counter_value from {view}"):
        if pat in name:
            tot[(name[:70], cname)] += float(val); cnt[(name[:70], cname)] += 1
    for (k, c) in sorted(tot):
        print(f"{k:70s} {c:28s} avg {tot[(k, c)] / cnt[(k, c)]:16.1f}  n={cnt[(k, c)]}")
