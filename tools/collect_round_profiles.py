"""Copy what tools/profile_round_full.sh left under gpurun_out/<tag>_prof into profiles/ under the names the documents cite, rebuild the derived
tables (matrix-pipe utilisation, the bf16x3 step's per-kernel table) and check the source stamp.   python tools/collect_round_profiles.py [r04]"""
import collections, json, os, shutil, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
src, dst = os.path.join(ROOT, "gpurun_out", f"{tag}_prof"), os.path.join(ROOT, "profiles")
pairs = {"kernel_stats.md": f"{tag}_kernel_stats.md", "pmc_traffic.json": "pmc_traffic.json", "gemm_counters.txt": f"{tag}_gemm_attn_counters.txt",
         "bench_under_rocprof.json": f"{tag}_bench_under_rocprof.json", "bench_bf16x3_under_rocprof.json": f"{tag}_bench_bf16x3_under_rocprof.json",
         "shapes_config2.txt": f"{tag}_shapes_config2.txt", "shapes_config3.txt": f"{tag}_shapes_config3.txt", "shapes_config5.txt": f"{tag}_shapes_config5.txt"}
for a, b in pairs.items():
    shutil.copy(os.path.join(src, a), os.path.join(dst, b))
shutil.copy(os.path.join(src, "pmc_traffic.json"), os.path.join(dst, f"{tag}_pmc_hbm_traffic.json"))
with open(os.path.join(dst, f"{tag}_gemm_attn_mfma_util.md"), "w") as f:
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "mfma_util.py"), os.path.join(dst, f"{tag}_gemm_attn_counters.txt")], stdout=f)
# the traced bf16x3 step (x3_seq.txt: gap, duration [us], kernel name -- one step)
rows = collections.defaultdict(lambda: [0, 0.0])
for l in open(os.path.join(src, "x3_seq.txt")):
    parts = l.rstrip("\n").split(None, 2)
    try:
        dur = float(parts[1])
    except (IndexError, ValueError):
        continue
    rows[parts[2][:90]][0] += 1
    rows[parts[2][:90]][1] += dur
x3 = os.path.join(dst, f"{tag}_kernel_stats_bf16x3.md")
head = [l.rstrip("\n") for l in open(x3)][:3] if os.path.exists(x3) else []
out = head + ["| kernel | launches/step | ms/step | avg us |", "|---|---|---|---|"]
out += [f"| `{n}` | {c} | {d / 1000:.3f} | {d / c:.1f} |" for n, (c, d) in sorted(rows.items(), key=lambda kv: -kv[1][1])]
out.append(f"\nGPU-busy total: {sum(v[1] for v in rows.values()) / 1000:.3f} ms/step ({sum(v[0] for v in rows.values())} launches)")
open(x3, "w").write("\n".join(out) + "\n")
import bench
ok = json.load(open(os.path.join(dst, "pmc_traffic.json")))["_csrc_sha256"] == bench.csrc_hash()
print("copied", len(pairs) + 3, "files; source stamp", "ok" if ok else "STALE")
sys.exit(0 if ok else 1)
