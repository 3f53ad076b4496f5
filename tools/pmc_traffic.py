"""HBM bytes per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --graph off`, aggregated under the
names the in-library profiler (and bench.py's roofline object) uses.
usage: python tools/pmc_traffic.py FETCH.db WRITE.db out.json"""
import json, re, sqlite3, sys
from collections import defaultdict


def bench_name(k):
    """rocprof kernel name -> in-library profiler row name (None: not a profiled kernel)."""
    if "gemm_ring_kernel" in k:
        return "gemm_bf16_NT"
    m = re.search(r"gemm_(?:w4|p192)_kernel<(\d)", k) or re.search(r"gemm_(?:w4|p192)_kernelILi(\d)E", k)
    if m:
        return "gemm_bf16_" + ("NT", "NN", "TN")[int(m.group(1))]
    m = re.search(r"gemm256_kernel<(\d)>", k) or re.search(r"gemm256_kernelILi(\d)E", k)
    if m:
        return "gemm_bf16_" + ("NT", "NN", "TN")[int(m.group(1))]
    m = re.search(r"gemm_kernelI(DF16b|f)Li(\d)ELi\dE", k)
    if m:
        return "gemm_%s_%s" % ("bf16" if m.group(1) == "DF16b" else "f32", ("NT", "NN", "TN")[int(m.group(2))])
    m = re.search(r"gemm_kernel<(\w+), (\d), \d>", k)
    if m and m.group(1) in ("float",):
        return "gemm_f32_" + ("NT", "NN", "TN")[int(m.group(2))]
    if "gemm_kernel<bool _Accum" in k:          # rocprof's rendering of <__bf16, 1 (NN), TM>
        return "gemm_bf16_NN"
    if "attn_fwd" in k:
        return "attn_fwd_bf16" if ("DF16b" in k or "pipe" in k) else "attn_fwd_f32"
    if "attn_bwd" in k:
        return "attn_bwd_bf16" if ("DF16b" in k or "pipe" in k) else "attn_bwd_f32"
    if "adam_kernel" in k:
        return "adam_kernel"
    return None


def collect(path, counter):
    db = sqlite3.connect(path)
    tot, cnt = defaultdict(float), defaultdict(int)
    for name, val in db.execute("select name, counter_value from pmc_events where counter_name = ?", (counter,)):
        b = bench_name(name)
        if b:
            tot[b] += float(val) * 1024.0        # the counters are in KiB
            cnt[b] += 1
    return tot, cnt


fetch, nf = collect(sys.argv[1], "FETCH_SIZE")
write, nw = collect(sys.argv[2], "WRITE_SIZE")
launches_per_profiler_row = {"attn_bwd_bf16": 2, "attn_bwd_f32": 2}          # dQ and dK/dV kernels are one profiled call
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import csrc_hash  # noqa: E402
out = {"_csrc_sha256": csrc_hash(), "_note": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, `bench.py --steps 3 --warmup 1 "
                "--graph off`), averaged over the launches of the step; FETCH_SIZE doubled per the gfx950 correction of "
                "/opt/skills/guides/MI355X_MICROARCH.md (validated on adam_kernel, whose traffic is known exactly). Units: bytes.",
       "_command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --graph off ; same with WRITE_SIZE ; tools/pmc_traffic.py"}
for k in sorted(fetch):
    per = launches_per_profiler_row.get(k, 1)
    out[k] = {"fetch_bytes": 2.0 * fetch[k] / nf[k] * per, "write_bytes": write[k] / max(nw[k], 1) * per, "launches_sampled": nf[k] // per}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out.items():
    if not k.startswith("_"):
        print(f"{k:16s} fetch {v['fetch_bytes']/1e6:8.1f} MB  write {v['write_bytes']/1e6:8.1f} MB  ({v['launches_sampled']} launches)")
