"""Matrix-pipe utilisation table from a tools/pmc_dump.py listing (SQ_VALU_MFMA_BUSY_CYCLES etc.).
usage: python tools/mfma_util.py counters.txt > profiles/rNN_gemm_attn_mfma_util.md"""
import re, sys, collections
rows = collections.defaultdict(dict)
for l in open(sys.argv[1]):
    m = re.match(r"(.+?)\s{2,}(\S+)\s+avg\s+([\d.]+)\s+n=(\d+)", l.rstrip())
    if m:
        rows[m.group(1)][m.group(2)] = (float(m.group(3)), int(m.group(4)))
print("# Matrix-pipe utilisation and L2 hit rate of the GEMM / attention kernels inside the training step")
print("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum -- python3 bench.py --steps 3 --warmup 1 --graph off")
print("# (tools/profile_round.sh; raw rows: the *_gemm_attn_counters.txt next to this file).  MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES (per shader engine) / 32 SIMDs / GRBM_GUI_ACTIVE (per XCD);")
print("# parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES (s_waitcnt / barrier), issue-stalled = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES; L2 hit = TCC_HIT / (TCC_HIT + TCC_MISS).")
print("| kernel | launches sampled | kernel cycles | MFMA-busy | waves parked | waves issue-stalled | L2 hit |")
print("|---|---|---|---|---|---|---|")
out = []
for k, c in rows.items():
    if "GRBM_GUI_ACTIVE" not in c or "SQ_VALU_MFMA_BUSY_CYCLES" not in c:
        continue
    cyc, n = c["GRBM_GUI_ACTIVE"]
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"][0] / 32.0 / cyc
    wave = c.get("SQ_WAVE_CYCLES", (1, 0))[0]
    hit, miss = c.get("TCC_HIT_sum", (0, 0))[0], c.get("TCC_MISS_sum", (0, 0))[0]
    out.append((cyc * n, f"| `{k[:80]}` | {n // 8} | {cyc:.0f} | {100 * busy:.1f} % | {100 * c.get('SQ_WAIT_ANY', (0, 0))[0] / wave:.0f} % | "
                f"{100 * c.get('SQ_WAIT_INST_ANY', (0, 0))[0] / wave:.0f} % | {100 * hit / max(hit + miss, 1):.0f} % |"))
for _, l in sorted(out, reverse=True):
    print(l)
