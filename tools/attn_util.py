"""Table of the attention kernels' matrix-pipe utilisation from the listing `tools/pmc_attn.sh` prints (sections `== a_B_N` / `== b_B_N` of
tools/pmc_dump.py rows: kernel, counter, avg).  MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / 32 SIMDs / GRBM_GUI_ACTIVE; us = GRBM_GUI_ACTIVE / 2.4 GHz (under the
profiler); parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES; issue-stalled = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES.      python tools/attn_util.py listing.txt"""
import collections, re, sys
sec, data = None, collections.defaultdict(lambda: collections.defaultdict(dict))
for l in open(sys.argv[1]):
    m = re.match(r"== ([ab])_(\d+)_(\d+)", l)
    if m:
        sec = (int(m.group(2)), int(m.group(3)))
        continue
    m = re.match(r"(.+?)\s+([A-Z][A-Z0-9_a-z]+)\s+avg\s+([\d.]+)\s+n=", l)
    if m and sec:
        data[sec][m.group(1).strip()][m.group(2)] = float(m.group(3))
shapes = {(64, 256): "B = 64, N = 256, bias table", (256, 197): "B = 256, N = 197, no bias"}
print("| shape | kernel | us | MFMA busy | waves parked / issue-stalled | LDS conflict cycles / LDS cycles |")
print("|---|---|---|---|---|---|")
for sec in sorted(data, key=lambda s: -s[0]):
    for k, c in sorted(data[sec].items()):
        if "GRBM_GUI_ACTIVE" not in c or "float" in k or "SQ_VALU_MFMA_BUSY_CYCLES" not in c:
            continue
        g = c["GRBM_GUI_ACTIVE"]
        busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 32.0 / g
        wc = c.get("SQ_WAVE_CYCLES", 0) or 1
        print(f"| {shapes.get(sec, sec)} | `{k[:58]}` | {g / 2400:.0f} | {100 * busy:.1f} % | {100 * c.get('SQ_WAIT_ANY', 0) / wc:.0f} % / {100 * c.get('SQ_WAIT_INST_ANY', 0) / wc:.0f} % | "
              f"{c.get('SQ_LDS_BANK_CONFLICT', 0) / 1000:.0f} k / {c.get('SQ_LDS_IDX_ACTIVE', 0) / 1000:.0f} k |")
