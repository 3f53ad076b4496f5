#!/bin/bash
# Same-box A/B of two kernel LIBRARIES under the same Python: tools/hip/variants/libdm_r04.so (tools/build_prev_round_library.sh ce9590d r04,
# see DESIGN.md section 5) against the shipped library; REPS alternating rounds of `bench.py --steps 100` (+ config 5 / config 3 with CONFIGS=1).
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cp deepmerge_amd/libdeepmerge_hip.so /tmp/lib_new.so
run() { timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline ${NUM:+--numerics $NUM} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
cfg() { timeout -k 10 300 python -c "
import json
from deepmerge_amd import workload as W
print(W.config5(steps=10, graph=True)['ms_per_step'], W.config3(steps=10)['ms_per_step'])" 2>/dev/null | tail -1; }
for i in $(seq 1 ${REPS:-3}); do
  cp tools/hip/variants/libdm_r04.so deepmerge_amd/libdeepmerge_hip.so; a=$(run); [ "${CONFIGS:-0}" = 1 ] && ac=$(cfg) || ac=""
  cp /tmp/lib_new.so deepmerge_amd/libdeepmerge_hip.so; b=$(run); [ "${CONFIGS:-0}" = 1 ] && bc=$(cfg) || bc=""
  echo "round $i: headline ms/step r04 library $a -> r05 library $b   | config 5, config 3 ms/step: r04 [$ac] -> r05 [$bc]"
done
cp /tmp/lib_new.so deepmerge_amd/libdeepmerge_hip.so
