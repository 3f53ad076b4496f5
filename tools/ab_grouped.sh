#!/bin/bash
# Same-box A/B for the grouped weight gradients (round 5): the library of the previous commit (tools/build_prev_round_library.sh HEAD~ prev;
# its dm_gemm_grouped stub is the separate calls) with grouping off, the shipped library with grouping off (is the kernel refactor
# neutral?) and the shipped library as shipped (DM_WGRAD_GROUP default).  REPS alternating rounds of bench.py --steps 100; NUM=bf16x3 for the
# tolerance mode; CONFIGS=1 adds config 5 / config 3.
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cp deepmerge_amd/libdeepmerge_hip.so /tmp/lib_new.so
run() { timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline ${NUM:+--numerics $NUM} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
cfg() { timeout -k 10 300 python -c "
from deepmerge_amd import workload as W
print(W.config5(steps=10, graph=True)['ms_per_step'], W.config3(steps=10)['ms_per_step'])" 2>/dev/null | tail -n 1; }
for i in $(seq 1 ${REPS:-3}); do
  cp tools/hip/variants/libdm_prev.so deepmerge_amd/libdeepmerge_hip.so; a=$(DM_WGRAD_GROUP=0 run); [ "${CONFIGS:-0}" = 1 ] && ac=$(DM_WGRAD_GROUP=0 cfg) || ac=""
  cp /tmp/lib_new.so deepmerge_amd/libdeepmerge_hip.so; b=$(DM_WGRAD_GROUP=0 run); c=$(run); [ "${CONFIGS:-0}" = 1 ] && cc=$(cfg) || cc=""
  echo "round $i: headline ms/step previous library $a | shipped library, grouping off $b | shipped, grouping on $c   | config 5, config 3: previous [$ac] -> shipped [$cc]"
done
cp /tmp/lib_new.so deepmerge_amd/libdeepmerge_hip.so
