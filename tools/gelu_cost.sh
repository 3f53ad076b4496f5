#!/bin/bash
# What does the GELU arithmetic in the fc1 forward epilogue cost?  Same box, shipped library against tools/hip/variants/libdm_<V>.so
# (tools/gelu_cost_build.sh), fc1 forward (GELU + GELU' saved) and fc2 dgrad (x saved GELU') at the headline's token counts.
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cp deepmerge_amd/libdeepmerge_hip.so /tmp/lib_shipped.so
for r in $(seq 1 ${REPS:-2}); do
for v in shipped ${VARIANTS:-geluabl}; do
  if [ $v = shipped ]; then cp /tmp/lib_shipped.so deepmerge_amd/libdeepmerge_hip.so; else cp tools/hip/variants/libdm_$v.so deepmerge_amd/libdeepmerge_hip.so; fi
  echo "== library $v (round $r) =="
  ONLY=fc1.fwd,fc2.dgrad timeout -k 10 300 python tools/routing_check.py 16384 4096 2>&1 | grep -E "^fc" | cut -c1-150
  [ "${STEP:-0}" = 1 ] && timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline ms/step', d['ms_per_step'])"
done
done
cp /tmp/lib_shipped.so deepmerge_amd/libdeepmerge_hip.so
