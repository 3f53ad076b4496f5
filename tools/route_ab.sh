#!/bin/bash
# In-step A/B of DM_GEMM_ROUTE overrides (one product -> one kernel family), headline step, same box: ROUTES="a;b;c" each run REPS times
# alternating with the default routing.
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
run() { timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
IFS=';' read -ra LIST <<< "${ROUTES:?}"
for i in $(seq 1 ${REPS:-2}); do
  echo "round $i: default $(run)"
  for r in "${LIST[@]}"; do echo "round $i: DM_GEMM_ROUTE=$r $(DM_GEMM_ROUTE="$r" run)"; done
done
