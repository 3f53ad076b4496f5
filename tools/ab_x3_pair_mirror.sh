#!/bin/bash
# bf16x3 step with the weights' plane pairs kept by the Adam kernel (default) against per-step splits (DM_X3_PAIR_MIRROR=0), same box, alternating.
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
run() { timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline --numerics bf16x3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for i in $(seq 1 ${REPS:-3}); do echo "round $i: per-step splits $(DM_X3_PAIR_MIRROR=0 run) | pairs from the optimizer $(run)"; done
