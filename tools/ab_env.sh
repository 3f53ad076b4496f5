#!/bin/bash
# Same-box A/B of ONE environment switch on the headline step: VAR=name A=value B=value [NUM=bf16x3] [REPS=3] bash tools/ab_env.sh
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
run() { timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline ${NUM:+--numerics $NUM} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for i in $(seq 1 ${REPS:-3}); do echo "round $i: $VAR=$A $(env $VAR=$A bash -c "$(declare -f run); run") | $VAR=$B $(env $VAR=$B bash -c "$(declare -f run); run")"; done
