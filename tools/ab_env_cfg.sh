#!/bin/bash
# Same-box A/B of ONE environment switch on config 5 (v3 [6,4,2], 120 pairs / GPU) and config 3 (ViT-B/16): VAR=name A=value B=value [REPS=2] bash tools/ab_env_cfg.sh
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cfg() { timeout -k 10 300 python -c "
from deepmerge_amd import workload as W
print(W.config5(steps=10, graph=True)['ms_per_step'], W.config3(steps=10)['ms_per_step'])" 2>/dev/null | tail -n 1; }
for i in $(seq 1 ${REPS:-2}); do echo "round $i (config 5, config 3 ms/step): $VAR=$A [$(env $VAR=$A bash -c "$(declare -f cfg); cfg")] | $VAR=$B [$(env $VAR=$B bash -c "$(declare -f cfg); cfg")]"; done
