# A/B of tuning switches on config 5's step (v3 [6,4,2], 120 pairs per GPU, hipGraph replay, 10 steps): default / setting alternating REPS times on the same box
set -e
run() { env "$@" timeout -k 10 300 python -c "
from deepmerge_amd import workload as W
print(W.config5(steps=10, graph=True, numerics='${NUM:-bf16}')['ms_per_step'])" 2>/dev/null | tail -1; }
for setting in "$@"; do
  line="$setting:"
  for i in $(seq 1 ${REPS:-2}); do a=$(run DM_NOOP=1); b=$(run $setting); line="$line  $a->$b"; done
  echo "$line"
done
