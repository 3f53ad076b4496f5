# A/B of ONE tuning switch on the headline step: default / setting alternating REPS times on the same box (100 graph-replayed steps each)
set -e
run() { env "$@" timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline ${NUM:+--numerics $NUM} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for setting in "$@"; do
  line="$setting:"
  for i in $(seq 1 ${REPS:-3}); do a=$(run DM_NOOP=1); b=$(run $setting); line="$line  $a->$b"; done
  echo "$line"
done
