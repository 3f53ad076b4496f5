set -e
timeout -k 10 900 python -m pytest tests/test_gpu_modules.py tests/test_gpu_kernels.py -x -q -k "x3 or adam or trainer or graph_replayed or parity" 2>&1 | tail -4
run() { echo "== $*"; env "$@" timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline --numerics bf16x3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in 1 2; do
run DM_X3_WEIGHT_MIRROR=0
run DM_X3_WEIGHT_MIRROR=1
done
