set -e
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -x -q -k "folded or x3 or parity or fused_column" 2>&1 | tail -4
run() { echo "== $*"; env "$@" timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline --numerics bf16x3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
run DM_NOOP=1
run DM_NOOP=1
