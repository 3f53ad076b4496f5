set -e
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "skinny" 2>&1 | tail -2
run() { echo "== $*"; env "$@" timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
run DM_GEMM_SKINNY=0
run DM_GEMM_SKINNY=1
run DM_GEMM_SKINNY=0
run DM_GEMM_SKINNY=1
