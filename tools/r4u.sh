set -e
run() { echo "== $*"; env "$@" timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline --numerics bf16x3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
run DM_GEMM_FOLD_ROUTES=0
run DM_GEMM_FOLD_ROUTES=1
done
