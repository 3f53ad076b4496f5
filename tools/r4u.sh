set -e
PYTHONPATH=. timeout -k 10 200 python tools/dbg_x3w8.py 2>&1 | grep "bad elements"
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_gpu_vit.py -x -q -k "attention_split or x3" 2>&1 | tail -3
run() { echo "== $*"; env "$@" timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline --numerics bf16x3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in 1 2; do
run DM_ATTN_X3_W8=0
run DM_ATTN_X3_W8=1
done
