set -e
mkdir -p gpurun_out/r4d
python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm or w4" > gpurun_out/r4d/gemm_tests.log 2>&1 || { tail -30 gpurun_out/r4d/gemm_tests.log; exit 1; }
tail -2 gpurun_out/r4d/gemm_tests.log
for cfg in "1 1" "1 0" "3 1" "1 1" "3 1"; do
  set -- $cfg
  DM_GEMM_W4=$1 DM_GEMM_EPI_LEAN=$2 python bench.py --steps 40 --warmup 4 --no-extras --no-cpu-baseline > gpurun_out/r4d/bench_w4$1_lean$2.json 2>/dev/null
  python - <<PY
import json
r=json.loads([l for l in open('gpurun_out/r4d/bench_w4$1_lean$2.json') if l.startswith('{')][-1])
print('w4=$1 lean=$2', r['value'], r['ms_per_step'], r['roofline']['frac'])
PY
done
DM_GEMM_W4=3 python tools/prof_shapes.py > gpurun_out/r4d/shapes_w43.txt 2>&1
head -22 gpurun_out/r4d/shapes_w43.txt
python tools/prof_shapes.py > gpurun_out/r4d/shapes_w41.txt 2>&1
head -22 gpurun_out/r4d/shapes_w41.txt
