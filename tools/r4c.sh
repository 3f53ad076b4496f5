set -e
mkdir -p gpurun_out/r4c
python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" > gpurun_out/r4c/gemm_tests.log 2>&1 || { tail -30 gpurun_out/r4c/gemm_tests.log; exit 1; }
tail -2 gpurun_out/r4c/gemm_tests.log
for lean in 1 0 1 0; do
  DM_GEMM_EPI_LEAN=$lean python bench.py --steps 40 --warmup 4 --no-extras --no-cpu-baseline > gpurun_out/r4c/bench_lean$lean.json 2>/dev/null
  python - <<PY
import json
r=json.loads([l for l in open('gpurun_out/r4c/bench_lean$lean.json') if l.startswith('{')][-1])
print('lean=$lean', r['value'], r['ms_per_step'], r['roofline']['frac'])
PY
done
python tools/prof_shapes.py > gpurun_out/r4c/shapes.txt 2>&1
head -45 gpurun_out/r4c/shapes.txt
