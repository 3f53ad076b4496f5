set -e
mkdir -p gpurun_out/r4f
python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" > gpurun_out/r4f/gemm_tests.log 2>&1 || { tail -40 gpurun_out/r4f/gemm_tests.log; exit 1; }
tail -2 gpurun_out/r4f/gemm_tests.log
for fs in 1 0 1 0; do
  DM_GEMM_W4_SLICES=$fs python bench.py --steps 40 --warmup 4 --no-extras --no-cpu-baseline > gpurun_out/r4f/bench_fs$fs.json 2>/dev/null
  python - <<PY
import json
r=json.loads([l for l in open('gpurun_out/r4f/bench_fs$fs.json') if l.startswith('{')][-1])
print('w4_slices=$fs', r['value'], r['ms_per_step'], r['roofline']['frac'])
PY
done
python tools/prof_shapes.py > gpurun_out/r4f/shapes.txt 2>&1
grep -E "4096x768x3072|1024x768x3072|4096x768x2304|1024x768x2304|4096x768x4096|total" gpurun_out/r4f/shapes.txt
