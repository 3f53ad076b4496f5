set -e
mkdir -p gpurun_out/r4j
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -x -q > gpurun_out/r4j/tests.log 2>&1 || { tail -40 gpurun_out/r4j/tests.log; exit 1; }
tail -2 gpurun_out/r4j/tests.log
for i in 1 2; do
python bench.py --steps 40 --warmup 4 --no-extras --no-cpu-baseline > gpurun_out/r4j/bench.json 2>/dev/null
python - <<'PY'
import json
r=json.loads([l for l in open('gpurun_out/r4j/bench.json') if l.startswith('{')][-1])
print(r['value'], r['ms_per_step'], r['roofline']['frac'])
PY
done
python tools/prof_shapes.py > gpurun_out/r4j/shapes.txt 2>&1
grep -E "f32|16384x3072x768_e3|total" gpurun_out/r4j/shapes.txt
