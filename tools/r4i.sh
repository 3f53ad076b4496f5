set -e
mkdir -p gpurun_out/r4i
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4i/gpu_tests.log 2>&1 || { tail -40 gpurun_out/r4i/gpu_tests.log; exit 1; }
tail -3 gpurun_out/r4i/gpu_tests.log
python bench.py --steps 50 --warmup 5 > gpurun_out/r4i/bench.json 2> gpurun_out/r4i/bench.err
python - <<'PY'
import json
r=json.loads([l for l in open('gpurun_out/r4i/bench.json') if l.startswith('{')][-1])
print(len(json.dumps(r)), r['value'], r['ms_per_step'], r['roofline']['frac'], r.get('value_at_tolerance'), r.get('ms_per_step_at_tolerance'))
print(json.dumps(r['extras'])[:1500])
PY
