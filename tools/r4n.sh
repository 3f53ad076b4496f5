set -e
mkdir -p gpurun_out/r4n
for rep in 1 2 3; do
for v in tm2old r4; do
  cp tools/variants/lib_$v.so deepmerge_amd/libdeepmerge_hip.so
  python bench.py --steps 100 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r4n/bench_$v.json 2>/dev/null
  python - <<PY
import json
r=json.loads([l for l in open('gpurun_out/r4n/bench_$v.json') if l.startswith('{')][-1])
print('$v', r['value'], r['ms_per_step'], r['roofline']['frac'])
PY
done
done
cp tools/variants/lib_r4.so deepmerge_amd/libdeepmerge_hip.so
