set -e
mkdir -p gpurun_out/r4l
for rep in 1 2; do
for v in 0 2 16; do
  cp tools/variants/lib_aux$v.so deepmerge_amd/libdeepmerge_hip.so
  python bench.py --steps 40 --warmup 4 --no-extras --no-cpu-baseline > gpurun_out/r4l/bench_aux$v.json 2>/dev/null
  python - <<PY
import json
r=json.loads([l for l in open('gpurun_out/r4l/bench_aux$v.json') if l.startswith('{')][-1])
print('aux=$v', r['value'], r['ms_per_step'], r['roofline']['frac'])
PY
done
done
for v in 0 2 16; do
  cp tools/variants/lib_aux$v.so deepmerge_amd/libdeepmerge_hip.so
  echo "== aux=$v"; python tools/mb_epi.py w4set 2>/dev/null
done
cp tools/variants/lib_aux0.so deepmerge_amd/libdeepmerge_hip.so
