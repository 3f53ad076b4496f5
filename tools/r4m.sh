set -e
mkdir -p gpurun_out/r4m
cp deepmerge_amd/libdeepmerge_hip.so /tmp/lib_keep.so
for rep in 1 2 3; do
for v in a0 a2 a2l2; do
  cp tools/variants/lib_$v.so deepmerge_amd/libdeepmerge_hip.so
  python bench.py --steps 100 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r4m/bench_$v.json 2>/dev/null
  python - <<PY
import json
r=json.loads([l for l in open('gpurun_out/r4m/bench_$v.json') if l.startswith('{')][-1])
print('$v', r['value'], r['ms_per_step'], r['roofline']['frac'])
PY
done
done
for v in a0 a2 a2l2; do
  cp tools/variants/lib_$v.so deepmerge_amd/libdeepmerge_hip.so
  echo "== $v"; python tools/mb_epi.py w4set 2>/dev/null | grep -v " none"; python tools/mb_epi.py proj 2>/dev/null | head -1
done
cp /tmp/lib_keep.so deepmerge_amd/libdeepmerge_hip.so
