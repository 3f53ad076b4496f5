set -e
mkdir -p gpurun_out/r4k
timeout -k 10 600 python bench.py --gpus 6 --backend gloo --steps 3 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r4k/dp6_gloo.json 2> gpurun_out/r4k/dp6_gloo.err || { tail -20 gpurun_out/r4k/dp6_gloo.err; exit 1; }
python - <<'PY'
import json
r=json.loads([l for l in open('gpurun_out/r4k/dp6_gloo.json') if l.startswith('{')][-1])
print(r['n_gpus'], r['value'], r['ms_per_step'], json.dumps(r['data_parallel']))
PY
DM_DP_FORCE=1 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r4k/rccl1.json 2> gpurun_out/r4k/rccl1.err || { tail -20 gpurun_out/r4k/rccl1.err; exit 1; }
python - <<'PY'
import json
r=json.loads([l for l in open('gpurun_out/r4k/rccl1.json') if l.startswith('{')][-1])
print(r['n_gpus'], r['value'], r['ms_per_step'], json.dumps(r['data_parallel']))
PY
python tools/mb_yardstick.py > gpurun_out/r4k/yardstick.txt 2>&1
tail -14 gpurun_out/r4k/yardstick.txt
