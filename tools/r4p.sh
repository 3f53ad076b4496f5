set -e
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" 2>&1 | tail -3
cp deepmerge_amd/libdeepmerge_hip.so /tmp/new.so
for rep in 1 2; do
  cp tools/variants/lib_head.so deepmerge_amd/libdeepmerge_hip.so
  echo "== head $rep"; timeout -k 10 200 python bench.py --steps 200 --warmup 30 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  cp /tmp/new.so deepmerge_amd/libdeepmerge_hip.so
  echo "== new $rep"; timeout -k 10 200 python bench.py --steps 200 --warmup 30 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
