set -e
mkdir -p gpurun_out/r4h
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm256" > gpurun_out/r4h/tests.log 2>&1 || { tail -40 gpurun_out/r4h/tests.log; exit 1; }
tail -2 gpurun_out/r4h/tests.log
for pm in 1 0 1 0; do
  echo "== DM_GEMM_256=2 persistent=$pm"
  DM_GEMM_256P=$pm DM_GEMM_256=2 DM_GEMM_W4=0 DM_GEMM_RING=0 python tools/mb_epi.py w4set 2>/dev/null
done
