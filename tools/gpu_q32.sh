#!/bin/bash
# first light of the 32-row attention forward: parity tests, then timing A/B against the 16-row pipelined kernel
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "attention" > gpurun_out/q32_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/q32_tests.log
tail -5 gpurun_out/q32_tests.log
for cfg in "64 256 0" "240 256 0" "256 197 1" "64 192 0"; do set -- $cfg
  for q in 0 1; do
    echo "== B=$1 N=$2 NOBIAS=$3 DM_ATTN_Q32=$q"
    B=$1 N=$2 NOBIAS=$3 FWD_ONLY=1 DM_ATTN_Q32=$q timeout -k 10 120 python tools/mb_attn.py 2>&1 | grep -v amdgpu
  done
done 2>&1 | tee gpurun_out/q32_timing.log
