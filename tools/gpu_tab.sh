#!/bin/bash
# table-in-LDS attention kernels: parity, then timing against the dense-bias kernels
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "relpos_table or q32 or pipelined" > gpurun_out/tab_tests.log 2>&1
rc=$?; echo "tests rc=$rc" | tee -a gpurun_out/tab_tests.log; tail -5 gpurun_out/tab_tests.log
[ $rc -eq 0 ] || exit $rc
for cfg in "64 256 0" "64 192 0" "256 197 1"; do set -- $cfg
  echo "== B=$1 N=$2 NOBIAS=$3"
  B=$1 N=$2 NOBIAS=$3 timeout -k 10 120 python tools/mb_attn.py 2>&1 | grep -v amdgpu
done 2>&1 | tee gpurun_out/tab_timing.log
