#!/bin/bash
# LDS counters of the GEMM kernels on the stage-0 shapes (tools/mb_yardstick.py): bank-conflict cycles against all LDS-array cycles.
set -euo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
RAW=/tmp/pmc_gemm_lds; rm -rf $RAW; mkdir -p $RAW gpurun_out
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $RAW/a -- python3 tools/mb_yardstick.py > gpurun_out/pmc_gemm_lds.log 2>&1
python tools/pmc_dump.py $RAW/a "" | grep -v "Cijk\|at::\|elementwise" > gpurun_out/pmc_gemm_lds.txt
rm -rf $RAW
