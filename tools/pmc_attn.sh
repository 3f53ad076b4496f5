#!/bin/bash
# MFMA / VALU / LDS counters of the attention kernels (tools/mb_attn.py), separate --pmc passes, program directly after `--`.
set -euo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
Bc="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES"
RAW=/tmp/pmc_attn_raw            # rocprofv3's databases are tens of MB each: keep them off gpurun_out (its 64 MiB merge limit), dump text only
rm -rf $RAW; mkdir -p $RAW gpurun_out
for cfg in "64 256 0" "256 197 1"; do set -- $cfg
  export B=$1 N=$2 NOBIAS=$3        # stage 0 of the headline config (bias table); config 3 = ViT-B/16 pairs (no bias)
  rocprofv3 --pmc $A -d $RAW/a_$1_$2 -- python3 tools/mb_attn.py > gpurun_out/pmc_attn_a_$1_$2.log 2>&1
  rocprofv3 --pmc $Bc -d $RAW/b_$1_$2 -- python3 tools/mb_attn.py > gpurun_out/pmc_attn_b_$1_$2.log 2>&1
  python3 tools/mb_attn.py 2>&1 | grep -v amdgpu
done
for d in a_64_256 b_64_256 a_256_197 b_256_197; do echo "== $d"; python tools/pmc_dump.py $RAW/$d attn; done
rm -rf $RAW
