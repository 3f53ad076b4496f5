#!/bin/bash
# timing ablations of the 32-row attention forward: rebuilds dm_attention_q32.o with -DDMQ_ABL=<bits> on the GPU box per variant
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p gpurun_out
OUT=gpurun_out/q32_abl.log
: > $OUT
for abl in ${ABLS:-0 1 2 4 8 16 32 64 128 12 28 158}; do
  touch deepmerge_amd/csrc/dm_attention_q32.hip
  make -C deepmerge_amd/csrc EXTRA=-DDMQ_ABL=$abl > gpurun_out/q32_abl_build.log 2>&1 || { echo "build failed for $abl" | tee -a $OUT; tail -5 gpurun_out/q32_abl_build.log; continue; }
  IFS=';' read -ra CFG_LIST <<< "${CFGS:-64 256 0;256 197 1}"        # CFGS="B N NOBIAS;B N NOBIAS"
  for cfg in "${CFG_LIST[@]}"; do set -- $cfg
    echo "abl=$abl B=$1 N=$2 NOBIAS=$3: $(B=$1 N=$2 NOBIAS=$3 FWD_ONLY=1 timeout -k 10 120 python tools/mb_attn.py 2>&1 | grep fwd)" | tee -a $OUT
  done
done
touch deepmerge_amd/csrc/dm_attention_q32.hip
make -C deepmerge_amd/csrc > /dev/null 2>&1
