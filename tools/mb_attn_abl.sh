#!/bin/bash
# Timing-only ablations of the two-workgroups-per-CU attention forward (library built with EXTRA=-DDM_ATTN_ABLATIONS; results of
# the ablated kernels are wrong by construction).  DM_ATTN_ABL bits: 1 no exp2, 2 no K fragment reads, 4 no V fragment reads,
# 8 no softmax arithmetic, 16 K / V DMA only for the first sample, 32 no M / P barriers, 64 every sample stages the chunk's FIRST
# sample (same addresses: L2 hits), 128 no write-back.
for a in ${ABLS:-0 1 2 4 6 8 14 16 32 48 62 64 128 192}; do
  echo "ABL=$a"; DM_ATTN_FWD2=1 DM_ATTN_ABL=$a FWD_ONLY=1 B=${B:-240} python tools/mb_attn.py 2>/dev/null | grep fwd
done
