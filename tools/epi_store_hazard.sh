#!/bin/bash
# Buffer-store hazard experiment: the exact-integer GELU + aux product on the 128x128 kernel, REPS runs per variant library
# built by tools/epi_store_hazard_build.sh (s0all: every GEMM family without the pad behind its 16-byte epilogue stores).
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cp deepmerge_amd/libdeepmerge_hip.so /tmp/lib_shipped.so
for v in shipped s0all; do
  if [ $v = shipped ]; then cp /tmp/lib_shipped.so deepmerge_amd/libdeepmerge_hip.so; else cp tools/hip/variants/libdm_$v.so deepmerge_amd/libdeepmerge_hip.so; fi
  bad=0
  for r in $(seq 1 ${REPS:-6}); do
    n=$(timeout -k 10 300 python tools/dbg_epi.py 2>&1 | grep "^family" | awk '{printf "%s:%s ", $2, $5}')
    bad="$bad | ${n:-err}"
  done
  echo "variant $v: wrong elements per family and run (4096 x 3072 outputs, two epilogues each) = $bad"
done
cp /tmp/lib_shipped.so deepmerge_amd/libdeepmerge_hip.so
