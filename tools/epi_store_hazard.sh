#!/bin/bash
# Buffer-store hazard experiment: the exact-integer GELU + aux product on the 128x128 kernel, REPS runs per variant library
# built by tools/epi_store_hazard_build.sh (s = row step in the SGPR soffset as shipped, v = added to the vector offset; number = pad cycles).
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cp deepmerge_amd/libdeepmerge_hip.so /tmp/lib_shipped.so
for v in shipped s0 s2 v0 v2; do
  if [ $v = shipped ]; then cp /tmp/lib_shipped.so deepmerge_amd/libdeepmerge_hip.so; else cp tools/hip/variants/libdm_$v.so deepmerge_amd/libdeepmerge_hip.so; fi
  bad=0
  for r in $(seq 1 ${REPS:-6}); do
    n=$(timeout -k 10 120 python tools/dbg_epi.py 2>&1 | grep "forced 128 bad" | awk '{print $4}')
    bad="$bad+${n:-err}"
  done
  echo "variant $v: wrong aux elements per run (4096 x 3072 outputs each) = $bad"
done
cp /tmp/lib_shipped.so deepmerge_amd/libdeepmerge_hip.so
