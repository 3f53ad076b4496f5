#!/bin/bash
# Builds the variant library for the buffer-store hazard experiment: every GEMM kernel family (dm_gemm, dm_gemm256,
# dm_gemm_ring, dm_gemm_w4) with NO pad behind the epilogues' 16-byte buffer stores (-DDM_EPI_STORE_PAD=0); the row step stays in the
# SGPR soffset as shipped.  tools/epi_store_hazard.sh runs the exact-integer GELU + aux product on each family against it.
set -euo pipefail
cd "$(dirname "$0")/.."
CS=deepmerge_amd/csrc
mkdir -p tools/hip/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$CS -Wno-unused-result -Wno-unused-value -Wno-pass-failed -DDM_EPI_STORE_PAD=0"
OBJS=$(ls $CS/build/*.o | grep -v -E "dm_gemm.o|dm_gemm256.o|dm_gemm_ring.o|dm_gemm_w4.o")
VAR=""
for f in dm_gemm dm_gemm256 dm_gemm_ring dm_gemm_w4; do
  hipcc $FLAGS -c $CS/$f.hip -o tools/hip/variants/${f}_s0.o &
  VAR="$VAR tools/hip/variants/${f}_s0.o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-z,defs -o tools/hip/variants/libdm_s0all.so $OBJS $VAR
rm -f tools/hip/variants/libdm_s0.so tools/hip/variants/libdm_s2.so tools/hip/variants/libdm_v0.so tools/hip/variants/libdm_v2.so tools/hip/variants/dm_gemm_s2.o tools/hip/variants/dm_gemm_v0.o tools/hip/variants/dm_gemm_v2.o
echo built s0all
