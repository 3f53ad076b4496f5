#!/bin/bash
# Variant library for tools/gelu_cost.sh: the GEMM families with the fast GELU / GELU' replaced by a three-instruction stand-in
# (-DDM_GELU_ABLATE; results are wrong on purpose) -- the time difference is what the epilogue's GELU arithmetic costs.
set -euo pipefail
cd "$(dirname "$0")/.."
CS=deepmerge_amd/csrc
mkdir -p tools/hip/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$CS -Wno-unused-result -Wno-unused-value -Wno-pass-failed ${VARIANT_FLAGS:--DDM_GELU_ABLATE}"
NAME=${VARIANT_NAME:-geluabl}
OBJS=$(ls $CS/build/*.o | grep -v -E "dm_gemm.o|dm_gemm256.o|dm_gemm_ring.o|dm_gemm_w4.o")
VAR=""
for f in dm_gemm dm_gemm256 dm_gemm_ring dm_gemm_w4; do
  hipcc $FLAGS -c $CS/$f.hip -o tools/hip/variants/${f}_$NAME.o &
  VAR="$VAR tools/hip/variants/${f}_$NAME.o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-z,defs -o tools/hip/variants/libdm_$NAME.so $OBJS $VAR
rm -f $VAR
echo built $NAME
