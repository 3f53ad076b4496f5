#!/bin/bash
# Builds variant libraries for the buffer-store hazard experiment: only dm_gemm.o (the 128x128 / 64x64 kernels) differs.
# Variants: <soff in V><pad cycles>: s16 (shipped: SGPR soffset + 16 cycles), s0, s2, v0 (VGPR offset, compiler's own hazard handling), v2
set -euo pipefail
cd "$(dirname "$0")/.."
CS=deepmerge_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$CS -Wno-unused-result -Wno-unused-value -Wno-pass-failed"
OBJS=$(ls $CS/build/*.o | grep -v dm_gemm.o)
for v in "s0:-DDM_EPI_STORE_PAD=0" "s2:-DDM_EPI_STORE_PAD=2" "v0:-DDM_EPI_SOFF_V=1 -DDM_EPI_STORE_PAD=0" "v2:-DDM_EPI_SOFF_V=1 -DDM_EPI_STORE_PAD=2"; do
  name=${v%%:*}; ex=${v#*:}
  hipcc $FLAGS $ex -c $CS/dm_gemm.hip -o tools/hip/variants/dm_gemm_$name.o
  hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-z,defs -o tools/hip/variants/libdm_$name.so $OBJS tools/hip/variants/dm_gemm_$name.o
  echo built $name
done
