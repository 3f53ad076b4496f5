set -e
cp deepmerge_amd/libdeepmerge_hip.so /tmp/keep.so
cp tools/variants/lib_abl.so deepmerge_amd/libdeepmerge_hip.so
for ne in 0 1; do
  echo "== 128x128 kernel only (DM_GEMM_RING=0 DM_GEMM_256=0 DM_GEMM_W4=0), no-epilogue=$ne"
  DM_GEMM_NOEPI=$ne DM_GEMM_RING=0 DM_GEMM_256=0 DM_GEMM_W4=0 python tools/mb_epi.py w4set 2>/dev/null
  DM_GEMM_NOEPI=$ne DM_GEMM_RING=0 DM_GEMM_256=0 DM_GEMM_W4=0 python tools/mb_epi.py proj 2>/dev/null | head -2
done
cp /tmp/keep.so deepmerge_amd/libdeepmerge_hip.so
