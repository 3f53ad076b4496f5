set -e
mkdir -p gpurun_out/r4e
for st in 0 4 8 12 16 24; do
  echo "== stagger $st (x1024 cycles)"
  DM_W4_STAGGER=$st DM_GEMM_W4=3 python tools/mb_epi.py w4set 2>/dev/null
done
