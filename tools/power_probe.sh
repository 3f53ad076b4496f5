#!/bin/bash
# Samples sclk / power while bench.py runs a long timed region (are the GEMMs clock- or power-limited in the real step?).
python bench.py --no-cpu-baseline --steps 4000 --warmup 3 > gpurun_out/pp_bench.log 2>&1 &
BP=$!
for i in $(seq 1 30); do
  sleep 1.5
  echo "t=$i $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed 's/.*: //' | tr '\n' ' ')"
done
wait $BP
tail -1 gpurun_out/pp_bench.log | cut -c1-200
