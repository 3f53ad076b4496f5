# Kernel-trace of the headline bench; idle time between kernels and one step's launch sequence (tools/rocpd_gaps.py).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/gaps
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT/trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench.json 2> $OUT/trace.log
DB=$(find $OUT/trace -name "*.db" | head -1)
DM_GAPS_SKIP_LAST=${DM_GAPS_SKIP_LAST:-7} python tools/rocpd_gaps.py $DB $OUT/gaps.md $OUT/seq.txt > /dev/null      # (bench.py ends with 2 x 3 eager, event-timed steps)
rm -rf $OUT/trace
