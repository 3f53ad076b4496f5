# One traced step of the bf16x3 numerics mode: launch sequence, kernel statistics.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/x3prof
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT/trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --numerics bf16x3 > $OUT/bench.json 2> $OUT/trace.log
DB=$(find $OUT/trace -name "*.db" | head -1)
python tools/rocpd_gaps.py $DB $OUT/gaps.md $OUT/seq.txt > /dev/null
python tools/rocpd_stats.py $DB 26 $OUT/kernel_stats.md > /dev/null
rm -rf $OUT/trace
