# Full end-of-round refresh: tools/profile_round.sh + one traced bf16x3 step + the per-shape tables of configs 2 / 3 / 5.
set -e
TAG=${1:-r05}
bash tools/profile_round.sh $TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG}_prof
rocprofv3 --kernel-trace -d $OUT/x3trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --numerics bf16x3 > $OUT/bench_bf16x3_under_rocprof.json 2> $OUT/x3trace.log
DB=$(find $OUT/x3trace -name "*.db" | head -1)
python tools/rocpd_gaps.py $DB $OUT/x3_gaps.md $OUT/x3_seq.txt > /dev/null
rm -rf $OUT/x3trace
python tools/prof_shapes.py > $OUT/shapes_config2.txt 2> /dev/null
python tools/prof_shapes.py --config 3 > $OUT/shapes_config3.txt 2> /dev/null
python tools/prof_shapes.py --depth 6,4,2 --pairs 120 > $OUT/shapes_config5.txt 2> /dev/null
ls $OUT
