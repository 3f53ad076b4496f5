#!/bin/bash
# Round profile of the headline bench on the GPU box: kernel-trace summary, HBM traffic (FETCH_SIZE / WRITE_SIZE in separate
# passes), MFMA-busy of the GEMM kernels.  Writes gpurun_out/rNN_*; copy what should be judged into profiles/.
#   bash tools/profile_round.sh r02
set -e
TAG=${1:-r02}
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/${TAG}_prof
rm -rf $OUT && mkdir -p $OUT
ARGS="bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats -d $OUT/trace -- python3 $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
DB=$(find $OUT/trace -name "*.db" | head -1)
python tools/rocpd_stats.py $DB 26 $OUT/kernel_stats.md > /dev/null
PARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --graph off"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -- python3 $PARGS > /dev/null 2> $OUT/fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -- python3 $PARGS > /dev/null 2> $OUT/write.log
python tools/pmc_traffic.py $(find $OUT/fetch -name "*.db" | head -1) $(find $OUT/write -name "*.db" | head -1) $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum -d $OUT/mfma -- python3 $PARGS > /dev/null 2> $OUT/mfma.log
python tools/pmc_dump.py $OUT/mfma gemm > $OUT/gemm_counters.txt
python tools/pmc_dump.py $OUT/mfma attn >> $OUT/gemm_counters.txt
cp $OUT/pmc_traffic.json profiles/pmc_traffic.json      # same sources: the bench line below quotes it (copy it back into the repo with the rest)
python bench.py --steps 50 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
rm -rf $OUT/trace $OUT/fetch $OUT/write $OUT/mfma
ls -la $OUT
