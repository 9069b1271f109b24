#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace + separate PMC passes around bench.py.
# usage: tools/profile_round.sh <tag>     (outputs under gpurun_out/prof_<tag>/)
set -u
TAG=${1:-r1}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 5 --warmup 1 --no-cpu-baseline"
cd $REPO
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > $OUT/bench_write.json 2> $OUT/pmc_write.log
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_sq -- python3 bench.py $ARGS > $OUT/bench_sq.json 2> $OUT/pmc_sq.log
python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
