#!/bin/bash
# the three passes bench.py's roofline object leans on: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE on their own
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_bench_$TAG
mkdir -p $OUT
cd /tmp
ARGS="--steps 20 --warmup 3 --cpu-steps 0"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/write -o write -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/write.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py "$OUT/*/*.db" > $OUT/summary.txt 2>&1
tail -1 $OUT/trace.log
