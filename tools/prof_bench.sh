#!/bin/bash
# rocprofv3 of the bench command: kernel trace + stats, then HBM traffic counters in their own passes.
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_bench_$TAG
mkdir -p $OUT
cd /tmp
ARGS="--steps 20 --warmup 3 --cpu-steps 0"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/write -o write -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/sq1 -o sq1 -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT -d $OUT/sq2 -o sq2 -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/sq2.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py "$OUT/*/*.db" > $OUT/summary.txt 2>&1
tail -2 $OUT/trace.log
