#!/bin/bash
# the passes bench.py's roofline object leans on: kernel trace + stats, then FETCH_SIZE, WRITE_SIZE and the SQ instruction
# counters each in their own run (MI355X_MICROARCH.md: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2)
#   bash tools/prof_bench_short.sh TAG [c3|c4|c5]   ->  gpurun_out/prof_bench_TAG/summary.txt (+ make_traffic_json.py for roofline.traffic)
TAG=${1:-r02}
CFG=${2:-c3}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_bench_$TAG
mkdir -p $OUT
cd /tmp
ARGS="--config $CFG --steps 20 --warmup 3 --cpu-steps 0 --windows 2 --no-gather"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/write -o write -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/sq1 -o sq1 -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE TA_BUSY_avr -d $OUT/sq2 -o sq2 -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/sq2.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py "$OUT/*/*.db" > $OUT/summary.txt 2>&1
tail -1 $OUT/trace.log
