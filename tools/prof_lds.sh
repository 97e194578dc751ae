#!/bin/bash
# LDS-side counters of the bench command (one pass): bash tools/prof_lds.sh [c3|c4|c5]  ->  gpurun_out/prof_lds_<cfg>/summary.txt
CFG=${1:-c3}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_lds_$CFG
mkdir -p $OUT
cd /tmp
ARGS="--config $CFG --steps 20 --warmup 3 --cpu-steps 0 --windows 2 --no-gather"
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA -d $OUT/lds -o lds -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/lds.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py "$OUT/*/*.db" > $OUT/summary.txt 2>&1
grep -A10 "\[pmc\].*SampTabEPKjPtPi" $OUT/summary.txt | head -14
