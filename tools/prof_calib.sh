#!/bin/bash
# FETCH_SIZE calibration on 16-byte gathers (tools/ubench/gather_calib.hip): kernel trace, then the counter passes
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_calib
mkdir -p $OUT
cd /tmp
$GRAFT_REPO_ROOT/tools/ubench/gather_calib > $OUT/plain.log 2>&1
timeout 150 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- $GRAFT_REPO_ROOT/tools/ubench/gather_calib > $OUT/fetch.log 2>&1
timeout 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $OUT/tcc -o tcc -- $GRAFT_REPO_ROOT/tools/ubench/gather_calib > $OUT/tcc.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py "$OUT/*/*.db" > $OUT/summary.txt 2>&1
cat $OUT/plain.log; grep -E "\[pmc\]|FETCH_SIZE|TCC_" $OUT/summary.txt | head -60
