#!/bin/bash
# counters of the C4 render pass (loop_pedestrians + domain randomisation + fisheye): bash tools/prof_c4.sh   (N env var, default 2048)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${TAGDIR:-prof_c4}
mkdir -p $OUT
cd /tmp
run() { MAP=${MAP:-loop_pedestrians} DR=${DR:-1} K=3 N=${N:-2048} timeout 200 rocprofv3 "$@" -d $OUT/$P -o $P -- python $GRAFT_REPO_ROOT/tools/time_render.py > $OUT/$P.log 2>&1; }
P=trace; run --kernel-trace --stats
P=sq1; run --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
P=sq2; run --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE TA_BUSY_avr
P=fetch; run --pmc FETCH_SIZE
P=write; run --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py "$OUT/*/*.db" > $OUT/summary.txt 2>&1
grep -E "calls" $OUT/summary.txt | head -8
