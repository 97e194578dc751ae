#!/bin/bash
# quick PMC look at the render kernels: bash tools/prof_quick.sh <tag>   (N env var, default 1024)
TAG=${1:-q}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
K=3 N=${N:-1024} rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/pmc1 -o pmc1 -- python $GRAFT_REPO_ROOT/tools/time_render.py > $OUT/pmc1.log 2>&1
K=3 N=${N:-1024} rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM -d $OUT/pmc2 -o pmc2 -- python $GRAFT_REPO_ROOT/tools/time_render.py > $OUT/pmc2.log 2>&1
K=3 N=${N:-1024} rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE -d $OUT/pmc3 -o pmc3 -- python $GRAFT_REPO_ROOT/tools/time_render.py > $OUT/pmc3.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py "$OUT/*/*.db" > $OUT/summary.txt 2>&1
grep -E "pmc\] .*${KPAT:-k_raster}" -A9 $OUT/summary.txt | head -60
