#!/bin/bash
# SQ wait / busy and TA / TCP counters of k_raster_v3 for several builds: bash tools/prof_variants_pmc.sh TAG default NAME ...
TAG=$1; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
for v in "$@"; do
  if [ $v = default ]; then unset DTSIM_LIB; else export DTSIM_LIB=$GRAFT_REPO_ROOT/gym-duckietown_amd/lib/libdtsim_$v.so; fi
  K=3 N=${N:-4096} rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/${v}_sq -o sq -- python $GRAFT_REPO_ROOT/tools/time_render.py > $OUT/${v}_sq.log 2>&1
  K=3 N=${N:-4096} rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE -d $OUT/${v}_ta -o ta -- python $GRAFT_REPO_ROOT/tools/time_render.py > $OUT/${v}_ta.log 2>&1
done
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  echo "== $v"
  python tools/rocpd_summary.py "$OUT/${v}_*/*.db" 2>&1 | grep -E "pmc\] .*${KPAT:-SampTabEPKjPtPi}" -A9 | grep -v "^--" | grep -v "pmc\]"
done > $OUT/summary.txt
cat $OUT/summary.txt
