#!/bin/bash
# SQ instruction counters of the headline render pass (small_loop, N = 4096) for variants of the library:
#   bash tools/prof_valu_variants.sh "" _nopool _norslv     (suffixes of gym-duckietown_amd/lib/libdtsim<suffix>.so)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_valuvar
mkdir -p $OUT
cd /tmp
for v in "$@"; do
  P=sq${v}
  DTSIM_LIB=$GRAFT_REPO_ROOT/gym-duckietown_amd/lib/libdtsim$v.so MAP=small_loop N=4096 K=3 timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $OUT/$P -o $P -- python $GRAFT_REPO_ROOT/tools/time_render.py > $OUT/$P.log 2>&1
  tail -1 $OUT/$P.log | cut -c1-100
done
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py "$OUT/*/*.db" > $OUT/summary.txt 2>&1
awk '/^==/{f=$2} /\[pmc\].*SampTabEPKjPtPi.kd$/{print f; for(i=0;i<9;i++){getline; print}}' $OUT/summary.txt
