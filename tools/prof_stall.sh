#!/bin/bash
# Stall attribution of the raster kernel: bash tools/prof_stall.sh <tag> [kernel-regex]     (N, K, MAP, DR env vars)
#   1. rocprofv3 --att (SQ thread trace) -- needs the trace-decoder library; the outcome (or the error) is kept in att.log
#   2. rocprofv3 stochastic PC sampling (hardware samples: PC, issued or not, the reason when not), then host-trap sampling
#      as the fall-back; tools/pcsamp_hist.py turns the samples of the kernel into per-instruction / per-reason histograms
# Output: gpurun_out/prof_<tag>/{att.log, pcs_*.log, pcs_*_hist.txt}; the raw samples stay on the box unless small.
TAG=${1:-stall}
KRE=${2:-k_raster}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
export N=${N:-4096} K=${K:-4}
if [ -z "$SKIP_ATT" ]; then
  timeout 240 rocprofv3 --att --att-target-cu 1 --att-shader-engine-mask 0x1 --kernel-include-regex "$KRE" --kernel-trace -d /tmp/att_$TAG -o att \
    -- python $GRAFT_REPO_ROOT/tools/time_render.py > $OUT/att.log 2>&1
  echo "rocprofv3 --att exit code $?" >> $OUT/att.log
  (ls -laR /tmp/att_$TAG 2>&1 | head -40) >> $OUT/att.log
fi
for M in ${PCS_METHODS:-stochastic host_trap}; do
  if [ $M = stochastic ]; then UNIT=cycles; IV=${PCS_CYCLES:-65536}; else UNIT=time; IV=${PCS_US:-500}; fi
  rm -rf /tmp/pcs_$M
  timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $M --pc-sampling-unit $UNIT --pc-sampling-interval $IV --kernel-trace \
    --output-format csv json -d /tmp/pcs_$M -o pcs -- python $GRAFT_REPO_ROOT/tools/time_render.py > $OUT/pcs_$M.log 2>&1
  echo "rocprofv3 pc sampling ($M, $UNIT, $IV) exit code $?" >> $OUT/pcs_$M.log
  (ls -laR /tmp/pcs_$M 2>&1 | head -30) >> $OUT/pcs_$M.log
  python $GRAFT_REPO_ROOT/tools/pcsamp_hist.py /tmp/pcs_$M "$KRE" > $OUT/pcs_${M}_hist.txt 2>&1
  # keep a small raw excerpt for the record (schema + first samples)
  for f in $(find /tmp/pcs_$M -name "*pc_sampling*.csv" | head -2); do head -c 200000 $f | gzip > $OUT/$(basename $f).head.gz; done
  for f in $(find /tmp/pcs_$M -name "*.json" | head -1); do head -c 300000 $f | gzip > $OUT/$(basename $f).head.gz; done
  if grep -q "samples of" $OUT/pcs_${M}_hist.txt; then break; fi     # the first method that delivers is enough
done
cd $GRAFT_REPO_ROOT
head -70 $OUT/pcs_*_hist.txt
