#!/bin/bash
# SQ instruction counts + render time of several builds: tools/pmc_variants.sh NAME...  ("default" = lib/libdtsim.so); N, MAP env vars
export TMPDIR=/tmp
for v in "$@"; do
  if [ $v = default ]; then unset DTSIM_LIB; else export DTSIM_LIB=$PWD/gym-duckietown_amd/lib/libdtsim_$v.so; fi
  OUT=/tmp/pmcv_$v; rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && K=3 N=${N:-2048} rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT -o p -- python $GRAFT_REPO_ROOT/tools/time_render.py > $OUT/log 2>&1)
  echo "== $v: $(K=10 N=${N:-2048} python tools/time_render.py 2>&1 | tail -1 | sed 's/.*event/event/' | cut -c1-20)"
  python tools/rocpd_summary.py "$OUT/*.db" | grep -A6 -E "pmc\] .*(PixTabEPKNS|SampTabEPKjPtPi)" | grep -E "SQ_INSTS" | sed 's/^ *//'
done
