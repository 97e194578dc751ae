#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
bash tools/ab_cfg.sh "c3" default sort2 sort1 > $O/ab21.txt 2>&1; cat $O/ab21.txt
export TMPDIR=/tmp; cd /tmp
for v in default sort2 sort1; do
  if [ $v = default ]; then unset DTSIM_LIB; else export DTSIM_LIB=$GRAFT_REPO_ROOT/gym-duckietown_amd/lib/libdtsim_$v.so; fi
  rm -rf /tmp/f_$v; timeout 200 rocprofv3 --pmc FETCH_SIZE -d /tmp/f_$v -o f -- python $GRAFT_REPO_ROOT/bench.py --config c3 --steps 10 --warmup 3 --windows 1 --cpu-steps 0 --no-gather > /tmp/f_$v.log 2>&1
  echo "== $v"; python $GRAFT_REPO_ROOT/tools/rocpd_summary.py "/tmp/f_$v/*.db" | grep -A1 "pmc\] .*SampTabEPKjPtPi" | tail -1
done
