#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
DTSIM_DEBUG_QUEUE=1 python bench.py --config c5 --steps 2 --warmup 1 --windows 1 --cpu-steps 0 --no-gather 2>&1 | grep "dtsim\]" | tail -5 > $O/stats12.txt
DTSIM_DEBUG_QUEUE=1 python bench.py --config c4 --steps 2 --warmup 1 --windows 1 --cpu-steps 0 --no-gather 2>&1 | grep "dtsim\]" | tail -5 >> $O/stats12.txt
cat $O/stats12.txt
cd /tmp; export TMPDIR=/tmp
for c in c5; do D=/tmp/tr_$c; rm -rf $D; mkdir -p $D; rocprofv3 --kernel-trace --stats -d $D -o t -- python $GRAFT_REPO_ROOT/bench.py --config $c --steps 10 --warmup 3 --windows 1 --cpu-steps 0 --no-gather > $D/log 2>&1; python $GRAFT_REPO_ROOT/tools/rocpd_summary.py "$D/*.db" | grep calls | head -6 | cut -c1-150; done
