#!/bin/bash
# round 5, call A: frame-tile launch order of k_raster_v3 (stride permutation / top-bottom alternation) against the default, one box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
bash tools/ab_cfg.sh "c3" default perm1 perm2 > $O/ab_perm.txt 2>&1
DTSIM_DEBUG_QUEUE=1 N=4096 K=2 python tools/time_render.py 2>&1 | grep -E "exact-path|resolve:|event" | tail -3 >> $O/ab_perm.txt
cat $O/ab_perm.txt
