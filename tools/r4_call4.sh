#!/bin/bash
# round 4, GPU call 4: what a vector instruction / a load / a store of k_raster_v3's loop is worth (ablation builds, wrong frames)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
bash tools/ab.sh default ch2 ch1 ablw ch1w noload oneblk nostore nores noresslow nostnold loop1 loop1ns > $O/ab4.txt 2>&1
cat $O/ab4.txt
