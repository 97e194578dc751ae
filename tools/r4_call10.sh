#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
bash tools/ab_cfg.sh "c5" default ronoz ronoshade ronostream > $O/ab10.txt 2>&1
DTSIM_LIB=$PWD/gym-duckietown_amd/lib/libdtsim_rostats.so DTSIM_DEBUG_QUEUE=1 python bench.py --config c5 --steps 2 --warmup 1 --windows 1 --cpu-steps 0 --no-gather 2>&1 | grep "dtsim\]" | tail -6 > $O/stats10.txt
DTSIM_LIB=$PWD/gym-duckietown_amd/lib/libdtsim_rostats.so DTSIM_DEBUG_QUEUE=1 python bench.py --config c4 --steps 2 --warmup 1 --windows 1 --cpu-steps 0 --no-gather 2>&1 | grep "dtsim\]" | tail -6 >> $O/stats10.txt
cat $O/ab10.txt $O/stats10.txt
bash tools/run_configs.sh r04a > $O/configs10.txt 2>&1; tail -30 $O/configs10.txt
