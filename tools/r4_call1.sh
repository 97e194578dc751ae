#!/bin/bash
# round 4, GPU call 1: format-load micro-benchmark, baseline timing, phase timers, ATT attempt + PC sampling, wait counters
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
O=gpurun_out/r4
(timeout 120 tools/ubench/fmt_load) > $O/fmt_load.txt 2>&1
N=4096 K=20 python tools/time_render.py > $O/base_time.txt 2>&1
N=4096 K=20 python tools/time_render.py >> $O/base_time.txt 2>&1
DTSIM_LIB=$PWD/gym-duckietown_amd/lib/libdtsim_timing.so DTSIM_DEBUG_TIMERS=1 N=4096 K=3 python tools/time_render.py > $O/timing.txt 2>&1
bash tools/prof_stall.sh r4_stall k_raster_v3 > $O/stall_stdout.txt 2>&1
N=4096 KPAT=k_raster bash tools/prof_quick.sh r4_quick > $O/quick_stdout.txt 2>&1
tail -3 $O/base_time.txt; cat $O/fmt_load.txt; grep dtsim $O/timing.txt | head; head -60 $O/stall_stdout.txt
