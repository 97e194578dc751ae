#!/bin/bash
# round 4, GPU call 2: dot2 / buffer-store variants of k_raster_v3 -- correctness against the round-3 arithmetic, A/B timing
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
L=$PWD/gym-duckietown_amd/lib
(timeout 200 tools/ubench/fmt_load) > $O/fmt_load2.txt 2>&1
rocprofv3-avail info --pc-sampling > $O/pcs_avail.txt 2>&1
rocprofv3-avail list --pc-sampling >> $O/pcs_avail.txt 2>&1
DTSIM_LIB=$L/libdtsim_old.so python tools/lib_frames.py old > $O/frames_cmp.txt 2>&1
python tools/lib_frames.py new >> $O/frames_cmp.txt 2>&1
python tools/lib_frames.py old new >> $O/frames_cmp.txt 2>&1
MAP=loop_only_duckies,small_loop_only_duckies DTSIM_LIB=$L/libdtsim_old.so python tools/lib_frames.py oldo >> $O/frames_cmp.txt 2>&1
MAP=loop_only_duckies,small_loop_only_duckies python tools/lib_frames.py newo >> $O/frames_cmp.txt 2>&1
python tools/lib_frames.py oldo newo >> $O/frames_cmp.txt 2>&1
bash tools/ab.sh default old d2only bsonly d2w4 d2w6 > $O/ab2.txt 2>&1
cat $O/frames_cmp.txt | grep -v amdgpu.ids; cat $O/ab2.txt; cat $O/fmt_load2.txt | tail -45; head -30 $O/pcs_avail.txt
