#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
L=$PWD/gym-duckietown_amd/lib
DTSIM_LIB=$L/libdtsim_prev.so python tools/lib_frames.py prev > $O/frames20.txt 2>&1
python tools/lib_frames.py new >> $O/frames20.txt 2>&1; python tools/lib_frames.py prev new >> $O/frames20.txt 2>&1
N=2048 MAP=loop_only_duckies,small_loop_only_duckies DTSIM_LIB=$L/libdtsim_prev.so python tools/lib_frames.py prevo >> $O/frames20.txt 2>&1
N=2048 MAP=loop_only_duckies,small_loop_only_duckies python tools/lib_frames.py newo >> $O/frames20.txt 2>&1; python tools/lib_frames.py prevo newo >> $O/frames20.txt 2>&1
grep -v amdgpu $O/frames20.txt
bash tools/ab.sh default prev > $O/ab20.txt 2>&1; cat $O/ab20.txt
bash tools/ab_cfg.sh "c3 c5" default prev > $O/ab20b.txt 2>&1; cat $O/ab20b.txt
