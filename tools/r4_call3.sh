#!/bin/bash
# round 4, GPU call 3: compact slot sub-blocks (DT_V3_MAP 2) -- frames identical?  A/B timing
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
L=$PWD/gym-duckietown_amd/lib
python tools/lib_frames.py new > $O/frames_cmp3.txt 2>&1
for v in m16 m32 m8; do DTSIM_LIB=$L/libdtsim_$v.so python tools/lib_frames.py $v >> $O/frames_cmp3.txt 2>&1; python tools/lib_frames.py new $v >> $O/frames_cmp3.txt 2>&1; done
bash tools/ab.sh default m16 m32 m8 old > $O/ab3.txt 2>&1
grep -v amdgpu.ids $O/frames_cmp3.txt; cat $O/ab3.txt
