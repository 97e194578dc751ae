#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
L=$PWD/gym-duckietown_amd/lib
DTSIM_LIB=$L/libdtsim_old.so python tools/lib_frames.py old > $O/frames_cmp6.txt 2>&1
python tools/lib_frames.py new >> $O/frames_cmp6.txt 2>&1; python tools/lib_frames.py old new >> $O/frames_cmp6.txt 2>&1
for v in st16m32w6; do DTSIM_LIB=$L/libdtsim_$v.so python tools/lib_frames.py $v >> $O/frames_cmp6.txt 2>&1; python tools/lib_frames.py old $v >> $O/frames_cmp6.txt 2>&1; done
bash tools/ab.sh default st12 st16m32 st16w6 st16m32w6 old > $O/ab6.txt 2>&1
grep -v amdgpu.ids $O/frames_cmp6.txt; cat $O/ab6.txt
python -m pytest tests -m gpu -x -q > $O/pytest6.txt 2>&1; tail -5 $O/pytest6.txt
