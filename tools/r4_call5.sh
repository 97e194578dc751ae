#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
L=$PWD/gym-duckietown_amd/lib
python tools/lib_frames.py new > $O/frames_cmp5.txt 2>&1
for v in latew xp0 w6; do DTSIM_LIB=$L/libdtsim_$v.so python tools/lib_frames.py $v >> $O/frames_cmp5.txt 2>&1; python tools/lib_frames.py new $v >> $O/frames_cmp5.txt 2>&1; done
bash tools/ab.sh default latew dumpst w6 xp0 latew6 m32 > $O/ab5.txt 2>&1
bash tools/prof_variants_pmc.sh r4_var default nostore nores noresslow noload dumpst > $O/pmc5.txt 2>&1
grep -v amdgpu.ids $O/frames_cmp5.txt; cat $O/ab5.txt; cat $O/pmc5.txt
