#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
python -m pytest tests/test_gpu_physics.py tests/test_gpu_device_reset.py -m gpu -x -q 2>&1 | tail -3
: > $O/lanes15.txt
for mapn in junction_map loop_pedestrians small_loop; do for L in 1 2 4 8; do
  for lib in default nocoop; do
    if [ $lib = default ]; then unset DTSIM_LIB; else export DTSIM_LIB=$PWD/gym-duckietown_amd/lib/libdtsim_$lib.so; fi
    echo -n "$lib " >> $O/lanes15.txt; MAP=$mapn N=4096 F=32 K=20 DTSIM_STEP_LANES=$L python tools/time_step.py 2>&1 | tail -1 >> $O/lanes15.txt
  done
done; done
cat $O/lanes15.txt
