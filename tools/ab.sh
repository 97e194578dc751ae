#!/bin/bash
# timing of several builds of libdtsim.so on the same GPU box: tools/ab.sh NAME... ("default" = lib/libdtsim.so)
for r in 1 2; do
  for v in "$@"; do
    if [ $v = default ]; then unset DTSIM_LIB; else export DTSIM_LIB=$PWD/gym-duckietown_amd/lib/libdtsim_$v.so; fi
    echo -n "$v: "; N=${N:-4096} K=20 python tools/time_render.py 2>&1 | tail -1 | sed 's/.*event/event/'
  done
done
