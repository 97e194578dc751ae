#!/bin/bash
for v in "$@"; do
  if [ $v = default ]; then unset DTSIM_LIB; else export DTSIM_LIB=$PWD/gym-duckietown_amd/lib/libdtsim_$v.so; fi
  echo -n "$v: "; python tools/time_observe.py 2>&1 | tail -2 | head -1
done
