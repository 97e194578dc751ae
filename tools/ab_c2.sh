#!/bin/bash
# C2 (physics only) throughput for several builds: tools/ab_c2.sh NAME... ("default" = lib/libdtsim.so)
for v in "$@"; do
  if [ $v = default ]; then unset DTSIM_LIB; else export DTSIM_LIB=$PWD/gym-duckietown_amd/lib/libdtsim_$v.so; fi
  echo -n "$v: "; python bench.py --config c2 --envs ${N:-1048576} --fuse 1 --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e9,3),'G env-steps/s', round(d['ms_per_step'],4),'ms')"
done
