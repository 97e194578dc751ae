#!/bin/bash
cd $GRAFT_REPO_ROOT
export N=1024 STEPS=6
for cfg in "loop_only_duckies,small_loop_only_duckies 0" "loop_pedestrians 1"; do
  set -- $cfg
  export MAP=$1 DR=$2
  timeout 300 python tools/lib_frames.py d8 2>&1 | tail -1
  DTSIM_LIB=$PWD/gym-duckietown_amd/lib/libdtsim_res1.so timeout 300 python tools/lib_frames.py d1 2>&1 | tail -1
  python tools/lib_frames.py d8 d1
done
bash tools/ab_cfg.sh "c5 c4" default res1 2>&1 | head -4
export DTSIM_WAVE_SPANS=/tmp/spans.bin
for v in spans1 spans2; do
for c in c5 c4; do echo "== $v"; DTSIM_LIB=$PWD/gym-duckietown_amd/lib/libdtsim_$v.so timeout 300 python tools/wave_spans.py $c 4096 2>&1 | grep -A7 "k_resolve_obj"; done
done
