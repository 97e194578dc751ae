#!/bin/bash
cd $GRAFT_REPO_ROOT
export N=1024 STEPS=6
for cfg in "loop_only_duckies,small_loop_only_duckies 0" "loop_pedestrians 1"; do
  set -- $cfg
  export MAP=$1 DR=$2
  DTSIM_LIB=$PWD/gym-duckietown_amd/lib/libdtsim_res1.so timeout 300 python tools/lib_frames.py d1 2>&1 | tail -1
  timeout 300 python tools/lib_frames.py dn 2>&1 | tail -1
  python tools/lib_frames.py d1 dn
  DTSIM_RENDER_PARTS=2 timeout 300 python tools/lib_frames.py dp 2>&1 | tail -1
  python tools/lib_frames.py d1 dp
done
bash tools/ab_cfg.sh "c5 c4" res1 noheavy default 2>&1
export DTSIM_WAVE_SPANS=/tmp/spans.bin
for c in c5 c4; do DTSIM_LIB=$PWD/gym-duckietown_amd/lib/libdtsim_spans.so timeout 300 python tools/wave_spans.py $c 4096 2>&1 | grep -A8 "k_resolve"; done
for c in c5 c4; do echo -n "$c parts=2: "; DTSIM_RENDER_PARTS=2 timeout 300 python bench.py --config $c --steps 20 --warmup 3 --cpu-steps 0 --no-gather --windows 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms/step' % (d['ms_per_step']))"; done
