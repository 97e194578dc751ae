#!/bin/bash
# render parts (DTSIM_RENDER_PARTS): frames identical to the one-part launch, then bench timing per part count
cd $GRAFT_REPO_ROOT
export N=1024 STEPS=6
for cfg in "loop_only_duckies,small_loop_only_duckies 0" "loop_pedestrians 1"; do
  set -- $cfg
  export MAP=$1 DR=$2
  DTSIM_RENDER_PARTS=1 timeout 300 python tools/lib_frames.py p1 2>&1 | tail -1
  DTSIM_RENDER_PARTS=4 timeout 300 python tools/lib_frames.py p4 2>&1 | tail -1
  python tools/lib_frames.py p1 p4
done
for r in 1 2; do
for c in c5 c4; do
  for p in 1 2 4 8; do
    echo -n "$c parts=$p: "
    DTSIM_RENDER_PARTS=$p timeout 300 python bench.py --config $c --steps 20 --warmup 3 --cpu-steps 0 --no-gather --windows 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms/step  %s' % (d['ms_per_step'], d.get('windows_ms')))"
  done
done
done
