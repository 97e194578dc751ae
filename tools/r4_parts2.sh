#!/bin/bash
# render parts, serial (no overlap): what the split alone costs, per kernel
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp DTSIM_RENDER_PARTS_SERIAL=1
for c in c5 c4; do
  for p in 1 4; do
    echo "== $c parts=$p serial"
    OUT=$PWD/gpurun_out/parts_${c}_$p
    (cd /tmp; DTSIM_RENDER_PARTS=$p timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o t -- python $GRAFT_REPO_ROOT/bench.py --config $c --steps 10 --warmup 3 --cpu-steps 0 --no-gather --windows 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms/step' % d['ms_per_step'])")
    python tools/rocpd_summary.py "$OUT/*/*.db" 2>&1 | grep -i "resolve\|raster\|EnvD\|PixTab\|obj_setup" | head -8
  done
done
