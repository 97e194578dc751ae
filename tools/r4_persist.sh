#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python bench.py --config c3 --steps 20 --warmup 3 --cpu-steps 0 --no-gather --windows 3 2>/tmp/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms/step' % d['ms_per_step'])"; grep "resident_blocks" /tmp/err.log | head -2; }
echo -n "non-persistent: "; run
export DTSIM_LIB=$PWD/gym-duckietown_amd/lib/libdtsim_p1.so DTSIM_DEBUG_RESIDENT=1
echo -n "persistent, occupancy API: "; run
for g in 1024 1280 1536 1792 2048 3072; do echo -n "persistent, grid $g: "; DTSIM_V3_GRID=$g run; done
