#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_baseline_configs.py tests/test_gpu_fullsize.py tests/test_gpu_facade.py -m gpu -x -q 2>&1 | tail -15
timeout 600 bash tools/ab_cfg.sh "c5 c4" default > $O/ab17.txt 2>&1
DTSIM_RESOLVE_OBJ_OLD=1 timeout 600 bash tools/ab_cfg.sh "c5 c4" default >> $O/ab17.txt 2>&1
cat $O/ab17.txt
cd /tmp; export TMPDIR=/tmp
for c in c5 c4; do D=/tmp/tr_$c; rm -rf $D; mkdir -p $D; timeout 300 rocprofv3 --kernel-trace --stats -d $D -o t -- python $GRAFT_REPO_ROOT/bench.py --config $c --steps 10 --warmup 3 --windows 1 --cpu-steps 0 --no-gather > $D/log 2>&1; python $GRAFT_REPO_ROOT/tools/rocpd_summary.py "$D/*.db" | grep calls | head -5 | cut -c1-150; done
