#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
python -m pytest tests/test_gpu_render.py tests/test_gpu_baseline_configs.py tests/test_gpu_fullsize.py -m gpu -x -q -s > $O/pytest11.txt 2>&1; grep -E "passed|failed|Error|error|worst" $O/pytest11.txt | tail -12
bash tools/ab_cfg.sh "c5 c4" default lobj4 ldr5 > $O/ab11.txt 2>&1
DTSIM_OBJ_LAYERS=0 bash tools/ab_cfg.sh "c5 c4" default >> $O/ab11.txt 2>&1
cat $O/ab11.txt
