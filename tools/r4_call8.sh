#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
python -m pytest tests/test_gpu_baseline_configs.py -m gpu -x -q -s > $O/pytest8.txt 2>&1; tail -8 $O/pytest8.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke8.txt 2>&1; tail -3 $O/smoke8.txt
