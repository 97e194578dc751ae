#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
bash tools/ab_cfg.sh "c3 c4 c5" default obj4 m0w5 old > $O/ab7.txt 2>&1
cat $O/ab7.txt
python -m pytest tests -m gpu -x -q > $O/pytest7.txt 2>&1; tail -3 $O/pytest7.txt
