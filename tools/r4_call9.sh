#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest9.txt 2>&1; tail -4 $O/pytest9.txt
python bench.py --steps 20 --warmup 5 > $O/bench9.txt 2>&1; tail -1 $O/bench9.txt | cut -c1-1500
