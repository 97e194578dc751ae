#!/bin/bash
# round 5: the GPU suite + smoke() on the current tree
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/pytest_b.txt 2>&1; tail -5 $O/pytest_b.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
