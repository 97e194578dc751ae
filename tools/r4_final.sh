#!/bin/bash
# round 4, closing run: the whole GPU suite, every BASELINE config through bench.py, the default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_final.txt 2>&1; tail -4 $O/pytest_final.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/run_configs.sh r04 > $O/configs_final.txt 2>&1; grep -E "env-steps|calls" $O/configs_final.txt | cut -c1-170
python bench.py > $O/bench_default.txt 2>&1; tail -1 $O/bench_default.txt | cut -c1-600
