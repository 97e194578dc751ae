#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest22.txt 2>&1; tail -3 $O/pytest22.txt
bash tools/prof_bench_short.sh r04s c3 > $O/prof22.txt 2>&1
bash tools/prof_bench_short.sh r04sc4 c4 >> $O/prof22.txt 2>&1
bash tools/prof_bench_short.sh r04sc5 c5 >> $O/prof22.txt 2>&1
bash tools/run_configs.sh r04b > $O/configs22.txt 2>&1; grep -E "env-steps" $O/configs22.txt | cut -c1-170
