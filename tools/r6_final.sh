#!/bin/bash
# round 6, closing run: the whole GPU suite (with -s: the GL-golden tests print their measured distances), smoke(), every BASELINE config through
# bench.py (+ kernel traces), the rocprofv3 passes of the bench command for C3, the default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6f; mkdir -p $O
python -W ignore -m pytest tests -m gpu -q -s > $O/pytest_final.txt 2>&1; grep -E "vs GL|facade vs|passed|failed" $O/pytest_final.txt | tail -24
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/run_configs.sh r06 > $O/configs_final.txt 2>&1; grep -E "env-steps|calls" $O/configs_final.txt | cut -c1-170
bash tools/prof_bench_short.sh r06 c3 > $O/prof_c3.log 2>&1
python bench.py > $O/bench_default.txt 2>&1; tail -1 $O/bench_default.txt | cut -c1-400
