#!/bin/bash
# round 5, closing run: the whole GPU suite, smoke(), every BASELINE config through bench.py (+ kernel traces of C4 / C5), the rocprofv3 passes
# of the bench command for C3 / C4 / C5 (kernel trace + stats, FETCH_SIZE, WRITE_SIZE, two SQ passes: each in its own run), the default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5f; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_final.txt 2>&1; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|Gloo" $O/pytest_final.txt | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/run_configs.sh r05 > $O/configs_final.txt 2>&1; grep -E "env-steps|calls" $O/configs_final.txt | cut -c1-170
bash tools/prof_bench_short.sh r05 c3 > $O/prof_c3.log 2>&1
bash tools/prof_bench_short.sh r05c4 c4 > $O/prof_c4.log 2>&1
bash tools/prof_bench_short.sh r05c5 c5 > $O/prof_c5.log 2>&1
python bench.py > $O/bench_default.txt 2>&1; tail -1 $O/bench_default.txt | cut -c1-400
