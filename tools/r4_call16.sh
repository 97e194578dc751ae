#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
bash tools/ab_cfg.sh "c4" default dr7 drnobs > $O/ab16.txt 2>&1; cat $O/ab16.txt
bash tools/prof_bench_short.sh r04 c3 > $O/prof_c3.txt 2>&1
bash tools/prof_bench_short.sh r04c4 c4 > $O/prof_c4.txt 2>&1
bash tools/prof_bench_short.sh r04c5 c5 > $O/prof_c5.txt 2>&1
for t in r04 r04c4 r04c5; do rm -rf gpurun_out/prof_bench_$t/*/*.db.tmp; du -sh gpurun_out/prof_bench_$t; done
python tools/make_traffic_json.py gpurun_out/prof_bench_r04 4096 "x" - - c3 | tail -1 | cut -c1-300
