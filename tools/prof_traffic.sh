#!/bin/bash
# HBM-side traffic of the render kernels around tools/time_render.py: FETCH_SIZE and WRITE_SIZE each in their own pass
# (MI355X_MICROARCH.md: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2).   bash tools/prof_traffic.sh <tag>   (N env var)
TAG=${1:-t}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
K=3 N=${N:-4096} rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- python $GRAFT_REPO_ROOT/tools/time_render.py > $OUT/fetch.log 2>&1
K=3 N=${N:-4096} rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/write -o write -- python $GRAFT_REPO_ROOT/tools/time_render.py > $OUT/write.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py "$OUT/*/*.db" > $OUT/summary.txt 2>&1
grep -E "\[pmc\]|FETCH_SIZE|WRITE_SIZE|TCC_" $OUT/summary.txt | grep -A4 -E "k_raster|k_resolve|PixTabEPKj|EnvCamPKtPKi"
