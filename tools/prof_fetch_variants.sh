#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the headline render pass (small_loop, N = 4096) for ablation variants of the library:
#   bash tools/prof_fetch_variants.sh "" _oneblock _norslv      (suffixes of gym-duckietown_amd/lib/libdtsim<suffix>.so)
# attributes the L2 fills of the pass: record gathers (oneblock: every gather from one L2-resident block) and the exact path (norslv).
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_fetchvar
mkdir -p $OUT
cd /tmp
for v in "$@"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    P=${c}${v}
    DTSIM_LIB=$GRAFT_REPO_ROOT/gym-duckietown_amd/lib/libdtsim$v.so MAP=small_loop N=4096 K=3 timeout 150 rocprofv3 --pmc $c -d $OUT/$P -o $P -- python $GRAFT_REPO_ROOT/tools/time_render.py > $OUT/$P.log 2>&1
  done
done
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py "$OUT/*/*.db" > $OUT/summary.txt 2>&1
grep -E "^==|k_raster|SIZE" $OUT/summary.txt | grep -B1 -A1 "k_raster" | head -60
