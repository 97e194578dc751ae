#!/bin/bash
# per-kernel durations (rocprofv3 kernel trace) of several builds: tools/ab_trace.sh NAME...   ("default" = lib/libdtsim.so)
export TMPDIR=/tmp
for v in "$@"; do
  if [ $v = default ]; then unset DTSIM_LIB; else export DTSIM_LIB=$PWD/gym-duckietown_amd/lib/libdtsim_$v.so; fi
  OUT=/tmp/abt_$v; rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && K=${K:-6} N=${N:-4096} rocprofv3 --kernel-trace --stats -d $OUT -o t -- python $GRAFT_REPO_ROOT/tools/time_render.py > $OUT/log 2>&1)
  echo "== $v"; python tools/rocpd_summary.py "$OUT/*.db" | grep calls | sed 's/^ *//' | cut -c1-140
done
