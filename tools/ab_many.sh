#!/bin/bash
# A/B of several builds with many short alternating processes (the spread between PROCESSES on one box is +-2 %: one run per build
# decides nothing).  tools/ab_many.sh ROUNDS "c3|c4|c5" NAME...   ("default" = lib/libdtsim.so); prints min / median / mean of the
# per-step milliseconds bench.py reports, per build.
R=$1; CFG=$2; shift; shift
T=/tmp/abmany_$$; : > $T
for r in $(seq 1 $R); do
  for v in "$@"; do
    if [ $v = default ]; then unset DTSIM_LIB; else export DTSIM_LIB=$PWD/gym-duckietown_amd/lib/libdtsim_$v.so; fi
    ms=$(python bench.py --config $CFG --steps 20 --warmup 3 --windows 3 --cpu-steps 0 --no-gather 2>/dev/null | tail -1 | python -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$v $ms" >> $T
  done
done
python - $T "$CFG" <<'PY'
import sys, collections, statistics
d = collections.defaultdict(list)
for l in open(sys.argv[1]):
    v, ms = l.split(); d[v].append(float(ms))
for v, xs in d.items():
    print(f"{sys.argv[2]} {v:12s} n={len(xs)}  min {min(xs):.3f}  median {statistics.median(xs):.3f}  mean {statistics.mean(xs):.3f}  max {max(xs):.3f} ms/step")
PY
