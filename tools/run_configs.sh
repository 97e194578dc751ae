#!/bin/bash
# Every BASELINE.json config through bench.py, one JSON line each -> gpurun_out/<tag>_configs.json (copy to profiles/).
#   bash tools/run_configs.sh r02            (on the GPU box; ~3 min)
export TAG=${1:-r02}
OUT=gpurun_out/${TAG}_configs.json
mkdir -p gpurun_out
: > $OUT
python bench.py --config c1 --steps 200 --warmup 20 --cpu-steps 8 2>/dev/null | tail -1 >> $OUT
python bench.py --config c2 --steps 20 --warmup 3 2>/dev/null | tail -1 >> $OUT
python bench.py --config c2 --envs 1048576 --fuse 1 --steps 20 --warmup 3 2>/dev/null | tail -1 >> $OUT
python bench.py --config c3 --steps 30 --warmup 5 --cpu-steps 0 2>/dev/null | tail -1 >> $OUT
python bench.py --config c4 --steps 20 --warmup 3 --cpu-steps 0 2>/dev/null | tail -1 >> $OUT
python bench.py --config c5 --steps 20 --warmup 3 --cpu-steps 0 2>/dev/null | tail -1 >> $OUT
python bench.py --config c3dr --steps 20 --warmup 3 --cpu-steps 0 --no-gather 2>/dev/null | tail -1 >> $OUT   # the gym.make defaults (domain_rand on, no fisheye): context, not a BASELINE config
python - <<'PY'
import json, sys, os
tag = os.environ.get("TAG", "r02")
for l in open(f"gpurun_out/{tag}_configs.json"):
    try:
        d = json.loads(l)
        print(f"{d['metric'][:44]:44s} {d['value']:14.1f} {d['unit']:12s} {d['ms_per_step']:9.3f} ms/step  frac {d['roofline']['frac']:.3f}  | {d['config']['workload'][:60]}")
    except Exception as e:
        print("bad line:", l[:120], e)
PY
# per-kernel durations of the two object configs (rocprofv3 kernel trace of the same bench commands)
export TMPDIR=/tmp
KOUT=$PWD/gpurun_out/${TAG}_configs_kernels.txt
: > $KOUT
for c in c4 c5 c3dr; do
  D=/tmp/cfgtrace_$c; rm -rf $D; mkdir -p $D
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $D -o t -- python $GRAFT_REPO_ROOT/bench.py --config $c --steps 10 --warmup 3 --cpu-steps 0 > $D/log 2>&1)
  echo "== bench.py --config $c --steps 10 --warmup 3 (N = 4096, 640x480 + fisheye): kernels of the timed loop" >> $KOUT
  python tools/rocpd_summary.py "$D/*.db" | grep calls | head -8 | sed 's/^ *//' | cut -c1-150 >> $KOUT
done
cat $KOUT
