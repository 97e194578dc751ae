#!/bin/bash
# bench.py timing of several builds for the given configs on one box: tools/ab_cfg.sh "c3 c4 c5" default NAME ...
CFGS=$1; shift
for r in 1 2; do
  for c in $CFGS; do
    for v in "$@"; do
      if [ $v = default ]; then unset DTSIM_LIB; else export DTSIM_LIB=$PWD/gym-duckietown_amd/lib/libdtsim_$v.so; fi
      echo -n "$c $v: "; python bench.py --config $c --steps 20 --warmup 3 --cpu-steps 0 --no-gather 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms/step  %.3f M env-steps/s  kernel median %s' % (d['ms_per_step'], d['value']/1e6, d.get('roofline',{}).get('kernel_ms',{}).get('median') if isinstance(d.get('roofline',{}).get('kernel_ms'),dict) else d.get('roofline',{}).get('kernel_ms')))"
    done
  done
done
