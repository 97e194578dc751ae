#!/bin/bash
cd $GRAFT_REPO_ROOT
export DTSIM_LIB=$PWD/gym-duckietown_amd/lib/libdtsim_spans.so DTSIM_WAVE_SPANS=/tmp/spans.bin
timeout 300 python tools/wave_spans.py c4 4096 2>&1 | grep "k_resolve pixels"
DTSIM_DEBUG_QUEUE=1 timeout 300 python tools/wave_spans.py c4 1024 2>&1 | grep "exact-path\|k_resolve pixels" | tail -2
