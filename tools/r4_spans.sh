#!/bin/bash
cd $GRAFT_REPO_ROOT
export DTSIM_LIB=$PWD/gym-duckietown_amd/lib/libdtsim_spans.so DTSIM_WAVE_SPANS=/tmp/spans.bin
for n in 4096 1024 256; do timeout 300 python tools/wave_spans.py c4 $n 2>&1 | grep -A8 "k_resolve:"; done
