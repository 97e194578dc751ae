#!/bin/bash
cd $GRAFT_REPO_ROOT
export DTSIM_LIB=$PWD/gym-duckietown_amd/lib/libdtsim_spans.so DTSIM_WAVE_SPANS=/tmp/spans.bin
timeout 300 python tools/raster_spans.py c3 4096 2>&1 | tail -12
timeout 300 python tools/raster_spans.py c5 4096 2>&1 | tail -12
