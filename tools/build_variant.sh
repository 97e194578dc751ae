#!/bin/bash
# build an ablation variant of libdtsim.so: tools/build_variant.sh NAME "-DDTSIM_EXP=1 ..."  ->  lib/libdtsim_NAME.so
set -e
D=/root/repo/gym-duckietown_amd
python $D/build.py >/dev/null 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -I/root/repo/include -ffp-contract=fast $2 -c $D/csrc/render.hip -o /tmp/render_$1.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/lib/libdtsim_$1.so $D/lib/physics.o /tmp/render_$1.o $D/lib/dtsim_api.o
echo built $D/lib/libdtsim_$1.so
