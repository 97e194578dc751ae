#!/bin/bash
# build an ablation variant of libdtsim.so: tools/build_variant.sh NAME "-DFOO=1 ..."  ->  lib/libdtsim_NAME.so
set -e
D=/root/repo/gym-duckietown_amd
python $D/build.py >/dev/null 2>&1
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-bitwise-instead-of-logical -I/root/repo/include"
/opt/rocm/bin/hipcc $F -ffp-contract=fast $2 -c $D/csrc/render.hip -o /tmp/render_$1.o 2>/dev/null
/opt/rocm/bin/hipcc $F $2 -c $D/csrc/observe.hip -o /tmp/observe_$1.o 2>/dev/null
/opt/rocm/bin/hipcc $F $2 -c $D/csrc/dtsim_api.hip -o /tmp/api_$1.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/lib/libdtsim_$1.so $D/lib/physics.o /tmp/render_$1.o /tmp/observe_$1.o /tmp/api_$1.o
echo built $D/lib/libdtsim_$1.so
