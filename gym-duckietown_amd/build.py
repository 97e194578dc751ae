"""Build libdtsim.so (HIP, gfx950) in-tree.

    python gym-duckietown_amd/build.py [--force]

hipcc cross-compiles without a GPU.  Output: gym-duckietown_amd/lib/libdtsim.so
(git-ignored, travels to the GPU box with the gpurun snapshot).
physics.hip is compiled with -ffp-contract=off (separately rounded f64 ops, in the
reference's operation order); render.hip with contraction allowed (speed, f32).
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libdtsim.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

UNITS = [
    ("physics.hip", ["-ffp-contract=off"]),
    ("render.hip", ["-ffp-contract=fast"]),
    ("observe.hip", []),
    ("dtsim_api.hip", []),
]
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-bitwise-instead-of-logical",
          "-I" + os.path.join(HERE, "..", "include")]


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    headers.append(os.path.join(HERE, "..", "include", "dtsim.h"))
    objs = []
    for name, flags in UNITS:
        src = os.path.join(CSRC, name)
        obj = os.path.join(LIBDIR, name.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer([src] + headers, obj):
            cmd = [HIPCC] + COMMON + flags + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _newer(objs, LIB):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
