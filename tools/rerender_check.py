"""One batch rendered several times from one state; do the frames repeat byte for byte?  (tests/test_gpu_render.py holds the kernels to it; this prints WHERE they differ.)  python tools/rerender_check.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gym-duckietown_amd"))
from dtsim.batched import BatchedSimulator
from dtsim import _ffi
W, H, N = 640, 480, 512
sim = BatchedSimulator("small_loop", N, camera_width=W, camera_height=H, distortion=True, domain_rand=False, seed=5, max_steps=100000)
acts = np.random.default_rng(9).uniform(0.2, 0.9, (4, N, 2)).astype(np.float32)
sim.step(acts, n_steps=4)
fr = []
for i in range(4):
    sim.render(); fr.append(sim.frames_host().copy())
for i in range(1, 4):
    d = (fr[0] != fr[i]).any(axis=-1)
    print("render 0 vs", i, "pixels that differ:", int(d.sum()))
    e, y, x = [v[:8] for v in np.nonzero(d)]
    for j in range(len(e)):
        print("   env", e[j], "y", y[j], "x", x[j], fr[0][e[j], y[j], x[j]], fr[i][e[j], y[j], x[j]])
