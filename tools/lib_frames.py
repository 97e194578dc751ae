"""Frames of one build of libdtsim.so (DTSIM_LIB) after a few steps -> /tmp/frames_<tag>.npy; with two tags given, compares them.
    DTSIM_LIB=.../libdtsim_old.so python tools/lib_frames.py old ; python tools/lib_frames.py new ; python tools/lib_frames.py old new
Env: MAP (comma list = MultiMap), N, DR, DIST, STEPS."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gym-duckietown_amd"))
import numpy as np

if len(sys.argv) == 3:
    a, b = (np.load(f"/tmp/frames_{t}.npy") for t in sys.argv[1:3])
    d = np.abs(a.astype(np.int16) - b.astype(np.int16)).max(axis=-1)
    print(f"{sys.argv[1]} vs {sys.argv[2]}: {a.shape}, differing pixels {int((d > 0).sum())} of {d.size}, max |diff| {int(d.max())}, > 1: {int((d > 1).sum())}")
    sys.exit(0)
from dtsim import BatchedSimulator
N = int(os.environ.get("N", "512"))
maps = os.environ.get("MAP", "small_loop").split(",")
mc = len(maps) > 1
sim = BatchedSimulator(maps if mc else maps[0], N, seed=7, distortion=os.environ.get("DIST", "1") == "1", domain_rand=os.environ.get("DR", "0") == "1",
                       **({"map_cycle": True} if mc else {}))
rng = np.random.default_rng(3)
for _ in range(int(os.environ.get("STEPS", "8"))):
    sim.step(rng.uniform(0.2, 0.9, (N, 2)).astype(np.float32))
sim.render()
f = sim.frames_host().copy()
np.save(f"/tmp/frames_{sys.argv[1]}.npy", f)
print(sys.argv[1], f.shape, "mean", float(f.mean()))
