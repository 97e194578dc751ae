"""k_raster_v3 against k_raster_q on the same states: frames must agree (same integer filter, same exact path).
Run on the GPU box: python tools/v3_check.py  (env MAP, N, W, H, DIST)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gym-duckietown_amd"))
import numpy as np
from dtsim import BatchedSimulator

N = int(os.environ.get("N", "256"))
W, H = int(os.environ.get("W", "640")), int(os.environ.get("H", "480"))
dist = os.environ.get("DIST", "1") == "1"
maps = os.environ.get("MAP", "small_loop").split(",")
mc = len(maps) > 1


def frames(old):
    os.environ["DTSIM_RASTER_OLD"] = "1" if old else "0"
    sim = BatchedSimulator(maps if mc else maps[0], N, seed=7, distortion=dist, domain_rand=False, camera_width=W, camera_height=H,
                           **({"map_cycle": True} if mc else {}))
    rng = np.random.default_rng(3)
    for _ in range(int(os.environ.get("STEPS", "8"))):
        sim.step(rng.uniform(0.2, 0.9, (N, 2)).astype(np.float32))
    sim.render()
    f = sim.frames_host().copy()
    sim.close()
    return f


a, b = frames(True), frames(False)
d = np.abs(a.astype(np.int16) - b.astype(np.int16)).max(axis=-1)
nz = int((d > 0).sum())
print(f"maps={maps} N={N} {W}x{H} dist={dist}: differing pixels {nz} of {d.size} ({100.0 * nz / d.size:.5f} %), max |diff| {int(d.max())}, "
      f"> 1: {int((d > 1).sum())}")
if nz:
    e, y, x = np.argwhere(d > 0)[0]
    print("first difference: env", e, "pixel", (x, y), "old", a[e, y, x], "v3", b[e, y, x])
    per_env = (d > 0).reshape(N, -1).sum(axis=1)
    print("envs with differences:", int((per_env > 0).sum()), "worst env", int(per_env.argmax()), int(per_env.max()))
    ys, xs = np.nonzero((d > 1).any(axis=0)) if (d > 1).any() else np.nonzero((d > 0).any(axis=0))
    print("bbox of differing pixels: x", xs.min(), xs.max(), "y", ys.min(), ys.max())
assert (d > 1).mean() < 1e-5, "k_raster_v3 disagrees with k_raster_q"
print("v3_check ok")
