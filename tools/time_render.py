"""Time dtsim_render for a given N (HIP events around the launches)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gym-duckietown_amd"))
import numpy as np
from dtsim import BatchedSimulator, _ffi
N = int(os.environ.get("N", "1024"))
dist = os.environ.get("DIST", "1") == "1"
dr = os.environ.get("DR", "0") == "1"
sim = BatchedSimulator(os.environ.get("MAP", "small_loop"), N, seed=1, distortion=dist, domain_rand=dr, profile=True)
for _ in range(3):
    sim.render()
sim.sync(); sim.profile_read(_ffi.KERNEL_RENDER)
K = int(os.environ.get("K", "10"))
t = time.time()
for _ in range(K):
    sim.render()
sim.sync()
dt = (time.time() - t) / K
n, ms = sim.profile_read(_ffi.KERNEL_RENDER)
print(f"N={N} dist={dist} dr={dr} msaa_off={os.environ.get('DTSIM_RASTER_NO_MSAA','0')}: wall {dt*1e3:.3f} ms, event {ms/n:.3f} ms "
      f"-> {N/(ms/n*1e-3)/1e6:.3f} M env-steps/s, {N*640*480*3/(ms/n*1e-3)/1e12:.3f} TB/s ({N*640*480*3/(ms/n*1e-3)/8e12*100:.1f}% of 8 TB/s)")
