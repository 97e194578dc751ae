"""Render a few frames on the GPU and save PNGs + timing to gpurun_out/ (eyeball check)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gym-duckietown_amd"))
import numpy as np
from PIL import Image
from dtsim import BatchedSimulator, _ffi

out = os.path.join(ROOT, "gpurun_out", "frames")
os.makedirs(out, exist_ok=True)
for name, kw in [("small_loop", dict(distortion=False, domain_rand=False)),
                 ("small_loop", dict(distortion=True, domain_rand=False)),
                 ("loop_only_duckies", dict(distortion=False, domain_rand=True))]:
    sim = BatchedSimulator(name, 4, seed=1, **kw)
    sim.render()
    fr = sim.frames_host()
    tag = f"{name}_d{int(kw['distortion'])}_r{int(kw['domain_rand'])}"
    for e in range(4):
        Image.fromarray(fr[e]).save(os.path.join(out, f"{tag}_{e}.png"))
    print(tag, fr.shape, fr.mean(), sim.read(_ffi.FIELD_POS)[0], sim.read(_ffi.FIELD_ANGLE)[0])
    sim.close()

# quick timing
N = int(os.environ.get("N", "512"))
sim = BatchedSimulator("small_loop", N, seed=1, distortion=True, domain_rand=False, profile=True)
for _ in range(3):
    sim.render()
sim.sync()
sim.profile_read(_ffi.KERNEL_RENDER)
t = time.time()
K = 10
for _ in range(K):
    sim.render()
sim.sync()
dt = (time.time() - t) / K
n, ms = sim.profile_read(_ffi.KERNEL_RENDER)
print(f"N={N} render wall {dt*1e3:.3f} ms/step, event {ms/n:.3f} ms -> {N/dt/1e6:.3f} M env-steps/s, "
      f"{N*640*480*3/(ms/n*1e-3)/1e12:.3f} TB/s")
