"""Time dtsim_observe (640x480 -> 160x120, CHW float32) for N envs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gym-duckietown_amd"))
from dtsim import BatchedSimulator, _ffi
N = int(os.environ.get("N", "4096"))
sim = BatchedSimulator("small_loop", N, seed=1, distortion=True, domain_rand=False, profile=True)
sim.render()
for mode in [dict(chw=False, normalize=False), dict(chw=True, normalize=True)]:
    for _ in range(3):
        sim.observe(120, 160, **mode)
    sim.sync(); sim.profile_read(_ffi.KERNEL_OBSERVE)
    for _ in range(10):
        sim.observe(120, 160, **mode)
    sim.sync()
    n, ms = sim.profile_read(_ffi.KERNEL_OBSERVE)
    gb = N * 640 * 480 * 3 / 1e9
    print(f"observe {mode}: {ms/n:.3f} ms per {N} envs -> {gb/(ms/n*1e-3)/1e3:.2f} TB/s of frame bytes read")
