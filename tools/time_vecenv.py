"""Throughput of the learner-facing loop (DuckietownVecEnv): step + device reset + render + 160x120 CHW float obs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gym-duckietown_amd"))
import torch
from dtsim import DuckietownVecEnv
N = int(os.environ.get("N", "4096"))
env = DuckietownVecEnv("small_loop", N, obs_shape=(120, 160), seed=0, domain_rand=False, distortion=True)
obs = env.reset()
a = torch.rand((N, 2), device="cuda") * 2 - 1
for _ in range(5):
    env.step(a)
torch.cuda.synchronize()
K = 30
t = time.perf_counter()
for _ in range(K):
    obs, r, d, info = env.step(a)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / K
print(f"VecEnv N={N}: {dt*1e3:.3f} ms per step -> {N/dt/1e6:.3f} M env-steps/s with [N,3,120,160] float32 observations")
