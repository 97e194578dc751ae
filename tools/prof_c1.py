import cProfile, pstats, sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gym-duckietown_amd"))
import numpy as np
from gym_duckietown.envs import DuckietownEnv
env = DuckietownEnv(map_name="small_loop", camera_width=84, camera_height=84, domain_rand=False, seed=1, max_steps=10**9)
env.reset()
acts = np.random.default_rng(0).uniform(0.2, 0.8, (3000, 2))
for a in acts[:200]: env.step(a)
pr = cProfile.Profile(); pr.enable()
for a in acts[200:2200]:
    o, r, d, i = env.step(a)
    if d: env.reset()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(22)
