"""The REFERENCE's own step loop -- DuckietownEnv.step: physics + _render_img on OpenGL + reward -- timed on this host's cores, by
the protocol of the reference's benchmark.py:9-48 (seed, reset, then steps for a fixed wall time, reset on done).

Test infrastructure (like everything under oracle/): runs the unmodified /root/reference package through oracle/gl/refgl.py (pyglet
shim over Mesa llvmpipe -- the reference CI's renderer; duckietown_world dynamics restated by oracle/sim.py:DynamicsDB18; cv2 absent,
so distortion=False), so it only works in the build container.  bench.py reports the RECORDED numbers
(profiles/reference_render_timings.json, written by this script) next to what it measures live on the bench box with the travelling
restatement of the same GL call stream (oracle/gl/glport.py), which this script times here as well -- same map, same size, same cores --
so that the two can be compared where both run.

    python -W ignore oracle/time_reference_render.py        # writes profiles/reference_render_timings.json
"""
from __future__ import annotations

import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gym-duckietown_amd"))

import numpy as np  # noqa: E402

W, H = 640, 480
SECONDS = 5.0


def reference_loop(map_name, domain_rand, seconds=SECONDS, tree="t256"):
    from oracle.gl import asset_trees, refgl
    t0 = time.time()
    env, _ns = refgl.make_simulator(map_name, asset_trees.roots(tree), env_class="DuckietownEnv", domain_rand=domain_rand, max_steps=20000,
                                    camera_width=W, camera_height=H, seed=0, distortion=False)
    env.reset()
    load_ms = 1000 * (time.time() - t0)
    t0 = time.time()
    for _ in range(20):
        env.reset()
    reset_ms = 1000 * (time.time() - t0) / 20
    rng = np.random.default_rng(1234)
    n, t0 = 0, time.time()
    t_render = 0.0
    while time.time() - t0 < seconds:
        _obs, _r, done, _info = env.step(rng.uniform(-1, 1, 2))       # (vel, steer) as bench.py draws them; benchmark.py itself creeps at 0.01
        if done:
            env.reset()
        n += 1
    dt = time.time() - t0
    t1 = time.time()
    for _ in range(50):
        env.render_obs()
    t_render = (time.time() - t1) / 50
    return dict(env_steps_per_s=n / dt, frame_ms=1000 * dt / n, render_obs_ms=1000 * t_render, load_ms=load_ms, reset_ms=reset_ms, steps=n)


def main():
    from oracle.gl import glshim, refgl
    assert refgl.available(), "needs /root/reference and Mesa's swrast_dri.so (the build container)"
    import bench
    out = {"what": "the reference's DuckietownEnv.step loop (unmodified /root/reference/src/gym_duckietown through oracle/gl/refgl.py: physics + "
                   "_render_img on OpenGL + lane pose / collision / reward), benchmark.py:9-48's protocol with bench.py's random (vel, steer) actions, "
                   f"{W}x{H}, distortion off (cv2 is absent), {SECONDS:.0f} s per case",
           "where": f"build container (no GPU), 1 core (LP_NUM_THREADS=1): {platform.machine()}", "renderer": glshim.renderer(), "numpy": np.__version__,
           "cases": {}}
    for name, (m, dr) in {"small_loop, domain_rand off": ("small_loop", False), "small_loop, domain_rand on (the gym.make default)": ("small_loop", True),
                          "loop_only_duckies, domain_rand off": ("loop_only_duckies", False)}.items():
        out["cases"][name] = {k: round(v, 3) if isinstance(v, float) else v for k, v in reference_loop(m, dr).items()}
        print(name, out["cases"][name], flush=True)
    n = 150
    dt = bench._glport_env_steps((n, 1000, False))
    dtf = bench._glport_env_steps((n, 1000, True))
    out["glport_same_host"] = {"what": "oracle/gl/glport.py + oracle/sim.py (bench.py's cpu_baseline leg) on small_loop, domain_rand off, this host, 1 core",
                               "env_steps_per_s": round(n / dt, 3), "env_steps_per_s_with_fisheye_remap": round(n / dtf, 3), "steps": n}
    print(out["glport_same_host"])
    with open(os.path.join(ROOT, "profiles", "reference_render_timings.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
