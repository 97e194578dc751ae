"""Golden FRAMES from the reference's own render path on real OpenGL (build container only).

    python -m oracle.make_gl_golden            # writes tests/golden/ref_gl_*.npz

The reference's Simulator (constructor, reset(), step(), render_obs(), _render_img()) runs UNMODIFIED through oracle/gl/refgl.py
on Mesa 23.2.1 llvmpipe -- the renderer of the reference's own CI (.circleci/config.yml:10,26).  Every record keeps the frame
and the state that produced it, read back from the reference instance and from GL itself (glGetLightfv: the light position in
eye space, i.e. after GL transformed it by the model-view in effect when reset() called glLightfv, simulator.py:579), so that
oracle/raster.py and the HIP raster can be given the same state (tests/test_gl_golden.py, tests/test_gpu_gl_golden.py).

Assets: tests/golden/assets (procedural OBJ / MTL / PNG tree, tests/golden/make_assets.py) -- tile images 128 x 128 -- and,
for the `t256` cases, the same meshes with the 256 x 256 procedural tile textures of dtsim.assets.make_texture written as PNG
(oracle/gl/asset_trees.py; PNG is lossless, so the product's in-memory fixtures are the same texels).  Distortion is off in
every record (cv2 is absent; the remap is pinned separately, tests/golden/ref_distortion.npz).
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gym-duckietown_amd"))

from oracle.gl import asset_trees, refgl  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def _f(v):
    return float(np.asarray(v, dtype=np.float64).reshape(-1)[0])


def light_eye(ns):
    import ctypes
    gl = ns.gl
    buf = (ctypes.c_float * 4)()
    gl.glGetLightfv(gl.GL_LIGHT0, gl.GL_POSITION, buf)
    return [float(v) for v in buf]


def snapshot(sim, ns, frame):
    """State of one rendered frame, as the reference instance and GL hold it."""
    dr = bool(sim.domain_rand)
    rs = sim.randomization_settings
    objs = sim.objects
    rec = dict(
        frame=np.asarray(frame, dtype=np.uint8),
        pos=np.asarray(sim.cur_pos, dtype=np.float64), angle=float(sim.cur_angle),
        cam_height=_f(sim.cam_height), cam_angle=_f(sim.cam_angle[0]), cam_fov_y=_f(sim.cam_fov_y),
        camera_noise=np.asarray(rs["camera_noise"], dtype=np.float64) if dr else np.zeros(3),
        horizon=np.asarray(sim.horizon_color, dtype=np.float64)[:3], ground=np.asarray(sim.ground_color, dtype=np.float64)[:3],
        light_eye=np.asarray(light_eye(ns), dtype=np.float64),
        light_raw=np.asarray((list(rs["light_pos"]) + [0.0]) if dr else [0.0, 3.0, 0.0, 1.0], dtype=np.float64),   # what reset() handed to glLightfv (a 3-vector under DR: w = 0)
        light_ambient=np.asarray(_gl_light(ns, "GL_AMBIENT"))[:3], light_diffuse=np.asarray(_gl_light(ns, "GL_DIFFUSE"))[:3],
        obj_pos=np.asarray([np.asarray(o.pos, dtype=np.float64) for o in objs]).reshape(len(objs), 3),
        obj_yrot=np.asarray([float(o.y_rot) for o in objs]), obj_visible=np.asarray([bool(o.visible) for o in objs]),
        obj_scale=np.asarray([float(o.scale) for o in objs]),
        obj_pattern=np.asarray([int(getattr(o, "pattern", 0)) for o in objs]),
        step_count=int(sim.step_count),
    )
    return rec


def _gl_light(ns, what):
    import ctypes
    gl = ns.gl
    buf = (ctypes.c_float * 4)()
    gl.glGetLightfv(gl.GL_LIGHT0, getattr(gl, what), buf)
    return [float(v) for v in buf]


def _stack(recs):
    keys = recs[0].keys()
    return {k: np.stack([np.asarray(r[k]) for r in recs]) for k in keys}


def case_reset_poses(map_name, tree, dr, W, H, seeds, segment=False, **kw):
    """One NEW Simulator per seed (a fresh GL context each, as every pyglet Window is): the frame its constructor's reset() ends with.
    segment=True: render_obs(segment=True), the segmentation view (only maps without mesh objects: with one in the scene the reference
    raises TypeError -- load_texture is an lru_cache and objmesh.py:268-292 hands it a LIST as segment_into_color)."""
    recs = []
    for seed in seeds:
        sim, ns = refgl.make_simulator(map_name, asset_trees.roots(tree), domain_rand=dr, seed=seed, camera_width=W, camera_height=H,
                                       max_steps=100000, **kw)
        recs.append(snapshot(sim, ns, sim.render_obs(segment=segment)))
    return recs


def case_placed(map_name, tree, dr, W, H, seed, poses, **kw):
    """One Simulator, the agent placed at the given (x, z, angle) poses -- e.g. in front of the objects of the map."""
    sim, ns = refgl.make_simulator(map_name, asset_trees.roots(tree), domain_rand=dr, seed=seed, camera_width=W, camera_height=H,
                                   max_steps=100000, **kw)
    recs = []
    for x, z, a in poses:
        sim.cur_pos = np.array([x, 0.0, z])
        sim.cur_angle = float(a)
        recs.append(snapshot(sim, ns, sim.render_obs()))
    return recs


def case_second_episode(map_name, tree, dr, W, H, seed, n_steps, n_resets):
    """Drive `n_steps`, reset(), keep the frame: the light of episode 2+ is positioned through the LAST frame's model-view
    (reset() calls glLightfv with whatever is on the MODELVIEW stack, simulator.py:579)."""
    sim, ns = refgl.make_simulator(map_name, asset_trees.roots(tree), domain_rand=dr, seed=seed, camera_width=W, camera_height=H,
                                   max_steps=100000)
    rng = np.random.default_rng(seed + 1000)
    recs = []
    for _ in range(n_resets):
        for _k in range(n_steps):
            _obs, _r, done, _info = sim.step(rng.uniform(0.3, 0.9, 2))
            if done:
                break
        mid = snapshot(sim, ns, sim.render_obs())            # a frame in the middle of an episode, objects stepped
        recs.append(mid)
        sim.reset()
        recs.append(snapshot(sim, ns, sim.render_obs()))
    return recs


def case_views(map_name, tree, W, H, seeds, view):
    """The two other views of _render_img, one new Simulator per seed: view = "top_down" (render(mode="top_down")'s image: the map from above with the
    agent's own mesh drawn at its pose, simulator.py:1786-1798, 1920-1927; W x H = the window size) or "bbox" (draw_bbox=True: the debugging camera
    0.8 m above the robot looking down, :1776-1778, with the collision rectangles as GL_LINE_LOOPs)."""
    recs = []
    for seed in seeds:
        kw = dict(draw_bbox=True, camera_width=W, camera_height=H) if view == "bbox" else {}
        sim, ns = refgl.make_simulator(map_name, asset_trees.roots(tree), domain_rand=False, seed=seed, max_steps=100000, **kw)
        if view == "bbox":
            frame = sim.render_obs()
        else:
            frame = sim._render_img(W, H, sim.multi_fbo_human, sim.final_fbo_human, sim.img_array_human, top_down=True, segment=False)
        recs.append(snapshot(sim, ns, frame))
    return recs


def case_trajectory(map_name, tree, dr, W, H, seed, n_steps, keep):
    """The reference's DuckietownEnv driven for `n_steps` with random (vel, steer) actions: per step the pose, reward and done flag its own step()
    returns (lane pose, collision, reward, done: the reference's code; the DB18 integrator: oracle/sim.py's restatement, see oracle/gl/refgl.py),
    and the observation of the steps in `keep` with the state that produced it.  Record k of the frame arrays is step keep[k] (0 = the reset frame)."""
    env, ns = refgl.make_simulator(map_name, asset_trees.roots(tree), env_class="DuckietownEnv", domain_rand=dr, seed=seed, camera_width=W, camera_height=H,
                                   max_steps=100000, distortion=False)
    rng = np.random.default_rng(seed + 77)
    actions = rng.uniform(-1, 1, (n_steps, 2))
    actions[:, 0] = np.abs(actions[:, 0]) * 0.6 + 0.2            # forward, so that the robot leaves the spawn tile
    recs = [snapshot(env, ns, env.render_obs())] if 0 in keep else []
    traj = dict(pos=[], angle=[], reward=[], done=[], speed=[])
    for t in range(n_steps):
        obs, reward, done, _info = env.step(actions[t])
        traj["pos"].append(np.asarray(env.cur_pos, dtype=np.float64)); traj["angle"].append(float(env.cur_angle))
        traj["reward"].append(float(reward)); traj["done"].append(bool(done)); traj["speed"].append(float(env.speed))
        if (t + 1) in keep:
            recs.append(snapshot(env, ns, obs))
        if done:
            break
    out = recs
    for r in out:
        r["traj_actions"] = actions
        for k_, v in traj.items():
            r["traj_" + k_] = np.asarray(v)
    return out


def town_poses():
    """(x, z, angle) looking at each object of test_town from 0.45 m, as tests/test_gpu_render.py places its envs."""
    import yaml
    with open(os.path.join(asset_trees.ASSETS, "maps", "test_town.yaml")) as f:
        md = yaml.safe_load(f)
    H, ts = len(md["tiles"]), md["tile_size"]
    out = []
    for e, desc in enumerate(md["objects"]):
        ox, oz = desc["pos"][0] * ts, (H - desc["pos"][1]) * ts                      # get_transform (README reading) + weird_from_cartesian
        oz = H * ts - oz                                                             # cartesian y -> simulator z (simulator.py:1638-1652)
        a = 0.7 * e
        out.append((ox - 0.45 * np.cos(a), oz + 0.45 * np.sin(a), a))
    return out


CASES = {
    # name: (builder, kwargs, meta)
    "small_loop_t256_640": (case_reset_poses, dict(map_name="small_loop_only_duckies", tree="t256", dr=False, W=640, H=480, seeds=[1, 2, 3, 4])),
    "small_loop_t256_160": (case_reset_poses, dict(map_name="small_loop_only_duckies", tree="t256", dr=False, W=160, H=120, seeds=list(range(10, 26)))),
    "small_loop_dr_t256_640": (case_reset_poses, dict(map_name="small_loop_only_duckies", tree="t256", dr=True, W=640, H=480, seeds=[5, 6])),
    "small_loop_dr_t256_160": (case_reset_poses, dict(map_name="small_loop_only_duckies", tree="t256", dr=True, W=160, H=120, seeds=list(range(30, 46)))),
    "loop_t256_160": (case_reset_poses, dict(map_name="loop_only_duckies", tree="t256", dr=False, W=160, H=120, seeds=list(range(50, 66)))),
    "loop_dr_t256_160": (case_reset_poses, dict(map_name="loop_only_duckies", tree="t256", dr=True, W=160, H=120, seeds=list(range(70, 86)))),
    "loop_dr_t256_640": (case_reset_poses, dict(map_name="loop_only_duckies", tree="t256", dr=True, W=640, H=480, seeds=[7, 8])),
    "town_t128_320": (case_placed, dict(map_name="test_town", tree="t128", dr=False, W=320, H=240, seed=3, poses="town")),
    "town_dr_t128_320": (case_placed, dict(map_name="test_town", tree="t128", dr=True, W=320, H=240, seed=4, poses="town")),
    "town_t128_640": (case_placed, dict(map_name="test_town", tree="t128", dr=False, W=640, H=480, seed=3, poses="town2")),
    "segment_small_loop_t256_320": (case_reset_poses, dict(map_name="small_loop", tree="t256", dr=False, W=320, H=240, seeds=[90, 91, 92, 93, 94, 95], segment=True)),
    # (segment=True with domain_rand: Texture.bind calls rng.randint, which the numpy Generator the reference's own reset() needs -- it calls
    #  .integers -- does not have: AttributeError in the reference itself; no golden)
    "view_top_down_t256_800": (case_views, dict(map_name="small_loop_only_duckies", tree="t256", W=800, H=600, seeds=[4, 5], view="top_down", dr=False)),
    "view_bbox_t256_320": (case_views, dict(map_name="small_loop_only_duckies", tree="t256", W=320, H=240, seeds=[4, 5, 6, 7], view="bbox", dr=False)),
    "trajectory_t256_160": (case_trajectory, dict(map_name="small_loop_only_duckies", tree="t256", dr=False, W=160, H=120, seed=21, n_steps=80, keep=[0, 20, 40, 60, 80])),
    "trajectory_dr_t256_160": (case_trajectory, dict(map_name="loop_only_duckies", tree="t256", dr=True, W=160, H=120, seed=22, n_steps=80, keep=[0, 20, 40, 60, 80])),
    "episode2_t256_160": (case_second_episode, dict(map_name="small_loop_only_duckies", tree="t256", dr=False, W=160, H=120, seed=9, n_steps=200, n_resets=4)),
    "episode2_dr_t256_160": (case_second_episode, dict(map_name="loop_only_duckies", tree="t256", dr=True, W=160, H=120, seed=11, n_steps=200, n_resets=4)),
}


def build(name):
    fn, kw = CASES[name]
    kw = dict(kw)
    if kw.get("poses") == "town":
        kw["poses"] = town_poses()
    elif kw.get("poses") == "town2":
        kw["poses"] = town_poses()[:2]
    meta = {k: v for k, v in CASES[name][1].items() if k not in ("poses",)}
    if fn is case_views:
        kw.pop("dr")                                    # (carried for the meta record only: the views are rendered without domain randomisation)
    recs = fn(**kw)
    meta["renderer"] = refgl.glshim.renderer()
    out = _stack(recs)
    out["meta"] = np.array(json.dumps(meta))
    return out


def main(argv):
    assert refgl.available(), "run in the build container (needs /root/reference and Mesa's swrast_dri.so)"
    names = argv or list(CASES)
    for name in names:
        data = build(name)
        path = os.path.join(OUT, f"ref_gl_{name}.npz")
        np.savez_compressed(path, **data)
        print(f"{name}: {data['frame'].shape} -> {os.path.getsize(path) / 1024:.0f} KB")


if __name__ == "__main__":
    main(sys.argv[1:])
