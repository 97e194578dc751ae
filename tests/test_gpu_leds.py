"""`-m gpu`: the LED spheres of the reference's enable_leds (objects.py:68-121: per LED of a duckiebot-kind object a 1 cm gluSphere at
alpha 1 and a halo at alpha 0.2, glBlendFunc(GL_SRC_ALPHA, GL_ONE), depth test and depth writes on) as the post-pass dtsim_draw_leds
(k_overlay_leds) against the oracle's statement of the same interpretation (oracle/raster.py: led_spheres, overlay_leds -- analytic
spheres, front surfaces, after all opaque objects: PARITY UNPINNED against real GL, whose result depends on gluSphere's strip order;
DESIGN.md 7 N4).

A sample at a sphere's silhouette, or where a sphere cuts a mesh / the ground, can fall on either side between float32 and float64: the
comparisons allow a small share of the TOUCHED pixels to differ, never a bulk difference.
"""
import math

import numpy as np
import pytest

from dtsim import BatchedSimulator, _ffi
from dtsim import distortion as pdist
from oracle import raster
from test_gpu_render import _camera, _obj_states, _scene, _stats

pytestmark = pytest.mark.gpu


def _spheres(sim, e, scene, seed):
    """World-space test spheres in front of env e's robot: free ones with halos, one cutting the ground, one through the nearest mesh
    object, one hidden behind it, one across the near plane, one at the image border, overlapping pairs (depth writes between spheres)."""
    rng = np.random.default_rng(seed)
    pos, ang = sim.read(_ffi.FIELD_POS)[e], float(sim.read(_ffi.FIELD_ANGLE)[e])
    d = np.array([math.cos(ang), 0.0, -math.sin(ang)])
    r = np.array([math.sin(ang), 0.0, math.cos(ang)])
    up = np.array([0.0, 1.0, 0.0])
    out = []
    for k in range(6):                                     # free spheres + halos, two of them overlapping on screen
        c = pos + d * rng.uniform(0.25, 0.9) + r * rng.uniform(-0.25, 0.25) + up * rng.uniform(0.03, 0.15)
        col = [(1, 0, 0), (0.5, 0.5, 0.5), (0, 0, 1), (0.5, 0, 0), (0, 0, 0.2), (0.2, 0.9, 0.3)][k]
        rad = rng.uniform(0.01, 0.035)
        out.append([*c, rad, *col, 1.0])
        out.append([*c, rad * (1.5 + 2.5 * float(np.mean(col))), *col, 0.2])
    c = pos + d * 0.4 + r * 0.05
    out.append([*c, 0.04, 0, 1, 0, 1.0])                   # centre on the road surface: half of it below the tile plane
    out.append([*(pos + d * 0.066 + up * 0.1), 0.05, 1, 1, 0, 1.0])   # around the camera: across the near plane
    out.append([*(pos + d * 0.5 + r * 0.62 + up * 0.05), 0.06, 1, 0, 1, 0.2])   # at the image border
    st = _obj_states(sim, e, scene)
    objs = [(np.linalg.norm(np.asarray(s["pos"]) - pos), np.asarray(s["pos"], dtype=np.float64)) for s in st if s["visible"]]
    if objs:
        p = min(objs, key=lambda t: t[0])[1]
        out.append([*(p + up * 0.05), 0.05, 0, 0.6, 1, 1.0])           # through the nearest object's mesh
        away = (p - pos) / max(np.linalg.norm(p - pos), 1e-6)
        out.append([*(p + away * 0.12 + up * 0.03), 0.02, 1, 1, 1, 1.0])   # behind it
    return np.asarray(out, np.float32)


@pytest.mark.parametrize("map_name,W,H,dr", [("loop_dyn_duckiebots", 320, 240, False), ("loop_only_duckies", 640, 480, False), ("loop_pedestrians", 320, 240, True)])
def test_draw_leds_match_oracle_on_the_rendered_frame(map_name, W, H, dr):
    """No fisheye: the device's own frame before the pass + the oracle's overlay_leds on the oracle's per-sample depth buffers = the device's
    frame after it, except at silhouettes; three envs with their own sphere lists."""
    N = 3
    sim = BatchedSimulator(map_name, N, camera_width=W, camera_height=H, distortion=False, domain_rand=dr, seed=31)
    sim.step(np.random.default_rng(4).uniform(0.2, 0.7, (6, N, 2)).astype(np.float32), n_steps=6)
    sim.render()
    before = sim.frames_host().copy()
    scene = _scene(map_name)
    sps = [_spheres(sim, e, scene, 50 + e) for e in range(N)]
    sim.draw_leds(np.concatenate(sps), np.repeat(np.arange(N), [len(s) for s in sps]))
    after = sim.frames_host()
    for e in range(N):
        cam = _camera(sim, e, W, H, dr)
        _, depths = raster.render_rectilinear(cam, scene, "pixel", _obj_states(sim, e, scene), return_depth=True)
        want = raster.overlay_leds(before[e], cam, depths, sps[e])
        touched = (want != before[e]).any(-1)
        assert touched.sum() > 150, int(touched.sum())    # the spheres are in view
        assert (want.astype(int) - before[e].astype(int)).min() >= 0   # additive
        bad = np.abs(after[e].astype(int) - want.astype(int)).max(-1) > 1
        assert bad.sum() <= 0.04 * touched.sum() + 4, (e, int(bad.sum()), int(touched.sum()))
        moved = (after[e] != before[e]).any(-1)
        assert not (bad & ~(touched | moved)).any()
        assert np.array_equal(after[e][~touched & ~moved], before[e][~touched & ~moved])
        assert (moved & ~touched).sum() <= 0.04 * touched.sum() + 4      # nothing moved far from where the oracle draws
    with pytest.raises(Exception):
        sim.draw_leds(np.zeros((2, 8), np.float32), [1, 0])       # env_idx must be non-decreasing
    with pytest.raises(Exception):
        sim.draw_leds(np.zeros((1, 8), np.float32), [N])          # out of range
    sim.close()


def test_draw_leds_through_the_fisheye_and_state_rules():
    """With the fisheye the pass runs per OUTPUT pixel at its source pixel: what it adds against what the oracle adds (render ->
    overlay -> remap, with and without spheres).  And the call needs a colour render of the current tables before it."""
    W, H, N = 640, 480, 2
    sim = BatchedSimulator("loop_dyn_duckiebots", N, camera_width=W, camera_height=H, distortion=True, domain_rand=False, seed=8)
    with pytest.raises(Exception):
        sim.draw_leds(np.array([[0, 0, 0, 0.01, 1, 1, 1, 1]], np.float32))       # nothing rendered yet
    sim.step(np.random.default_rng(3).uniform(0.2, 0.7, (4, N, 2)).astype(np.float32), n_steps=4)
    sim.render()
    before = sim.frames_host().copy()
    scene = _scene("loop_dyn_duckiebots")
    sps = [_spheres(sim, e, scene, 70 + e) for e in range(N)]
    sim.draw_leds(np.concatenate(sps), np.repeat(np.arange(N), [len(s) for s in sps]))
    after = sim.frames_host()
    rmap = pdist.distortion_maps(W, H)
    for e in range(N):
        cam = _camera(sim, e, W, H, False)
        st = _obj_states(sim, e, scene)
        plain = raster.render_obs(cam, scene, "pixel", rmap, obj_states=st)
        ref = raster.render_obs(cam, scene, "pixel", rmap, obj_states=st, leds=sps[e])
        touched = (ref != plain).any(-1)
        assert touched.sum() > 300
        got = after[e].astype(int) - before[e].astype(int)
        want = ref.astype(int) - plain.astype(int)
        bad = np.abs(got - want).max(-1) > 2
        assert bad.sum() <= 0.05 * touched.sum() + 4, (e, int(bad.sum()), int(touched.sum()))
        s = _stats(after[e], ref)
        assert s["mean"] <= 0.1 and s["frac_gt2"] <= 2e-3 + 0.05 * touched.sum() / (W * H), (e, s)
    sim.close()


def test_simulator_enable_leds():
    """The drop-in facade: Simulator(enable_leds=True) states the reference's spheres for every duckiebot-kind object -- followers with
    DuckiebotObj.leds_color, in the object's translate / scale / rotate -- and blends them into every frame it returns."""
    from gym_duckietown.simulator import Simulator
    W, H = 320, 240
    env = Simulator(map_name="loop_dyn_duckiebots", domain_rand=False, enable_leds=True, camera_width=W, camera_height=H, seed=2, distortion=False)
    for _ in range(3):
        obs, _, _, _ = env.step(np.array([0.4, 0.4]))
    sp = env._led_spheres()
    scene = _scene("loop_dyn_duckiebots")
    st = _obj_states(env._sim, 0, scene)
    ref_sp = raster.led_spheres(scene, st)
    n_bots = sum(1 for o in env.objects if o.kind == "duckiebot")
    assert n_bots >= 1 and sp.shape == (10 * n_bots, 8) and ref_sp.shape == sp.shape
    assert np.allclose(sp, ref_sp, atol=1e-5), float(np.abs(sp - ref_sp).max())
    assert np.allclose(sp[0::2, 7], 1.0) and np.allclose(sp[1::2, 7], 0.2)                    # sphere, halo, sphere, halo ...
    assert np.allclose(sp[1::2, 3], sp[1::2, 4:7].mean(axis=1) * 0.04 * env.objects[0].scale)  # halo radius = mean(colour) x 4 cm (x scale)
    cam = _camera(env._sim, 0, W, H, False)
    ref = raster.render_obs(cam, scene, __import__("util").oracle_mode(env._sim), None, obj_states=st, leds=ref_sp)
    s = _stats(obs, ref)
    assert s["mean"] <= 0.1 and s["frac_gt2"] <= 3e-3, s
    img = env.render(mode="top_down")                                                          # the window views take the same pass
    assert img.shape[2] == 3
    env.close()
    with pytest.raises(NotImplementedError):
        Simulator(map_name="small_loop", camera_rand=True)


def test_batched_led_spheres_for_every_env_and_map():
    """BatchedSimulator.led_spheres(): the vector API's statement of the spheres -- per env, from ITS map's objects and their current poses --
    against the oracle's list per env; drawn for all envs in one dtsim_draw_leds call."""
    N, W, H = 4, 160, 120
    names = ["loop_dyn_duckiebots", "loop_only_duckies"]
    sim = BatchedSimulator(names, N, camera_width=W, camera_height=H, distortion=False, domain_rand=False, seed=12, map_cycle=True)
    sim.reset(mask=(np.arange(N) % 2 == 0))
    sim.step(np.random.default_rng(9).uniform(0.2, 0.6, (20, N, 2)).astype(np.float32), n_steps=20)
    sp, idx = sim.led_spheres()
    scenes = [_scene(n) for n in names]
    mid = sim.read(_ffi.FIELD_MAP_ID)
    assert set(np.unique(mid)) == {0, 1}
    for e in range(N):
        sc = scenes[int(mid[e])]
        want = raster.led_spheres(sc, _obj_states(sim, e, sc))
        got = sp[idx == e]
        assert got.shape == want.shape and (want.shape[0] > 0) == (int(mid[e]) == 0)       # only the duckiebot map has LEDs
        assert np.allclose(got, want, atol=1e-5)
    assert np.all(np.diff(idx) >= 0)
    sim.render()
    sim.draw_leds(sp, idx)                               # (the fixture's unit meshes put these LEDs inside the bodies: the call itself is what is exercised)
    sub, _ = sim.led_spheres([1])
    assert sub.shape[0] == int((idx == 1).sum())
    sim.close()
