"""`-m gpu`: the reference's GL_LINE overlays -- draw_curve (simulator.py:1886-1904, graphics.py:336-349) and draw_bbox
(simulator.py:1776-1778, 1907-1918, objects.py:131-139) -- as the post-pass dtsim_draw_lines (k_overlay_lines) against the oracle's
statement of the same interpretation (oracle/raster.py: overlay_lines; the GL state at those draw calls is whatever the previous
call left behind: PARITY UNPINNED against real GL).

Coverage of a sample by a 1-px line can flip where float32 projection meets the oracle's float64: the comparisons allow a small share
of the LINE pixels to differ, never a bulk difference.
"""
import numpy as np
import pytest

from dtsim import BatchedSimulator, _ffi
from dtsim import distortion as pdist
from oracle import raster
from test_gpu_render import _camera, _obj_states, _scene, _stats

pytestmark = pytest.mark.gpu


def _segments(sim, e, n, seed):
    """Random world-space segments around env e's robot: on the floor (y = 0.01), some raised, some running behind the camera."""
    rng = np.random.default_rng(seed)
    pos, ang = sim.read(_ffi.FIELD_POS)[e], float(sim.read(_ffi.FIELD_ANGLE)[e])
    d = np.array([np.cos(ang), 0.0, -np.sin(ang)])
    r = np.array([np.sin(ang), 0.0, np.cos(ang)])
    out = []
    for k in range(n):
        a = pos + d * rng.uniform(-0.3, 1.5) + r * rng.uniform(-0.6, 0.6)
        b = a + d * rng.uniform(-0.5, 0.5) + r * rng.uniform(-0.5, 0.5)
        ya, yb = (0.01, 0.01) if k % 3 else (rng.uniform(0.0, 0.2), rng.uniform(0.0, 0.2))
        col = [(1, 0, 0), (0, 0, 1), (0.3, 0.9, 0.2)][k % 3]
        out.append([a[0], ya, a[2], b[0], yb, b[2], *col])
    return np.asarray(out, np.float32)


@pytest.mark.parametrize("W,H,dr", [(160, 120, False), (640, 480, False), (320, 240, True)])
def test_draw_lines_match_oracle_on_the_rendered_frame(W, H, dr):
    """No fisheye: the device's own frame before the overlay + the oracle's overlay = the device's frame after it, except where a
    sample's coverage flips; three envs with their own segment lists (env_idx), near-plane clipping, several lines per pixel."""
    N = 3
    sim = BatchedSimulator("small_loop", N, camera_width=W, camera_height=H, distortion=False, domain_rand=dr, seed=21)
    sim.step(np.random.default_rng(1).uniform(0.2, 0.8, (5, N, 2)).astype(np.float32), n_steps=5)
    sim.render()
    before = sim.frames_host().copy()
    segs = [_segments(sim, e, 40, 100 + e) for e in range(N)]
    sim.draw_lines(np.concatenate(segs), np.repeat(np.arange(N), [len(s) for s in segs]))
    after = sim.frames_host()
    for e in range(N):
        cam = _camera(sim, e, W, H, dr)
        want = raster.overlay_lines(before[e], cam, segs[e])
        touched = (want != before[e]).any(-1)
        assert touched.sum() > 50                         # the lines are in view
        diff = (after[e].astype(int) - want.astype(int))
        bad = (np.abs(diff).max(-1) > 1)
        assert bad.sum() <= 0.03 * touched.sum(), (e, int(bad.sum()), int(touched.sum()))
        assert not (bad & ~(touched | (after[e] != before[e]).any(-1))).any()
        assert np.array_equal(after[e][~touched & ~bad], before[e][~touched & ~bad])   # nothing else moved
    with pytest.raises(Exception):
        sim.draw_lines(np.zeros((2, 9), np.float32), [1, 0])      # env_idx must be non-decreasing
    with pytest.raises(Exception):
        sim.draw_lines(np.zeros((1, 9), np.float32), [N])         # out of range
    sim.close()


def test_draw_lines_through_the_fisheye_match_oracle():
    """With the fisheye the overlay is drawn per OUTPUT pixel at its source pixel: the whole frame against the oracle's
    render -> overlay -> remap, thresholds of the plane-only frames (tests/test_gpu_render.py)."""
    W, H, N = 640, 480, 2
    sim = BatchedSimulator("small_loop", N, camera_width=W, camera_height=H, distortion=True, domain_rand=False, seed=5)
    sim.step(np.random.default_rng(2).uniform(0.2, 0.8, (6, N, 2)).astype(np.float32), n_steps=6)
    sim.render()
    segs = [_segments(sim, e, 30, 7 + e) for e in range(N)]
    sim.draw_lines(np.concatenate(segs), np.repeat(np.arange(N), [len(s) for s in segs]))
    frames = sim.frames_host()
    scene = _scene("small_loop")
    rmap = pdist.distortion_maps(W, H)
    for e in range(N):
        cam = _camera(sim, e, W, H, False)
        plain = raster.render_obs(cam, scene, "pixel", rmap)
        ref = raster.render_obs(cam, scene, "pixel", rmap, lines=segs[e])
        n_line = int((ref != plain).any(-1).sum())
        assert n_line > 200
        s = _stats(frames[e], ref)
        assert s["mean"] <= 0.05 and s["frac_gt2"] <= 5e-4 + 0.05 * n_line / (W * H), (e, s, n_line)
    sim.close()


def test_simulator_draw_curve_and_draw_bbox():
    """The drop-in facade: Simulator(draw_curve=True) overlays the lane curves of every drivable tile (the one along the heading red,
    the others blue); Simulator(draw_bbox=True) switches to the reference's debugging view (0.8 m above the robot, looking down, no
    fisheye) and outlines the objects' and the agent's collision rectangles."""
    from gym_duckietown.simulator import Simulator
    W, H = 320, 240
    env = Simulator(map_name="small_loop", domain_rand=False, draw_curve=True, camera_width=W, camera_height=H, seed=3, distortion=False)
    for _ in range(3):
        obs, _, _, _ = env.step(np.array([0.5, 0.5]))
    lines = env._overlay_lines()
    n_curves = sum(len(t["curves"]) for t in env.grid if t is not None and t["drivable"])
    assert lines.shape == (19 * n_curves, 9)
    assert {tuple(c) for c in lines[:, 6:9].tolist()} == {(1.0, 0.0, 0.0), (0.0, 0.0, 1.0)}
    red = (obs[..., 0] > 120) & (obs[..., 1] < 60) & (obs[..., 2] < 60)
    assert red.sum() > 30, int(red.sum())                                      # the curve ahead, drawn red
    scene = _scene("small_loop")
    cam = _camera(env._sim, 0, W, H, False)
    ref = raster.render_obs(cam, scene, __import__("util").oracle_mode(env._sim), None, lines=lines)
    s = _stats(obs, ref)
    assert s["mean"] <= 0.1 and s["frac_gt2"] <= 3e-3, s
    env.close()

    env = Simulator(map_name="loop_only_duckies", domain_rand=False, draw_bbox=True, camera_width=W, camera_height=H, seed=4, distortion=True)
    assert env.distortion is False                                             # simulator.py:125: no fisheye in this mode
    obs = env.render_obs()
    lines = env._overlay_lines()
    n_vis = int(env._sim.read(_ffi.FIELD_OBJ_VISIBLE)[0][:len(env.objects)].sum())
    assert lines.shape == (4 * (n_vis + 1), 9) and np.allclose(lines[:, 1], 0.01) and np.allclose(lines[:, 4], 0.01)
    red = (obs[..., 0] > 120) & (obs[..., 1] < 60) & (obs[..., 2] < 60)
    assert red.sum() > 40, int(red.sum())                                      # at least the agent's own rectangle, in the image centre
    ys, xs = np.nonzero(red)
    assert abs(xs.mean() - W / 2) < W / 4 and abs(ys.mean() - H / 2) < H / 3
    v = env._viewers[(False, (W, H))]
    scene = _scene("loop_only_duckies")
    cam = _camera(v, 0, W, H, True)
    assert abs(cam.sth - 1.0) < 1e-6 and abs(cam.C[1] - 0.8) < 1e-6            # looking straight down from 0.8 m
    ref = raster.render_obs(cam, scene, __import__("util").oracle_mode(v), None, obj_states=_obj_states(v, 0, scene), lines=lines)
    s = _stats(obs, ref)
    assert s["mean"] <= 0.3 and s["frac_gt2"] <= 1e-2, s                       # (the lines are drawn over the meshes here: no depth test against them)
    env.close()
