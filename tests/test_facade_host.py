"""`not gpu`: the gym facade's pure host helpers (gym_duckietown/simulator.py) against the oracle's camera model -- no device, no reference tree.
(The same helpers are pinned on the reference's own GL calls where /root/reference exists: tests/test_oracle_vs_reference.py.)"""
import math

import numpy as np

from gym_duckietown.simulator import agent_bbox_angle, curve_overlay_segments, gl_light_to_eye, viewer_camera
from oracle import raster


def _mv(cam):
    return dict(C=cam.C, sa=cam.sa, ca=cam.ca, sth=cam.sth, cth=cam.cth)


def test_gl_light_to_eye_is_the_camera_s_world_to_eye_transform():
    rng = np.random.default_rng(0)
    for _ in range(20):
        cam = raster.Camera(rng.uniform(0, 4, 3) * [1, 0, 1], rng.uniform(-7, 7), cam_height=rng.uniform(0.08, 0.9), cam_angle_deg=rng.uniform(5, 90))
        p = rng.uniform(-5, 5, 3)
        got = gl_light_to_eye(_mv(cam), [*p, 1.0])
        assert got[3] == 1.0 and np.allclose(got[:3], cam.to_eye(p), atol=1e-12)
        got = gl_light_to_eye(_mv(cam), [*(2.0 * p), 2.0])                       # homogeneous: the same point
        assert np.allclose(got[:3], cam.to_eye(p), atol=1e-12)
        d = rng.uniform(-200, 200, 3)
        got = gl_light_to_eye(_mv(cam), list(d))                                   # three components: w = 0, a direction -- rotated, not translated
        assert got[3] == 0.0 and np.allclose(got[:3], cam.normal_to_eye(d), atol=1e-9)
        assert abs(np.linalg.norm(got[:3]) - np.linalg.norm(d)) < 1e-9


def test_viewer_camera_reproduces_the_window_views_eye_and_axes():
    ts, gw, gh, fov = 0.585, 8, 7, 75.0
    vp, va, vh, vdeg = viewer_camera(True, False, [1.0, 0.0, 2.0], 0.3, gw, gh, ts, fov)
    cam = raster.Camera(vp, va, cam_height=vh, cam_angle_deg=vdeg, cam_fov_y_deg=fov)
    a, b = gw * ts / 2, gh * ts / 2
    Hf = (max(a, b) + 0.1) / math.tan(math.radians(fov) / 2)
    assert np.allclose(cam.C, [a, Hf, b], atol=1e-12)                              # gluLookAt((a, H, b), (a, 0, b - 0.01), +y)
    fwd = np.array([0.0, -Hf, -0.01]) / math.hypot(Hf, 0.01)
    assert np.allclose(cam.to_eye(cam.C + fwd), [0, 0, -1], atol=1e-9) and np.allclose(cam.to_eye(cam.C + np.array([1.0, 0, 0])), [1, 0, 0], atol=1e-9)
    pos, ang = np.array([1.3, 0.0, 0.9]), 2.2
    vp, va, vh, vdeg = viewer_camera(False, True, pos, ang, gw, gh, ts, fov)       # draw_bbox: 0.8 m above the robot, straight down, no forward offset
    cam = raster.Camera(vp, va, cam_height=vh, cam_angle_deg=vdeg)
    assert np.allclose(cam.C, [pos[0], 0.8, pos[2]], atol=1e-15) and abs(cam.sth - 1.0) < 1e-15
    vp, va, vh, vdeg = viewer_camera(False, False, pos, ang, gw, gh, ts, fov)
    assert vp == [1.3, 0.0, 0.9] and va == ang and vh is None and vdeg is None     # the agent camera: the env's own height / pitch


def test_overlay_quirks_follow_the_tile_loop_s_angle():
    """simulator.py:1853-1918: the tile loop (columns outer) rebinds `angle` to the tile's orientation index."""
    def tile(angle, curves=None):
        return {"angle": angle, "drivable": curves is not None, "curves": curves}
    c0 = np.array([[[0, 0, 0], [0.1, 0, 0], [0.2, 0, 0], [0.3, 0, 0]],           # chord +x
                   [[0.3, 0, 0.1], [0.2, 0, 0.1], [0.1, 0, 0.1], [0, 0, 0.1]]], dtype=np.float64)   # chord -x
    grid = [tile(0, c0), None, tile(2, c0), tile(3)]                               # 2 x 2, row-major: (0,0) (1,0) / (0,1) (1,1)
    assert agent_bbox_angle(grid, 2, 2, 0.7) == 3.0                                # last tile of the loop: i = 1, j = 1
    assert agent_bbox_angle([tile(1), None, None, None], 2, 2, 0.7) == 1.0 and agent_bbox_angle([None] * 4, 2, 2, 0.7) == 0.7
    segs = np.asarray(curve_overlay_segments(grid, 2, 2))
    assert segs.shape == (2 * 2 * 19, 9)
    # tile (0, 0), angle index 0: get_dir_vec(0) = +x -> the +x chord is red; tile (0, 1), index 2: get_dir_vec(2 rad) has cos < 0 -> the -x chord
    first = segs[:19]
    assert np.all(first[:, 6:9] == [1, 0, 0]) and first[0, 0] == 0.0 and first[-1, 3] == 0.3
    third = segs[38:57]
    assert np.all(third[:, 6:9] == [1, 0, 0]) and third[0, 0] == 0.3 and third[-1, 3] == 0.0 and np.allclose(third[:, 2], 0.1, atol=1e-15)
    assert np.all(segs[19:38, 6:9] == [0, 0, 1]) and np.all(segs[57:, 6:9] == [0, 0, 1])
