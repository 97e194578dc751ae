"""`not gpu`: the oracle (oracle/sim.py, oracle/distortion.py) against the golden vectors that
oracle/make_golden.py recorded from the REFERENCE'S OWN code (tests/golden/).  Runs everywhere
(the GPU box has no /root/reference)."""
import copy
import json
import os

import numpy as np
import pytest

from dtsim import assets
from oracle import sim as osim
from util import EXT, make_oracle

G = os.path.join(os.path.dirname(__file__), "golden")
MAPS = ["small_loop", "small_loop_only_duckies", "loop_only_duckies"]


@pytest.mark.parametrize("m", MAPS + ["junctions"])
def test_probes_bit_exact(m):
    g = np.load(os.path.join(G, f"ref_probes_{m}.npz"))
    if m == "junctions":                           # every tile kind x orientation (oracle/fixtures.py)
        from util import junction_map
        o = osim.OracleSim(junction_map(), EXT, do_reset=False)
    else:
        o = make_oracle(m, do_reset=False)
    for q, (x, z, a) in enumerate(g["poses"]):
        pos = np.array([x, 0, z])
        assert tuple(g["tile"][q]) == o.map.get_grid_coords(pos)
        assert bool(g["drivable"][q]) == o._drivable_pos(pos)
        assert bool(g["collision"][q]) == o._collision(osim.get_agent_corners(pos, a))
        assert bool(g["valid"][q]) == o._valid_pose(pos, a) and bool(g["valid13"][q]) == o._valid_pose(pos, a, 1.3)
        assert g["prox"][q] == o.proximity_penalty2(pos, a)
        try:
            lp = o.get_lane_pos2(pos, a)
            assert g["in_lane"][q] and tuple(g["lane"][q]) == lp
        except osim.NotInLane:
            assert not g["in_lane"][q]
        assert g["reward"][q] == o.compute_reward(pos, a, o.robot_speed)
        o.cur_pos, o.cur_angle, o.step_count = pos, a, 5
        d = o._compute_done_reward()
        assert (bool(g["done"][q]), g["done_reward"][q]) == (d[0], d[1])


def test_map_tables_bit_exact():
    g = np.load(os.path.join(G, "ref_maps.npz"))
    for m in MAPS:
        om = make_oracle(m, do_reset=False).map
        assert np.array_equal(g[f"{m}_curves"], np.concatenate([t["curves"] for t in om.drivable_tiles]))
        if f"{m}_corners" in g:
            assert np.array_equal(g[f"{m}_corners"], om.collidable_corners)
            assert np.array_equal(g[f"{m}_norms"], om.collidable_norms)
            assert np.array_equal(g[f"{m}_centers"], om.collidable_centers)
            assert np.array_equal(g[f"{m}_radii"], om.collidable_safety_radii)


def test_reset_rng_order_bit_exact():
    for rec in json.load(open(os.path.join(G, "ref_resets.json"))):
        o = make_oracle(rec["map"], domain_rand=rec["domain_rand"], seed=rec["seed"], do_reset=False)
        for r in rec["resets"]:
            o.reset()
            assert [float(v) for v in o.cur_pos] == r["pos"] and float(o.cur_angle) == r["angle"]
            assert [float(v) for v in o.horizon_color] == r["horizon"] and [float(v) for v in o.ground_color] == r["ground"]
            assert float(o.wheel_dist) == r["wheel_dist"]
            assert float(np.asarray(o.cam_fov_y).reshape(-1)[0]) == r["cam_fov_y"]
            assert float(np.asarray(o.cam_height).reshape(-1)[0]) == r["cam_height"]
            assert float(o.randomization_settings["trim"][0]) == r["trim"]
            assert [float(v) for v in o.randomization_settings["light_pos"]] == r["light_pos"]


def test_duckie_walk_bit_exact():
    g = np.load(os.path.join(G, "ref_duckie_walk.npz"))
    om = make_oracle("loop_pedestrians", do_reset=False).map
    for t in range(g["center"].shape[0]):
        for k, ob in enumerate(om.objects):
            ob.step(1 / 30)
            assert np.array_equal(g["center"][t, k], np.asarray(ob.center, float)[[0, 2]])
            assert bool(g["active"][t, k]) == ob.pedestrian_active and g["y_rot"][t, k] == ob.y_rot
            assert np.array_equal(g["corners"][t, k], ob.obj_corners)


def test_duckiebot_drive_bit_exact():
    g = np.load(os.path.join(G, "ref_duckiebot_drive.npz"))
    o = make_oracle("loop_dyn_duckiebots", do_reset=False)
    bots = [ob for ob in o.map.objects if ob.kind == "duckiebot"]
    for t in range(g["pos"].shape[0]):
        for k, ob in enumerate(bots):
            ob.step_duckiebot(1 / 30, o.closest_curve_point)
            assert np.array_equal(g["pos"][t, k], np.asarray(ob.pos, float)[[0, 2]]) and g["angle"][t, k] == ob.angle
            assert np.array_equal(g["corners"][t, k], ob.obj_corners)
    assert np.abs(g["pos"][-1] - g["pos"][0]).max() > 0.5


def test_survey_appendix_a_kat():
    k = json.load(open(os.path.join(G, "ref_kat.json")))
    o = make_oracle("small_loop_only_duckies", do_reset=False)
    TS = 0.585
    for r in k["lane"]:
        lp = o.get_lane_pos2(np.array([r["a"] * TS, 0, r["b"] * TS]), r["angle"])
        assert (lp[0], lp[1], lp[2]) == (r["dist"], r["dot_dir"], r["angle_deg"])
    # the values printed in SURVEY.md Appendix A
    assert k["lane"][0]["dist"] == 0.04299622202336341 and k["lane"][0]["angle_deg"] == -53.408636156270326
    assert np.array_equal(osim.get_agent_corners(np.array([1.0, 0, 1.0]), 0.3), np.array(k["corners"]))
    assert osim.AGENT_SAFETY_RAD == k["agent_safety_rad"] == 0.162
    # bezier_closest KAT: straight curve 0 of a tile at the origin
    cps = osim.get_curve("straight", 0, 0, 0, 0.585)[0] - np.array([0.5 * 0.585, 0, 0.5 * 0.585])
    assert osim.bezier_closest(cps, np.array([0, 0, 0.1])) == 0.654296875


def test_distortion_oracle_matches_reference_inversion():
    from oracle import distortion as od
    g = np.load(os.path.join(G, "ref_distortion.npz"))
    for (w, h) in ((160, 120), (84, 84)):
        rx, ry = od.distortion_maps(w, h)
        assert np.array_equal(np.rint(rx.astype(np.float64)).astype(np.int16), g[f"sx_{w}x{h}"])
        assert np.array_equal(np.rint(ry.astype(np.float64)).astype(np.int16), g[f"sy_{w}x{h}"])
    x = np.zeros(3); x[[0, 0, 1]] += 1          # the numpy semantic _invert_map relies on
    assert x.tolist() == [1, 1, 0]


def test_dynamics_model_anchors():
    """duckietown_world DB18 restatement (parity unpinned): published sanity anchors."""
    d = osim.DynamicsDB18(0, 0, 0.0, delay_steps=5)
    for _ in range(400):
        d.integrate(1 / 30, 1.0, 1.0)
    assert abs(d.u - 0.6) < 1e-6 and abs(d.w) < 1e-12           # 0.6 m/s at PWM (1,1)
    d = osim.DynamicsDB18(0, 0, 0.0, delay_steps=5)
    for _ in range(400):
        d.integrate(1 / 30, -1.0, 1.0)
    assert abs(d.w - 7.5) < 1e-6                                   # 7.5 rad/s at (-1, 1)
    d = osim.DynamicsDB18(1.0, 2.0, 0.3, delay_steps=5)
    for k in range(5):                                            # 0.15 s delay = 5 steps: nothing moves yet
        d.integrate(1 / 30, 1.0, 1.0)
        assert (d.x, d.y, d.u) == (1.0, 2.0, 0.0)
    d.integrate(1 / 30, 1.0, 1.0)
    assert d.u > 0


def test_trafficlight_step_matches_reference_golden():
    """TrafficLightObj.step (objects.py:455-463): pattern switches when round(time, 3) is a multiple of freq."""
    from oracle import sim as osim
    g = np.load(os.path.join(G, "ref_trafficlight.npz"))
    for tag, dt in (("30hz", 1 / 30), ("20hz", 1 / 20), ("frame_skip_dt", 1 / 30 / 3)):
        o = osim.OracleObj(kind="trafficlight", pos=np.zeros(3), angle=0.0, scale=1.0, static=True, optional=False,
                           min_coords=np.zeros(3), max_coords=np.ones(3), safety_radius=0.0,
                           obj_corners=np.zeros((4, 2)), obj_norm=np.eye(2), light_freq=5, light_pattern=0)
        pat = np.zeros(1300, np.uint8)
        for t in range(1300):
            o.step(dt)
            pat[t] = o.light_pattern
        assert np.array_equal(pat, g[tag]), tag


def test_checkerboard_step_matches_reference_golden():
    """CheckerboardObj.step (objects.py:531-587): the scripted calibration motion of the centre."""
    from oracle import sim as osim
    g = np.load(os.path.join(G, "ref_checkerboard.npz"))["center"]
    p0 = np.array([1.5, 0.0, 2.25])
    o = osim.OracleObj(kind="checkerboard", pos=p0.copy(), angle=0.0, scale=1.0, static=False, optional=False,
                       min_coords=np.zeros(3), max_coords=np.ones(3), safety_radius=0.0, obj_corners=np.zeros((4, 2)),
                       obj_norm=np.eye(2), center=p0.copy(), reset_start=p0.copy(), steps=-20)
    got = np.zeros_like(g)
    for t in range(len(g)):
        o.step(1 / 30)
        got[t] = o.center
    assert np.array_equal(got, g)
