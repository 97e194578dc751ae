"""`not gpu`, build container only: pin the oracle against the reference's OWN code, imported
through oracle/refstub.py (skipped where /root/reference does not exist, e.g. the GPU box)."""
import copy

import numpy as np
import pytest

from dtsim import assets
from oracle import refstub, sim as osim
from util import EXT

pytestmark = pytest.mark.skipif(not refstub.available(), reason="reference tree not present")


def _ref(map_name, dr=False, seed=None, md=None):
    from oracle.make_golden import ref_sim
    return ref_sim(map_name, dr, seed, md)


@pytest.mark.parametrize("m", ["small_loop_only_duckies", "loop_only_duckies"])
def test_fresh_random_poses(m):
    r, ns = _ref(m)
    o = osim.OracleSim(assets.get_map(m), EXT, do_reset=False)
    rng = np.random.default_rng(99)
    for _ in range(800):
        c = r.collidable_centers[rng.integers(len(r.collidable_centers))]
        pos = c + np.array([rng.uniform(-0.4, 0.4), 0, rng.uniform(-0.4, 0.4)])
        a = rng.uniform(-7, 7)
        assert r._valid_pose(pos, a) == o._valid_pose(pos, a)
        assert r._collision(ns.simulator.get_agent_corners(pos, a)) == o._collision(osim.get_agent_corners(pos, a))
        assert r.proximity_penalty2(pos, a) == o.proximity_penalty2(pos, a)
        try:
            lp = tuple(float(v) for v in r.get_lane_pos2(pos, a))
        except ns.simulator.NotInLane:
            lp = None
        try:
            lp2 = o.get_lane_pos2(pos, a)
        except osim.NotInLane:
            lp2 = None
        assert lp == lp2


@pytest.mark.parametrize("dr", [False, True])
def test_reset_streams(dr):
    for seed in (11, 12, 13):
        r, _ = _ref("loop_only_duckies", dr, seed)
        o = osim.OracleSim(assets.get_map("loop_only_duckies"), EXT, domain_rand=dr, seed=seed, do_reset=False)
        for _ in range(3):
            r.reset(); o.reset()
            assert np.array_equal(r.cur_pos, o.cur_pos) and r.cur_angle == o.cur_angle
            assert np.array_equal(r.ground_color, o.ground_color)


def test_reference_fill_holes_equals_oracle_and_product():
    from dtsim import distortion as pd
    from oracle import distortion as od
    ns = refstub.load()
    Dm = ns.distortion.Distortion.__new__(ns.distortion.Distortion)
    mapx, mapy = od.rectify_maps(160, 120)
    rx, ry = Dm._invert_map(mapx.copy(), mapy.copy())
    ox, oy = od.invert_map(mapx.copy(), mapy.copy())
    px, py = pd.distortion_maps(160, 120)
    assert np.array_equal(rx, ox) and np.array_equal(ry, oy)
    assert np.array_equal(rx, px) and np.array_equal(ry, py)
