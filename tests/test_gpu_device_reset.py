"""GPU: device-side reset sampler.  Same distributions and acceptance test as the
reference's reset() (simulator.py:546-738), different RNG stream (Philox): checked through the properties
the reference's reset guarantees, not through RNG-order parity (tests/test_gpu_physics.py covers that mode)."""
import numpy as np
import pytest

from dtsim import BatchedSimulator, _ffi

pytestmark = pytest.mark.gpu


def _spawn_ok(sim, pos, ang, accept):
    poses = np.stack([pos[:, 0], pos[:, 2], ang], axis=1)
    pr = sim.query(np.arange(sim.num_envs, dtype=np.int32), poses, safety_factor=1.3)
    return (pr["valid"].astype(bool) & ~pr["inconvenient"].astype(bool) & pr["in_lane"].astype(bool)
            & (np.abs(pr["angle_deg"]) < accept)), pr


@pytest.mark.parametrize("map_name,dr", [("small_loop", False), ("loop_only_duckies", True), ("loop_pedestrians", True)])
def test_device_reset_spawns_are_valid(map_name, dr):
    N = 2048
    sim = BatchedSimulator(map_name, N, render=False, domain_rand=dr, seed=42, device_reset=True, accept_start_angle_deg=60)
    pos, ang = sim.read(_ffi.FIELD_POS), sim.read(_ffi.FIELD_ANGLE)
    ok, pr = _spawn_ok(sim, pos, ang, 60)
    assert ok.all(), f"{(~ok).sum()} invalid spawns"
    # every drivable tile gets used, roughly uniformly (simulator.py:675: uniform over drivable tiles)
    mt = sim.maps[0]
    tiles = pr["tile_j"].astype(int) * mt.grid_w + pr["tile_i"].astype(int)
    want = sorted(j * mt.grid_w + i for i, j in mt.drivable_tiles)
    cnt = np.array([(tiles == t).sum() for t in want])
    assert (cnt > 0).all() and cnt.max() < 6 * max(cnt.min(), 1) + 30
    # headings cover the accepted cone on both lane directions
    assert np.std(ang) > 0.5
    sim.close()


def test_device_reset_is_counter_based_and_seeded():
    a = BatchedSimulator("small_loop", 256, render=False, domain_rand=False, seed=7, device_reset=True)
    b = BatchedSimulator("small_loop", 256, render=False, domain_rand=False, seed=7, device_reset=True)
    c = BatchedSimulator("small_loop", 256, render=False, domain_rand=False, seed=8, device_reset=True)
    pa, pb, pc = a.read(_ffi.FIELD_POS), b.read(_ffi.FIELD_POS), c.read(_ffi.FIELD_POS)
    assert np.array_equal(pa, pb) and not np.array_equal(pa, pc)
    # a second reset draws episode 2 -> different poses; masked reset only touches the masked envs
    mask = np.zeros(256, bool); mask[::2] = True
    a.reset(mask)
    pa2 = a.read(_ffi.FIELD_POS)
    assert np.array_equal(pa2[1::2], pa[1::2]) and not np.array_equal(pa2[::2], pa[::2])
    for s in (a, b, c):
        s.close()


def test_auto_reset_uses_the_sampler():
    N = 1024
    sim = BatchedSimulator("small_loop_only_duckies", N, render=False, domain_rand=True, seed=3, device_reset=True,
                           auto_reset=True, max_steps=40, action_mode="vel_steer")
    rng = np.random.default_rng(0)
    seen_ep = np.zeros(N, np.int64)
    for t in range(6):
        sim.step(rng.uniform(-1, 1, (25, N, 2)).astype(np.float32), n_steps=25)
        seen_ep = np.maximum(seen_ep, sim.read(_ffi.FIELD_EPISODE))
    assert seen_ep.min() >= 2                              # max_steps = 40 forces several episodes everywhere
    # envs that are currently in progress sit at valid poses; freshly reset ones satisfy the spawn test
    done = sim.read(_ffi.FIELD_DONE).astype(bool)
    pos, ang = sim.read(_ffi.FIELD_POS), sim.read(_ffi.FIELD_ANGLE)
    pr = sim.query(np.arange(N, dtype=np.int32), np.stack([pos[:, 0], pos[:, 2], ang], 1), safety_factor=1.0)
    assert pr["valid"].astype(bool)[~done].all()
    sim.close()


def test_multimap_cycle_on_device():
    sim = BatchedSimulator(["loop_only_duckies", "small_loop_only_duckies"], 64, render=False, domain_rand=False, seed=1,
                           device_reset=True, map_cycle=True)
    m0 = sim.read(_ffi.FIELD_MAP_ID).copy()
    sim.reset()
    m1 = sim.read(_ffi.FIELD_MAP_ID)
    assert set(m0.tolist()) == {0, 1} and np.array_equal(m1, 1 - m0)     # MultiMapEnv: next env at every reset
    pos, ang = sim.read(_ffi.FIELD_POS), sim.read(_ffi.FIELD_ANGLE)
    pr = sim.query(np.arange(64, dtype=np.int32), np.stack([pos[:, 0], pos[:, 2], ang], 1), safety_factor=1.3)
    assert pr["valid"].astype(bool).all()
    sim.close()


def test_sampler_errors():
    sim = BatchedSimulator("small_loop", 4, render=False, seed=1)
    rc = sim._lib.dtsim_reset(sim._h, None, None)
    assert rc != 0 and b"sampler" in sim._lib.dtsim_last_error()
    sim.close()


def test_device_sampler_randomises_walking_duckies():
    """DuckieObj under domain_rand with the device sampler: the reference's distributions (objects.py:348-363 at
    creation, :424-427 at finish_walk -- drawn there from the unseeded global np.random, so only the distributions
    can be matched): wait = randint(3, 20), vel = |N(0.02, 0.005)| with the sign flipping per walk,
    wiggle = pi / choice([14, 15, 16])."""
    N = 1024
    kw = dict(render=False, domain_rand=True, device_reset=True, max_steps=10**6)
    sim = BatchedSimulator("loop_pedestrians", N, seed=11, **kw)
    nd = sim.maps[0].n_dynamic
    assert nd >= 4
    p0 = sim.read(_ffi.FIELD_OBJ_PARAMS)[:, :nd]                       # [N, nd, (vel, wait, wiggle)]
    vel, wait, wig = p0[..., 0], p0[..., 1], p0[..., 2]
    assert (wait == np.round(wait)).all() and wait.min() == 3 and wait.max() == 19
    cnt = np.bincount(wait.astype(int).ravel(), minlength=20)[3:20]
    assert cnt.min() > 0.6 * cnt.mean()                                # uniform over the 17 values
    assert (vel > 0).all() and abs(vel.mean() - 0.02) < 5e-4 and abs(vel.std() - 0.005) < 5e-4
    k = np.round(np.pi / wig).astype(int)
    assert np.allclose(np.pi / k, wig, rtol=0, atol=1e-15) and set(np.unique(k)) == {14, 15, 16}
    # same seed -> same draws; another seed -> different; envs differ from each other
    same = BatchedSimulator("loop_pedestrians", N, seed=11, **kw)
    other = BatchedSimulator("loop_pedestrians", N, seed=12, **kw)
    assert np.array_equal(same.read(_ffi.FIELD_OBJ_PARAMS), sim.read(_ffi.FIELD_OBJ_PARAMS))
    assert not np.array_equal(other.read(_ffi.FIELD_OBJ_PARAMS)[:, :nd], p0)
    assert len(np.unique(vel[:, 0])) > N // 2
    same.close(); other.close()
    # walk: a duckie waits `wait` seconds, walks tile_size at |vel| per step, then finish_walk redraws
    T = 30 * 20 + 60
    sim.step(np.zeros((T, N, 2), np.float32), n_steps=T)
    p1 = sim.read(_ffi.FIELD_OBJ_PARAMS)[:, :nd]
    flipped = p1[..., 0] < 0
    assert 0.6 < flipped.mean() <= 1.0                                 # most have finished their first walk by now
    mag = np.abs(p1[..., 0][flipped])
    assert abs(mag.mean() - 0.02) < 6e-4 and abs(mag.std() - 0.005) < 6e-4
    assert not np.allclose(mag, np.abs(vel[flipped]))                  # a fresh draw, not just the sign flip
    w1 = p1[..., 1][flipped]
    assert w1.max() <= 19 and w1.min() > -1.0 / 30 - 1e-9              # counting down from an integer in [3, 19]
    sim.close()
    # without the device sampler the host path keeps the explicit (non-random) parameters
    host = BatchedSimulator("loop_pedestrians", 8, render=False, domain_rand=True, seed=11)
    ph = host.read(_ffi.FIELD_OBJ_PARAMS)[:, :nd]
    assert (ph[..., 1] == 8).all() and (ph[..., 0] == 0.02).all()
    host.close()


# ---- the distributions of the reference's reset(), as constants (pinned against the reference's own files by
# tests/test_oracle_vs_reference.py::test_dr_constants_match_the_reference when /root/reference is present)
DR_CONFIG = {   # randomization/config/default_dr.json == randomizer.py:8-16 DEFAULT_CONFIG
    "horz_mode": {"type": "int", "low": 0, "high": 4},
    "light_pos": {"type": "uniform", "low": [-150, 170, -150], "high": [150, 220, 150], "size": 3},
    "camera_noise": {"type": "uniform", "low": -0.005, "high": 0.005, "size": 3},
    "trim": {"type": "normal", "loc": 0, "scale": 0.02},
    "camera_height": {"type": "uniform", "low": 0.92, "high": 1.08},
    "camera_angle": {"type": "uniform", "low": 0.8, "high": 1.2},
    "camera_fov_y": {"type": "uniform", "low": 0.8, "high": 1.2},
}
CAMERA_FLOOR_DIST, CAMERA_ANGLE, CAMERA_FOV_Y, WHEEL_DIST = 0.108, 19.15, 75.0, 0.102   # simulator.py:119-131, 137


def _uniform_ok(x, lo, hi, tol=0.02):
    """samples of U(lo, hi): inside the interval, mean and spread of a uniform (N >= 16k: 2 % of the width is > 5 sigma)"""
    x = np.asarray(x, np.float64)
    w = hi - lo
    e = 1e-6 * max(abs(lo), abs(hi), 1e-3)
    assert x.min() >= lo - e and x.max() <= hi + e, (x.min(), x.max(), lo, hi)
    assert abs(x.mean() - 0.5 * (lo + hi)) < tol * w and abs(x.std() - w / np.sqrt(12)) < tol * w
    assert x.min() < lo + 0.01 * w and x.max() > hi - 0.01 * w


def test_device_sampler_draws_the_reference_dr_distributions():
    """simulator.py:546-614 + randomizer.py:36-91 + _perturb :1065-1085: every domain-randomised quantity the device
    sampler draws, read back for 32768 envs and checked against the reference's ranges."""
    N = 32768
    sky, gnd = (0.45, 0.82, 1.0), (0.15, 0.15, 0.15)
    sim = BatchedSimulator("small_loop", N, render=False, domain_rand=True, seed=11, device_reset=True)
    cam, col, wd = sim.read(_ffi.FIELD_CAMERA), sim.read(_ffi.FIELD_COLORS), sim.read(_ffi.FIELD_WHEEL_DIST)
    c = DR_CONFIG
    _uniform_ok(cam[:, 0], CAMERA_FLOOR_DIST * c["camera_height"]["low"], CAMERA_FLOOR_DIST * c["camera_height"]["high"])
    _uniform_ok(np.degrees(cam[:, 1]), CAMERA_ANGLE * c["camera_angle"]["low"], CAMERA_ANGLE * c["camera_angle"]["high"])
    _uniform_ok(np.degrees(cam[:, 2]), CAMERA_FOV_Y * c["camera_fov_y"]["low"], CAMERA_FOV_Y * c["camera_fov_y"]["high"])
    for k in range(3):
        _uniform_ok(cam[:, 3 + k], c["camera_noise"]["low"], c["camera_noise"]["high"])
        _uniform_ok(col[:, 12 + k], c["light_pos"]["low"][k], c["light_pos"]["high"][k])
        _uniform_ok(col[:, 3 + k], gnd[k] * 0.7, gnd[k] * 1.3)          # ground_color = _perturb(color_ground, 0.3)
        _uniform_ok(col[:, 6 + k], 0.25 * 0.7, 0.25 * 1.3)              # ambient = _perturb(0.5 * DIM, 0.3), DIM = 0.5
        _uniform_ok(col[:, 9 + k], 0.35 * 0.01, 0.35 * 1.99)            # diffuse = _perturb(0.7 * DIM, 0.99)
    assert (col[:, 15] == 0).all()                                      # 3-component light_pos -> directional
    _uniform_ok(wd, WHEEL_DIST * 0.9, WHEEL_DIST * 1.1)                 # _perturb(WHEEL_DIST)
    # horizon colour: horz_mode in {0..3} uniformly, each a perturbed base colour (simulator.py:551-560)
    modes = [(sky, 0.1), ((0.64, 0.71, 0.28), 0.1), ((0.15,) * 3, 0.4), ((0.9,) * 3, 0.4)]
    hit = np.zeros((N, 4), bool)
    for m, (base, s) in enumerate(modes):
        b = np.array(base)
        hit[:, m] = ((col[:, 0:3] >= b * (1 - s) - 1e-6) & (col[:, 0:3] <= b * (1 + s) + 1e-6)).all(axis=1)
    assert hit.any(axis=1).all()
    only = hit & (hit.sum(axis=1, keepdims=True) == 1)                  # boxes of modes 0 / 3 overlap a little: count the unambiguous ones
    frac = only.sum(axis=0) / N
    assert (frac > 0.17).all() and (frac < 0.27).all(), frac
    sim.close()
    # without domain randomisation nothing is perturbed
    sim = BatchedSimulator("small_loop", 64, render=False, domain_rand=False, seed=11, device_reset=True)
    cam, col, wd = sim.read(_ffi.FIELD_CAMERA), sim.read(_ffi.FIELD_COLORS), sim.read(_ffi.FIELD_WHEEL_DIST)
    assert np.allclose(cam[:, 0], CAMERA_FLOOR_DIST) and np.allclose(np.degrees(cam[:, 1]), CAMERA_ANGLE, atol=1e-4)
    assert np.allclose(col[:, 0:3], sky, atol=1e-6) and np.allclose(col[:, 12:16], [0, 3, 0, 1]) and (wd == WHEEL_DIST).all()
    sim.close()


@pytest.mark.parametrize("map_name", ["small_loop_only_duckies", "loop_pedestrians"])
def test_device_reset_spawns_pass_the_oracles_acceptance_test(map_name):
    """The spawn acceptance test of simulator.py:692-738, evaluated by the ORACLE (not by the library's own query
    kernel) on poses the device sampler produced."""
    from util import make_oracle
    from oracle import sim as osim
    N = 512
    sim = BatchedSimulator(map_name, N, render=False, domain_rand=False, seed=9, device_reset=True, accept_start_angle_deg=60)
    pos, ang = sim.read(_ffi.FIELD_POS), sim.read(_ffi.FIELD_ANGLE)
    o = make_oracle(map_name, domain_rand=False, seed=0)
    for e in range(0, N, 2):
        p, a = pos[e], float(ang[e])
        assert not o._inconvenient_spawn(p), e
        assert o._valid_pose(p, a, safety_factor=1.3), e
        lp = o.get_lane_pos2(p, a)                         # raises NotInLane if the sampler accepted an off-lane pose
        assert -60 < lp[2] < 60, (e, lp)
    sim.close()


def test_device_sampler_randomises_follower_duckiebots():
    """DuckiebotObj.__init__ under domain randomisation (objects.py:198-207): follow_dist ~ U(0.3, 0.4), velocity ~
    U(0.05, 0.15), gain = 2 + U(-0.3, 0.3), trim = 0 + U(-0.1, 0.1) + 2, radius / wheel_dist / robot size jittered."""
    N = 16384
    sim = BatchedSimulator("loop_dyn_duckiebots", N, render=False, domain_rand=True, seed=5, device_reset=True)
    mt = sim.maps[0]
    par, ext = sim.read(_ffi.FIELD_OBJ_PARAMS), sim.read(_ffi.FIELD_OBJ_EXTRA)
    bots = [o.dyn_slot for o in mt.objects if o.dyn_kind == 2]
    assert len(bots) == 4
    for d in bots:
        _uniform_ok(par[:, d, 0], 0.05, 0.15)
        _uniform_ok(par[:, d, 1], 2.0 - 0.3, 2.0 + 0.3)
        _uniform_ok(par[:, d, 2], 2.0 - 0.1, 2.0 + 0.1)
        _uniform_ok(ext[:, d, 0], 0.3, 0.4)
        _uniform_ok(ext[:, d, 1], 0.0318 - 0.0002, 0.0318 + 0.0002)
        _uniform_ok(ext[:, d, 2], 0.102 - 0.01, 0.102 + 0.01)
        _uniform_ok(ext[:, d, 3], 0.15 - 0.01, 0.15 + 0.01)
        _uniform_ok(ext[:, d, 4], 0.18 - 0.01, 0.18 + 0.01)
    sim.close()
    sim = BatchedSimulator("loop_dyn_duckiebots", 8, render=False, domain_rand=False, seed=5, device_reset=True)
    par, ext = sim.read(_ffi.FIELD_OBJ_PARAMS), sim.read(_ffi.FIELD_OBJ_EXTRA)
    for d in bots:                                        # the non-DR branch (objects.py:208-216)
        assert np.allclose(par[:, d], [0.1, 2.0, 0.0]) and np.allclose(ext[:, d], [0.3, 0.0318, 0.102, 0.15, 0.18])
    sim.close()


def test_device_sampler_honours_the_maps_start_pose():
    """simulator.py:679-688: a map with `start_pose` spawns at start_tile * tile_size + pose, deterministically -- on the
    device sampler as on the host path."""
    from dtsim import assets
    md = assets.get_map("small_loop")
    md["start_tile"] = [1, 2]
    md["start_pose"] = [[0.35, 0.0, 0.29], 1.5707]
    host = BatchedSimulator("small_loop", 4, map_data=md, render=False, domain_rand=False, seed=1)
    dev = BatchedSimulator("small_loop", 4, map_data=md, render=False, domain_rand=False, seed=1, device_reset=True)
    want = np.array([1 * 0.585 + 0.35, 0.0, 2 * 0.585 + 0.29])
    for s in (host, dev):
        assert np.allclose(s.read(_ffi.FIELD_POS), want, atol=1e-12) and np.allclose(s.read(_ffi.FIELD_ANGLE), 1.5707)
        s.close()
