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
