"""GPU parity: HIP step/query/reset path (through the C-ABI) vs the CPU oracle.

Bar (BASELINE.json north_star): done/collision/valid flags, done codes and tile indices
BIT-EXACT; pose / reward / lane / proximity within FLOAT_TOL (float64 path; the only
differences are libm (ocml vs glibc) ulps in sin/cos/atan2/acos and BLAS-vs-scalar
dot-product rounding inside numpy).
"""
import numpy as np
import pytest

from dtsim import BatchedSimulator, _ffi
from oracle import sim as osim
from util import EXT, init_state_from_oracle, make_oracle, random_poses

pytestmark = pytest.mark.gpu
FLOAT_TOL = 1e-9


def _assert_queries(sim, o, poses, sf):
    """dtsim_query vs the oracle for every pose: flags / tile / curve index bit-exact, floats within FLOAT_TOL."""
    m = o.map
    pr = sim.query(np.zeros(len(poses), np.int32), poses, safety_factor=sf)
    nflag = nlane = 0
    for q, (x, z, a) in enumerate(poses):
        pos = np.array([x, 0, z])
        i, j = m.get_grid_coords(pos)
        assert (pr["tile_i"][q], pr["tile_j"][q]) == (i, j)
        assert bool(pr["drivable"][q]) == o._drivable_pos(pos)
        assert bool(pr["collision"][q]) == o._collision(osim.get_agent_corners(pos, a))
        assert bool(pr["valid"][q]) == o._valid_pose(pos, a, sf)
        assert bool(pr["inconvenient"][q]) == o._inconvenient_spawn(pos)
        nflag += int(pr["collision"][q])
        assert abs(pr["prox"][q] - o.proximity_penalty2(pos, a)) <= FLOAT_TOL
        try:
            lp = o.get_lane_pos2(pos, a)
            assert pr["in_lane"][q] == 1
            assert pr["curve_idx"][q] == o._last_curve[0] and pr["t"][q] == o._last_curve[1]
            assert np.allclose([pr["dist"][q], pr["dot_dir"][q], pr["angle_rad"][q]], [lp[0], lp[1], lp[3]],
                               rtol=0, atol=FLOAT_TOL)
            assert abs(pr["angle_deg"][q] - lp[2]) <= 1e-7
            nlane += 1
        except osim.NotInLane:
            assert pr["in_lane"][q] == 0
        assert abs(pr["reward"][q] - o.compute_reward(pos, a, o.robot_speed)) <= 1e-8
    return nflag, nlane


@pytest.mark.parametrize("map_name", ["small_loop", "small_loop_only_duckies", "loop_only_duckies", "loop_pedestrians"])
def test_query_matches_oracle(map_name):
    sim = BatchedSimulator(map_name, 2, render=False, domain_rand=False, seed=3)
    o = make_oracle(map_name, do_reset=False)
    m = o.map
    rng = np.random.default_rng(11)
    cents = np.array([[ob.pos[0], ob.pos[2]] for ob in m.objects]) if m.objects else None
    poses = random_poses(rng, m.grid_width, m.grid_height, m.tile_size, 4000, cents)
    for sf in (1.0, 1.3):
        nflag, _ = _assert_queries(sim, o, poses, sf)
        if m.objects:
            assert nflag > 10  # the sample does exercise collisions
    sim.close()


def test_query_on_every_tile_kind_and_orientation():
    """straight / curve_left / curve_right / 3way_left / 3way_right in all four orientations and 4way (2, 6 and 12
    curves per tile, simulator.py:1151-1335); the oracle is pinned against the reference on this very map
    (tests/test_oracle_vs_reference.py)."""
    import copy
    from util import junction_map, EXT
    md = junction_map()
    sim = BatchedSimulator("junctions", 2, map_data=copy.deepcopy(md), render=False, domain_rand=False, seed=3)
    o = osim.OracleSim(copy.deepcopy(md), EXT, do_reset=False)
    m = o.map
    rng = np.random.default_rng(13)
    cents = np.array([[ob.pos[0], ob.pos[2]] for ob in m.objects])
    poses = random_poses(rng, m.grid_width, m.grid_height, m.tile_size, 6000, cents)
    nflag, nlane = _assert_queries(sim, o, poses, 1.0)
    assert nflag > 10 and nlane > 2000
    sim.close()


def test_query_matches_oracle_on_real_assets():
    """Collision / safety-circle / spawn geometry with OBB extents that come from parsed OBJ meshes
    (non-symmetric after ObjMesh's recentring quirk, `height` and `scale` forms, an `optional` object)."""
    import os
    from dtsim import assets
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "assets")
    lib = assets.AssetLibrary(root)
    md = lib.map_data("test_town")
    ext = {"*": (assets.get_mesh("*").min_coords, assets.get_mesh("*").max_coords)}
    for desc in md["objects"]:
        m = lib.object_mesh(desc)[1]
        ext[desc["kind"]] = (m.min_coords, m.max_coords)
    sim = BatchedSimulator("test_town", 2, asset_root=root, render=False, domain_rand=False, seed=3)
    o = osim.OracleSim(md, ext, do_reset=False)
    m = o.map
    rng = np.random.default_rng(12)
    cents = np.array([[ob.pos[0], ob.pos[2]] for ob in m.objects])
    poses = random_poses(rng, m.grid_width, m.grid_height, m.tile_size, 3000, cents)
    pr = sim.query(np.zeros(len(poses), np.int32), poses, safety_factor=1.0)
    nflag = 0
    for q, (x, z, a) in enumerate(poses):
        pos = np.array([x, 0, z])
        assert (pr["tile_i"][q], pr["tile_j"][q]) == tuple(m.get_grid_coords(pos))
        assert bool(pr["drivable"][q]) == o._drivable_pos(pos)
        assert bool(pr["collision"][q]) == o._collision(osim.get_agent_corners(pos, a))
        assert bool(pr["valid"][q]) == o._valid_pose(pos, a, 1.0)
        assert bool(pr["inconvenient"][q]) == o._inconvenient_spawn(pos)
        assert abs(pr["prox"][q] - o.proximity_penalty2(pos, a)) <= FLOAT_TOL
        nflag += int(pr["collision"][q])
    assert nflag > 10
    sim.close()


@pytest.mark.parametrize("map_name,dr", [("small_loop", False), ("small_loop_only_duckies", True),
                                         ("loop_only_duckies", False), ("loop_only_duckies", True)])
def test_reset_rng_order_matches_oracle(map_name, dr):
    """Host RNG order + device geometry == the oracle's (reference-pinned) reset, twice in a row."""
    N = 12
    sim = BatchedSimulator(map_name, N, render=False, domain_rand=dr, seed=100)
    oracles = [make_oracle(map_name, domain_rand=dr, seed=100 + e) for e in range(N)]
    for rep in range(3):
        pos, ang = sim.read(_ffi.FIELD_POS), sim.read(_ffi.FIELD_ANGLE)
        for e, o in enumerate(oracles):
            assert np.array_equal(pos[e], o.cur_pos), (rep, e, pos[e], o.cur_pos)
            assert ang[e] == o.cur_angle
            st = sim.init_states[e]
            assert st.wheel_dist == float(o.wheel_dist)
            assert list(st.horizon_color) == [float(v) for v in o.horizon_color]
            assert st.cam_fov_y_deg == float(np.asarray(o.cam_fov_y).reshape(-1)[0])
            assert sim.env_state[e].spawn_attempts == o.spawn_attempts
        sim.reset()
        for o in oracles:
            o.reset()
    sim.close()


@pytest.mark.parametrize("case", ["user_tile_start", "start_tile", "start_pose", "optional"])
def test_reset_tile_and_pose_branches_match_oracle(case):
    """reset()'s tile / pose selection (simulator.py:659-688) and the optional objects' visibility draws (:648-656)
    on the host RNG-order path; the oracle is pinned on these branches against the reference
    (tests/test_oracle_vs_reference.py::test_reset_start_tile_branches)."""
    import copy
    from dtsim import assets
    from util import junction_map, EXT
    md, kw = copy.deepcopy(assets.get_map("loop_only_duckies")), {}
    if case == "user_tile_start":
        kw = dict(user_tile_start=(1, 0))
    elif case == "start_tile":
        md["start_tile"] = [2, 0]
    elif case == "start_pose":
        md["start_tile"] = [1, 0]
        md["start_pose"] = [[0.3, 0, 0.25], 1.2]
    else:
        md = junction_map()
        md["tiles"] = [[("grass" if c == "empty" else c) for c in row] for row in md["tiles"]]   # as in the reference pin
    N = 4
    sim = BatchedSimulator("case", N, map_data=copy.deepcopy(md), render=False, domain_rand=True, seed=50, **kw)
    oracles = [osim.OracleSim(copy.deepcopy(md), EXT, domain_rand=True, seed=50 + e, **kw) for e in range(N)]
    for rep in range(2):
        pos, ang = sim.read(_ffi.FIELD_POS), sim.read(_ffi.FIELD_ANGLE)
        vis = sim.read(_ffi.FIELD_OBJ_VISIBLE)
        for e, o in enumerate(oracles):
            assert np.array_equal(pos[e], o.cur_pos), (case, rep, e, pos[e], o.cur_pos)
            assert ang[e] == o.cur_angle
            assert [bool(v) for v in vis[e][:len(o.map.objects)]] == [bool(ob.visible) for ob in o.map.objects]
        sim.reset()
        for o in oracles:
            o.reset()
    sim.close()


def _run_traj(map_name, mode, N, T, seed, frame_skip=1, max_steps=1500, dr=False, actions_f64=False):
    sim = BatchedSimulator(map_name, N, render=False, domain_rand=dr, seed=seed, action_mode=mode,
                           frame_skip=frame_skip, max_steps=max_steps, actions_f64=actions_f64)
    oracles = [make_oracle(map_name, domain_rand=dr, seed=seed + e, frame_skip=frame_skip, max_steps=max_steps)
               for e in range(N)]
    rng = np.random.default_rng(1234)
    acts = rng.uniform(-1, 1, size=(T, N, 2)).astype(np.float64 if actions_f64 else np.float32)
    if mode == "vel_steer":
        acts[..., 0] = np.abs(acts[..., 0]) * 0.6 + 0.1   # mostly forward so episodes last
    alive = np.ones(N, bool)
    n_done = 0
    max_err = 0.0
    for t in range(T):
        sim.step(acts[t])
        pos, ang, rew = sim.read(_ffi.FIELD_POS), sim.read(_ffi.FIELD_ANGLE), sim.read(_ffi.FIELD_REWARD)
        done, code, tile = sim.read(_ffi.FIELD_DONE), sim.read(_ffi.FIELD_DONE_CODE), sim.read(_ffi.FIELD_TILE)
        lane, inl, prox = sim.read(_ffi.FIELD_LANE), sim.read(_ffi.FIELD_IN_LANE), sim.read(_ffi.FIELD_PROX)
        speed, sc = sim.read(_ffi.FIELD_SPEED), sim.read(_ffi.FIELD_STEP_COUNT)
        for e, o in enumerate(oracles):
            if not alive[e]:
                continue
            a = acts[t, e].astype(np.float64)
            r, d, c = (o.step_vel_steer(a) if mode == "vel_steer" else o.step(a))
            inf = o.info()
            assert bool(done[e]) == d and code[e] == c, (t, e, done[e], d, code[e], c)
            assert tuple(tile[e]) == inf["tile"]
            assert bool(inl[e]) == inf["in_lane"]
            assert sc[e] == inf["step_count"]
            err = max(np.abs(pos[e] - inf["pos"]).max(), abs(ang[e] - inf["angle"]), abs(rew[e] - r),
                      abs(prox[e] - inf["prox"]), abs(lane[e, 0] - inf["lane"][0]), abs(lane[e, 1] - inf["lane"][1]),
                      abs(lane[e, 3] - inf["lane"][3]), abs(speed[e] - inf["speed"]) * 1e-2)
            max_err = max(max_err, err)
            assert err <= FLOAT_TOL, (t, e, err)
            if d:
                alive[e] = False
                n_done += 1
        if not alive.any():
            break
    sim.close()
    return n_done, max_err


@pytest.mark.parametrize("map_name,mode", [("small_loop", "vel_steer"), ("small_loop_only_duckies", "vel_steer"),
                                           ("loop_only_duckies", "wheels"), ("small_loop", "wheels")])
def test_trajectory_matches_oracle(map_name, mode):
    n_done, max_err = _run_traj(map_name, mode, N=24, T=400, seed=1000)
    assert n_done >= 1            # invalid-pose terminations are exercised
    print(f"{map_name}/{mode}: episodes ended {n_done}, max abs err {max_err:.3e}")


def test_trajectory_frame_skip_max_steps_f64_actions():
    n_done, _ = _run_traj("small_loop", "vel_steer", N=8, T=60, seed=7, frame_skip=3, max_steps=40, actions_f64=True)
    assert n_done == 8            # everyone hits invalid-pose or max-steps within 60 steps


def test_fused_steps_equal_single_steps():
    """dtsim_step(n_steps=K) == K x dtsim_step(1) bit for bit."""
    N, T = 64, 50
    acts = np.random.default_rng(5).uniform(-1, 1, (T, N, 2)).astype(np.float32)
    a = BatchedSimulator("small_loop", N, render=False, domain_rand=False, seed=9)
    b = BatchedSimulator("small_loop", N, render=False, domain_rand=False, seed=9)
    for t in range(T):
        a.step(acts[t])
    b.step(acts, n_steps=T)
    assert np.array_equal(a.read(_ffi.FIELD_STATE_BLOB), b.read(_ffi.FIELD_STATE_BLOB))
    a.close(); b.close()


def test_dynamic_duckies_match_oracle():
    """loop_pedestrians stand-in: DuckieObj walk/wait state machine + collisions with it."""
    N, T = 4, 620
    sim = BatchedSimulator("loop_pedestrians", N, render=False, domain_rand=False, seed=21, max_steps=100000)
    oracles = [make_oracle("loop_pedestrians", domain_rand=False, seed=21 + e, max_steps=100000) for e in range(N)]
    zero = np.zeros((N, 2), np.float32)
    rng = np.random.default_rng(2)
    reversed_seen = False
    for t in range(T):
        sim.step(zero)
        for o in oracles:
            o.step(np.zeros(2))
        reversed_seen = reversed_seen or any(ob.vel < 0 for ob in oracles[0].map.objects)
        if t % 20 == 0 or 238 <= t <= 275:
            cen, act, yrot = sim.read(_ffi.FIELD_OBJ_CENTER), sim.read(_ffi.FIELD_OBJ_ACTIVE), sim.read(_ffi.FIELD_OBJ_YROT)
            for e, o in enumerate(oracles):
                for d, ob in enumerate(o.map.objects):
                    assert np.array_equal(cen[e, d], np.asarray(ob.center, float)[[0, 2]]), (t, e, d)
                    assert bool(act[e, d]) == ob.pedestrian_active
                    assert abs(yrot[e, d] - ob.y_rot) <= 1e-9
            # probe collisions / proximity around the (moving) duckies
            ob = oracles[0].map.objects[t % 8]
            poses = np.stack([ob.center[0] + rng.uniform(-0.2, 0.2, 64), ob.center[2] + rng.uniform(-0.2, 0.2, 64),
                              rng.uniform(-3, 3, 64)], axis=1)
            pr = sim.query(np.zeros(64, np.int32), poses)
            o = oracles[0]
            for q, (x, z, a) in enumerate(poses):
                pos = np.array([x, 0, z])
                assert bool(pr["collision"][q]) == o._collision(osim.get_agent_corners(pos, a))
                assert abs(pr["prox"][q] - o.proximity_penalty2(pos, a)) <= FLOAT_TOL
    assert reversed_seen   # a walk finished and reversed
    sim.close()


def test_dynamic_duckiebots_match_oracle():
    """DuckiebotObj followers (objects.py:180-336): pure pursuit + own kinematics, stale SAT axes."""
    N, T = 3, 900
    sim = BatchedSimulator("loop_dyn_duckiebots", N, render=False, domain_rand=False, seed=31, max_steps=100000)
    oracles = [make_oracle("loop_dyn_duckiebots", domain_rand=False, seed=31 + e, max_steps=100000) for e in range(N)]
    zero = np.zeros((N, 2), np.float32)
    rng = np.random.default_rng(4)
    moved = 0.0
    for t in range(T):
        sim.step(zero)
        for o in oracles:
            o.step(np.zeros(2))
        if t % 25 == 0 or t == T - 1:
            cen, yrot = sim.read(_ffi.FIELD_OBJ_CENTER), sim.read(_ffi.FIELD_OBJ_YROT)
            for e, o in enumerate(oracles):
                bots = [ob for ob in o.map.objects if not ob.static]
                for d, ob in enumerate(bots):
                    assert np.abs(cen[e, d] - np.asarray(ob.pos, float)[[0, 2]]).max() <= 1e-9, (t, e, d)
                    assert abs(yrot[e, d] - ob.y_rot) <= 1e-7
            o = oracles[0]
            ob = [x for x in o.map.objects if not x.static][t % 4]
            poses = np.stack([ob.pos[0] + rng.uniform(-0.25, 0.25, 48), ob.pos[2] + rng.uniform(-0.25, 0.25, 48),
                              rng.uniform(-3, 3, 48)], axis=1)
            pr = sim.query(np.zeros(48, np.int32), poses)
            for q, (x, z, a) in enumerate(poses):
                pos = np.array([x, 0, z])
                assert bool(pr["collision"][q]) == o._collision(osim.get_agent_corners(pos, a))
                assert bool(pr["valid"][q]) == o._valid_pose(pos, a)
                assert abs(pr["prox"][q] - o.proximity_penalty2(pos, a)) <= FLOAT_TOL
    start = np.array([[3.5, 1.7], [1.3, 3.0], [6.7, 2.5], [2.5, 5.3]]) * 0.585
    moved = np.abs(sim.read(_ffi.FIELD_OBJ_CENTER)[0, :4] - start).max()
    assert moved > 0.5            # the followers actually drove along the loop
    sim.close()


def test_auto_reset_from_pool_and_checkpoint():
    N = 32
    sim = BatchedSimulator("small_loop", N, render=False, domain_rand=False, seed=50, auto_reset=True, max_steps=30)
    pool = sim.make_spawn_pool(4 * N)
    sim.reset()
    acts = np.random.default_rng(8).uniform(-1, 1, (100, N, 2)).astype(np.float32)
    sim.step(acts[:40], n_steps=40)
    ep = sim.read(_ffi.FIELD_EPISODE)
    assert (ep >= 1).all()                       # max_steps=30 forces at least one auto reset
    sc = sim.read(_ffi.FIELD_STEP_COUNT)
    assert (sc <= 30).all()
    # restarted envs came from pool[(e + ep*N) % n_pool]; run a fresh env from that state
    blob = sim.read(_ffi.FIELD_STATE_BLOB)
    sim.step(acts[40:60], n_steps=20)
    after = sim.read(_ffi.FIELD_STATE_BLOB)
    sim.write(_ffi.FIELD_STATE_BLOB, blob)        # checkpoint / resume
    sim.step(acts[40:60], n_steps=20)
    assert np.array_equal(after, sim.read(_ffi.FIELD_STATE_BLOB))
    sim.close()


def test_error_paths():
    import ctypes as C
    lib = _ffi.load()
    cfg = _ffi.Config()
    h = C.c_void_p()
    cfg.struct_size = 12
    assert lib.dtsim_create(C.byref(cfg), C.byref(h)) == _ffi.E_INVALID
    assert b"ABI" in lib.dtsim_last_error()
    sim = BatchedSimulator("small_loop", 2, render=False, domain_rand=False, do_reset=False)
    with pytest.raises(_ffi.DtsimError) as ei:
        sim.step(np.zeros((2, 2), np.float32))
    assert ei.value.code == _ffi.E_STATE
    with pytest.raises(_ffi.DtsimError):
        sim.render()
    sim.close()


def test_large_map_with_many_objects_matches_oracle():
    """Maximum-size style inputs: a 30x30 grid (900 tiles, mixed kinds incl. 3way / 4way / empty cells) with 56 static
    and 8 walking duckies -- the per-map limits of include/dtsim.h -- against the oracle's geometry."""
    from dtsim import assets
    G = "grass"
    W = H = 30
    tiles = [[G] * W for _ in range(H)]
    for k in range(2, 28):
        tiles[2][k] = "straight/E"; tiles[27][k] = "straight/E"; tiles[k][2] = "straight/S"; tiles[k][27] = "straight/S"
    tiles[2][2], tiles[2][27], tiles[27][27], tiles[27][2] = "curve_left/W", "curve_left/N", "curve_left/E", "curve_left/S"
    for k in range(3, 27):
        tiles[14][k] = "straight/E"
    tiles[14][2], tiles[14][27] = "3way_left/S", "3way_left/N"
    tiles[14][14] = "4way"
    for k in range(3, 14):
        tiles[k][14] = "straight/S"
    tiles[2][14] = "3way_left/W"
    tiles[20][20] = "empty"
    rng = np.random.default_rng(2)
    objs = [dict(kind="duckie", pos=[float(rng.uniform(1, 29)), float(rng.uniform(1, 29))], rotate=float(rng.uniform(0, 360)),
                 height=0.06, static=True) for _ in range(56)]
    objs += [dict(kind="duckie", pos=[float(3 + 3 * k), 14.4], rotate=90.0, height=0.06, static=False) for k in range(8)]
    md = dict(tiles=tiles, objects=objs, tile_size=0.585)
    sim = BatchedSimulator("big", 2, map_data=md, render=False, domain_rand=False, seed=3, do_reset=False)
    o = osim.OracleSim(md, EXT, do_reset=False)
    m = o.map
    poses = random_poses(rng, m.grid_width, m.grid_height, m.tile_size, 3000,
                         np.array([[ob.pos[0], ob.pos[2]] for ob in m.objects]))
    st = (_ffi.InitState * 2)()
    for e in range(2):
        st[e].pos[:] = [2.5 * 0.585, 0.0, 2.5 * 0.585]; st[e].angle = 0.0; st[e].wheel_dist = 0.102
        st[e].cam_height, st[e].cam_angle_deg, st[e].cam_fov_y_deg = 0.108, 19.15, 75.0
    sim.reset(states=st)
    pr = sim.query(np.zeros(len(poses), np.int32), poses, safety_factor=1.0)
    n_coll = 0
    for q, (x, z, a) in enumerate(poses):
        pos = np.array([x, 0, z])
        assert (pr["tile_i"][q], pr["tile_j"][q]) == tuple(m.get_grid_coords(pos))
        assert bool(pr["drivable"][q]) == o._drivable_pos(pos)
        assert bool(pr["collision"][q]) == o._collision(osim.get_agent_corners(pos, a))
        assert bool(pr["valid"][q]) == o._valid_pose(pos, a, 1.0)
        assert abs(pr["prox"][q] - o.proximity_penalty2(pos, a)) <= FLOAT_TOL
        try:
            lp = o.get_lane_pos2(pos, a)
            assert pr["in_lane"][q] == 1 and abs(pr["dist"][q] - lp[0]) <= FLOAT_TOL
        except osim.NotInLane:
            assert pr["in_lane"][q] == 0
        n_coll += int(pr["collision"][q])
    assert n_coll > 10
    sim.close()


def test_checkerboard_motion_matches_oracle():
    """CheckerboardObj (objects.py:479-587): scripted motion incl. the vertical excursion; the collision box stays
    at the initial pose, proximity uses the moving 3-D centre."""
    from dtsim import assets
    md = assets.get_map("small_loop")
    md["objects"] = [dict(kind="checkerboard", pos=[2.4, 1.35], rotate=0, height=0.2, static=False),
                     dict(kind="duckie", pos=[1.75, 2.5], rotate=60, height=0.06, static=True)]
    N = 3
    sim = BatchedSimulator("cb", N, map_data=md, render=False, domain_rand=False, seed=2, max_steps=10**6)
    o = osim.OracleSim(md, EXT, do_reset=False)
    ob = o.map.objects[0]
    assert sim.maps[0].objects[0].dyn_kind == 3
    rng = np.random.default_rng(5)
    T = 500
    for t in range(T):
        sim.step(np.zeros((1, N, 2), np.float32))
        ob.step(1 / 30)
        if t % 25 == 0 or t in (19, 20, 67, 155, 156, 219, 220):
            cen = sim.read(_ffi.FIELD_OBJ_CENTER)[0, 0]
            cy = sim.read(_ffi.FIELD_OBJ_Y)[0, 0]
            assert cen[0] == ob.center[0] and cen[1] == ob.center[2] and cy == ob.center[1], (t, cen, cy, ob.center)
            poses = random_poses(rng, 5, 5, 0.585, 60, np.array([[ob.center[0], ob.center[2]]]))
            pr = sim.query(np.zeros(len(poses), np.int32), poses, safety_factor=1.0)
            for q, (x, z, a) in enumerate(poses):
                pos = np.array([x, 0, z])
                assert abs(pr["prox"][q] - o.proximity_penalty2(pos, a)) <= FLOAT_TOL
                assert bool(pr["collision"][q]) == o._collision(osim.get_agent_corners(pos, a))
                assert bool(pr["inconvenient"][q]) == o._inconvenient_spawn(pos)
    assert ob.center[1] == 0.0 or abs(ob.center[1]) < 0.2
    sim.close()
