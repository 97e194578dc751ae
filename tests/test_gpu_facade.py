"""GPU: the drop-in gym_duckietown surface (Simulator / DuckietownEnv / MultiMapEnv, N=1 views of the
HIP library) behaves like the reference's API, incl. the properties run_tests.py asserts
(run_tests.py:10-52), and the batched MultiMap slot alternation (envs/multimap_env.py:44-49)."""
import numpy as np
import pytest

from dtsim import BatchedSimulator, _ffi
from oracle import sim as osim
from util import make_oracle
from dtsim import distortion as pdist

pytestmark = pytest.mark.gpu


def test_duckietown_env_api_and_run_tests_properties():
    from gym_duckietown.envs import DuckietownEnv
    from gym_duckietown.simulator import get_agent_corners, NotInLane
    env = DuckietownEnv(map_name="small_loop_only_duckies", domain_rand=False, seed=4, camera_width=160,
                        camera_height=120, full_transparency=True)
    first_obs = env.reset()
    assert first_obs.shape == env.observation_space.shape == (120, 160, 3) and first_obs.dtype == np.uint8
    m0, m1 = first_obs.mean(), env.render("rgb_array").mean()
    assert 0 < m0 < 255 and abs(m0 - m1) < 5                        # run_tests.py:17-22
    o = make_oracle("small_loop_only_duckies", domain_rand=False, seed=4)
    o.reset()                      # like the reference, the constructor already reset once
    assert np.array_equal(env.cur_pos, o.cur_pos) and env.cur_angle == o.cur_angle
    for i in range(10):
        obs, reward, done, info = env.step(np.array([0.4, 0.3]))
        r, d, _ = o.step_vel_steer(np.array([0.4, 0.3]))
        assert obs.shape == first_obs.shape and done == d and abs(reward - r) < 1e-9
        assert "Simulator" in info and "DuckietownEnv" in info and info["Simulator"]["tile_coords"] == list(o.map.get_grid_coords(o.cur_pos))
        assert abs(info["Simulator"]["lane_position"]["dist"] - o.get_lane_pos2(o.cur_pos, o.cur_angle)[0]) < 1e-9
    assert env.step_count == 10 and abs(env.timestamp - 10 / 30) < 1e-12
    # query methods evaluate on the device and agree with the oracle
    pos, ang = env.cur_pos, env.cur_angle
    assert env._valid_pose(pos, ang) == o._valid_pose(pos, ang)
    assert env._collision(get_agent_corners(pos, ang)) == o._collision(osim.get_agent_corners(pos, ang))
    assert abs(env.proximity_penalty2(pos, ang) - o.proximity_penalty2(pos, ang)) < 1e-9
    pt, tg = env.closest_curve_point(pos, ang)
    opt, otg = o.closest_curve_point(pos, ang)
    assert np.allclose(pt, opt, atol=1e-9) and np.allclose(tg, otg, atol=1e-9)
    assert env.get_grid_coords(pos) == o.map.get_grid_coords(pos)
    with pytest.raises(NotInLane):
        env.get_lane_pos2(np.array([0.1, 0, 0.1]), 0.0)
    assert env._get_tile(1, 1)["kind"] == "curve_right" and env._get_tile(9, 9) is None
    d = env._compute_done_reward()
    assert d.done_code == "in-progress" and isinstance(d.reward, float)
    env.close()


def test_spawn_is_collision_free_like_run_tests():
    """run_tests.py:47-52: no collision at spawn nor after one step of [0.05, 0]."""
    from gym_duckietown.envs import DuckietownEnv
    from gym_duckietown.simulator import get_agent_corners
    env = DuckietownEnv(map_name="loop_only_duckies", seed=0, camera_width=84, camera_height=84)
    for _ in range(15):
        env.reset()
        assert not env._collision(get_agent_corners(env.cur_pos, env.cur_angle)), "collision on spawn"
        env.step(np.array([0.05, 0]))
        assert not env._collision(get_agent_corners(env.cur_pos, env.cur_angle)), "collision after one step"


def test_multimap_env_round_robin():
    from gym_duckietown.envs import MultiMapEnv
    env = MultiMapEnv(domain_rand=False, seed=1, camera_width=84, camera_height=84)
    names = []
    for _ in range(4):
        env.reset()
        names.append(env.env_list[env.cur_env_idx].map_name)
        obs, r, d, info = env.step(np.array([0.2, 0.0]))
        assert obs.shape == (84, 84, 3)
    assert names == ["small_loop_only_duckies", "loop_only_duckies"] * 2     # first reset selects index 1
    env.close()


def test_batched_multimap_slot_alternation():
    """BASELINE config C5 semantics on one GPU: every slot alternates between the two maps per reset."""
    N = 8
    sim = BatchedSimulator(["loop_only_duckies", "small_loop_only_duckies"], N, map_cycle=True, domain_rand=False,
                           seed=10, camera_width=84, camera_height=84)
    assert (sim.read(_ffi.FIELD_MAP_ID) == 1).all()                # first reset -> index 1
    oracles = [make_oracle("small_loop_only_duckies", domain_rand=False, seed=10 + e) for e in range(N)]
    pos = sim.read(_ffi.FIELD_POS)
    for e, o in enumerate(oracles):
        assert np.array_equal(pos[e], o.cur_pos)
    mask = np.zeros(N, np.uint8); mask[::2] = 1
    sim.reset(mask)
    assert sim.read(_ffi.FIELD_MAP_ID).tolist() == [0, 1] * (N // 2)
    sim.step(np.full((N, 2), 0.3, np.float32))
    sim.render()
    fr = sim.frames_host()
    assert fr.shape == (N, 84, 84, 3) and 0 < fr.mean() < 255
    tiles = sim.read(_ffi.FIELD_TILE)
    assert (tiles[::2, 0] < 8).all() and (tiles[1::2, 0] < 5).all()
    sim.close()


def test_segmentation_render_like_test_segmentation_py():
    """test_segmentation.py:164-186: `reset()`, `render(segment=True)`, `step`, alternating `render(segment=...)`;
    the segmented view keeps the geometry of the camera view: magenta where the normal frame shows sky / ground,
    black asphalt, paint and flat-coloured objects elsewhere."""
    from gym_duckietown.envs import DuckietownEnv
    env = DuckietownEnv(map_name="loop_only_duckies", domain_rand=False, max_steps=10**6, seed=2, camera_width=160, camera_height=120)
    obs = env.reset()
    win = env.render("rgb_array", segment=True)                  # the 800x600 window image
    assert win.shape == (600, 800, 3) and (win == np.array([255, 0, 255], np.uint8)).all(-1).any()
    seg = env.render_obs(segment=True)
    assert seg.shape == obs.shape and seg.dtype == np.uint8
    sky = (np.abs(obs.astype(int) - np.round(np.array(env.horizon_color) * 255)).max(-1) <= 1)
    mag = (seg == np.array([255, 0, 255], np.uint8)).all(-1)
    assert (mag[sky]).mean() > 0.95 and mag.mean() < 0.9
    assert ((seg == 0).all(-1)).mean() > 0.1                      # blanked asphalt / grass
    for i in range(4):
        obs, _, _, _ = env.step(np.array([0.3, 0.0]))
        img = env.render_obs(segment=(i % 2 == 0))
        assert (i % 2 == 0) == bool((img == np.array([255, 0, 255], np.uint8)).all(-1).any())
    assert np.array_equal(env.reset(segment=True), env.render_obs(segment=True))
    env.close()


def test_render_modes_window_views_match_oracle():
    """render(mode) (simulator.py:1974-2003): 800x600 views of the current state.  "rgb_array" / "free_cam" are the
    agent camera (with / without the fisheye); "top_down" is gluLookAt((a, H, b), (a, 0, b - 0.01), +y) with the
    agent's mesh at its pose (:1786-1798, 1920-1927).  Checked against the oracle raster on the same inputs."""
    from gym_duckietown.envs import DuckietownEnv
    from gym_duckietown.simulator import WINDOW_WIDTH as WW, WINDOW_HEIGHT as WH
    from dtsim import assets
    from oracle import raster
    import math
    envd = DuckietownEnv(map_name="small_loop_only_duckies", domain_rand=False, seed=9, distortion=True)
    envd.reset()
    imgd, freed = envd.render("rgb_array"), envd.render("free_cam")
    assert imgd.shape == freed.shape == (WH, WW, 3)
    assert (np.abs(imgd.astype(int) - freed.astype(int)).max(-1) > 8).mean() > 0.02       # the fisheye moves pixels; free_cam has none
    envd.close()
    env = DuckietownEnv(map_name="small_loop_only_duckies", domain_rand=False, seed=9, camera_width=160, camera_height=120)
    env.reset()
    for _ in range(5):
        env.step(np.array([0.4, 0.2]))
    obs = env.render_obs()
    img = env.render("rgb_array")
    assert img.shape == (WH, WW, 3) and img.dtype == np.uint8
    assert abs(float(obs.mean()) - float(img.mean())) < 5                      # run_tests.py:17-22
    free = env.render("free_cam")
    top = env.render("top_down")
    assert np.array_equal(free, img) and top.shape == (WH, WW, 3)

    om = osim.OracleMap(assets.get_map("small_loop_only_duckies"), __import__("util").EXT)
    kinds = {t["kind"] for t in om.grid if t is not None}
    meshes = {"duckie": assets.get_mesh("duckie"), "*": assets.get_mesh("*")}
    scene = raster.Scene(om, {k: assets.get_texture(k) for k in kinds}, meshes)
    st = env._sim.init_states[0]
    states = [dict(pos=o.pos, y_rot=o.y_rot, visible=True) for o in om.objects]
    cam = raster.Camera(env.cur_pos, env.cur_angle, width=WW, height=WH, horizon_color=list(st.horizon_color),
                        ground_color=list(st.ground_color), light_pos=list(st.light_pos))
    mode = __import__("util").oracle_mode(next(v for (_d, sz), v in env._viewers.items() if sz == (WW, WH)))   # the window views come from a per-env-camera handle
    ref = raster.render_obs(cam, scene, mode, None, obj_states=states)
    d = np.abs(free.astype(int) - ref.astype(int)).max(-1)
    assert (d > 1).mean() <= 2e-3 and np.abs(free.astype(int) - ref.astype(int)).mean() <= 0.03, ((d > 1).mean(),)

    # top-down: oracle camera straight from the reference's gluLookAt arguments
    a, b = env.grid_width * env.road_tile_size / 2, env.grid_height * env.road_tile_size / 2
    Hf = (max(a, b) + 0.1) / math.tan(math.radians(75.0) / 2)
    tcam = raster.Camera([a, 0.0, b + 0.066], math.pi / 2, cam_height=Hf, cam_angle_deg=math.degrees(math.atan2(Hf, 0.01)),
                         width=WW, height=WH, horizon_color=list(st.horizon_color), ground_color=list(st.ground_color), light_pos=list(st.light_pos))
    assert np.allclose(tcam.C, [a, Hf, b], atol=1e-12)
    fwd = np.array([0.0, -Hf, -0.01]) / math.hypot(Hf, 0.01)                   # gluLookAt forward
    assert np.allclose(tcam.to_eye(tcam.C + fwd), [0, 0, -1], atol=1e-9)       # eye space looks down -z
    assert np.allclose(tcam.to_eye(tcam.C + np.array([1.0, 0, 0])), [1, 0, 0], atol=1e-9)
    # the agent marker: an extra duckiebot (stand-in mesh, 0.12 m tall) at cur_pos, rotated by cur_angle
    import copy
    md = copy.deepcopy(assets.get_map("small_loop_only_duckies"))
    md["objects"] = list(md["objects"]) + [{"kind": "duckiebot", "pos": [0.5, 0.5], "rotate": 0, "static": False, "height": 0.12}]
    om2 = osim.OracleMap(md, __import__("util").EXT)
    scene2 = raster.Scene(om2, scene.textures, meshes)
    states2 = states + [dict(pos=env.cur_pos, y_rot=math.degrees(env.cur_angle), visible=True)]
    rmap = None
    tref = raster.render_obs(tcam, scene2, mode, rmap, obj_states=states2)
    d = np.abs(top.astype(int) - tref.astype(int)).max(-1)
    assert (d > 1).mean() <= 3e-3 and np.abs(top.astype(int) - tref.astype(int)).mean() <= 0.05, ((d > 1).mean(),)
    no_agent = raster.render_obs(tcam, scene2, mode, rmap, obj_states=states + [dict(states2[-1], visible=False)])
    assert (np.abs(tref.astype(int) - no_agent.astype(int)).max(-1) > 0).sum() > 30     # the marker is in the picture
    seg = env.render("top_down", segment=True)
    assert (seg == np.array([255, 0, 255], np.uint8)).all(-1).mean() > 0.05
    env.close()


def test_randomize_maps_on_reset():
    """simulator.py:373-378, 541-544: every reset first draws a map (np_random.choice == integers(0, n)) and reloads it."""
    from gym_duckietown.envs import DuckietownEnv
    from gym_duckietown.simulator import get_agent_corners
    from dtsim import assets
    env = DuckietownEnv(randomize_maps_on_reset=True, seed=3, domain_rand=False, camera_width=160, camera_height=120)
    names = env.map_names
    assert names == sorted(assets.MAPS) and len(names) >= 3
    assert env.map_name == names[int(np.random.default_rng(3).integers(0, len(names)))]      # the constructor's reset
    seen = set()
    for _ in range(24):
        obs = env.reset()
        seen.add(env.map_name)
        md = assets.get_map(env.map_name)
        assert (env.grid_height, env.grid_width) == (len(md["tiles"]), len(md["tiles"][0]))
        assert obs.shape == (120, 160, 3) and 0 < obs.mean() < 255
        assert env._valid_pose(env.cur_pos, env.cur_angle, 1.3)
        assert not env._collision(get_agent_corners(env.cur_pos, env.cur_angle))
        env.step(np.array([0.3, 0.3]))
    assert len(seen) >= 3
    env.close()


def test_batched_random_maps_device_sampler():
    from dtsim import assets
    names = sorted(assets.MAPS)
    N = 1024
    sim = BatchedSimulator(names, N, render=False, domain_rand=False, seed=5, device_reset=True, map_random=True)
    m0 = sim.read(_ffi.FIELD_MAP_ID).copy()
    cnt = np.bincount(m0, minlength=len(names))
    assert cnt.min() > 0.6 * N / len(names)                      # uniform over the maps
    sim.reset()
    m1 = sim.read(_ffi.FIELD_MAP_ID)
    assert (m1 != m0).mean() > 0.5                               # a fresh draw per reset
    pos, ang = sim.read(_ffi.FIELD_POS), sim.read(_ffi.FIELD_ANGLE)
    pr = sim.query(np.arange(N, dtype=np.int32), np.stack([pos[:, 0], pos[:, 2], ang], 1), safety_factor=1.3)
    assert pr["valid"].all() and pr["in_lane"].all() and not pr["inconvenient"].any()
    # the reload re-creates the walking duckies: none is mid-walk right after a reset, even where the map did not change
    sim.step(np.zeros((300, N, 2), np.float32), n_steps=300)
    sim.reset()
    assert not sim.read(_ffi.FIELD_OBJ_ACTIVE).any()
    sim.close()


def test_update_physics_and_update_pos_follow_the_reference_semantics():
    """simulator.py:1551-1584: update_physics is ONE physics update (frame_skip belongs to step(), :1674) that advances
    step_count / timestamp / speed / the objects; simulator.py:2076-2088: the module-level _update_pos only integrates the
    dynamics state and returns the pose -- no counters, no objects, no reward."""
    from gym_duckietown.simulator import Simulator, _update_pos
    env = Simulator(map_name="loop_pedestrians", domain_rand=False, seed=3, frame_skip=3, camera_width=64, camera_height=48)
    o = make_oracle("loop_pedestrians", domain_rand=False, seed=3, frame_skip=3)
    assert np.array_equal(env.cur_pos, o.cur_pos)     # both constructors reset once
    a = np.array([0.5, 0.3])
    # step(): frame_skip updates
    env.step(a)
    o.step(a)
    assert env.step_count == o.step_count == 3 and np.allclose(env.cur_pos, o.cur_pos, atol=1e-9)
    # update_physics(): exactly one
    env.update_physics(a)
    o.update_physics(a)
    assert env.step_count == o.step_count == 4 and abs(env.timestamp - o.timestamp) < 1e-12
    assert np.allclose(env.cur_pos, o.cur_pos, atol=1e-9) and abs(env.cur_angle - o.cur_angle) < 1e-9
    assert abs(env.speed - o.speed) < 1e-9
    assert np.allclose(env.wheelVels, a * env.robot_speed) and np.allclose(env.last_action, a)
    d, r, _ = o._compute_done_reward()
    dr = env._compute_done_reward()
    assert dr.done == d and abs(dr.reward - r) < 1e-9
    obj_c = env._sim.read(_ffi.FIELD_OBJ_CENTER)[0].copy()
    # _update_pos(): pose only
    sc, ts, rew, spd = env.step_count, env.timestamp, env._compute_done_reward().reward, env.speed
    pos, ang = _update_pos(env, a)
    o.state.integrate(o.delta_time, a[0], a[1])
    opos = np.asarray([o.state.x, 0, o.map.grid_height * o.map.tile_size - o.state.y])
    assert np.allclose(pos, opos, atol=1e-9) and abs(ang - o.state.angle()) < 1e-9
    assert env.step_count == sc and env.timestamp == ts and env.speed == spd
    assert env._compute_done_reward().reward == rew                      # not recomputed
    assert np.array_equal(env._sim.read(_ffi.FIELD_OBJ_CENTER)[0], obj_c)  # objects not stepped
    env.cur_pos, env.cur_angle = pos, ang                               # the reference's call pattern (:1558)
    assert np.array_equal(env.cur_pos, pos) and env.cur_angle == ang
    # after a pose write the reference derives reward / done / lane from the NEW pose on demand (:1685, :1586)
    o.cur_pos, o.cur_angle = opos, o.state.angle()
    d2, r2, _ = o._compute_done_reward()
    dr2 = env._compute_done_reward()
    assert dr2.done == d2 and abs(dr2.reward - r2) < 1e-9
    off = np.array([-0.5, 0.0, -0.5])                                   # off the grid: invalid pose
    env.cur_pos = off
    o.cur_pos = off
    d3, r3, _ = o._compute_done_reward()
    dr3 = env._compute_done_reward()
    assert d3 and dr3.done and dr3.reward == r3 == -1000 and dr3.done_code == "invalid-pose"
    env.close()


def test_update_physics_takes_the_wheel_pair_unclipped_in_every_action_mode():
    """simulator.py:1551: update_physics integrates the wheel pair it is handed -- the (vel, steering) kinematics are
    DuckietownEnv.step's (envs/duckietown_env.py:36-61) and np.clip(action, -1, 1) is Simulator.step's (:1670).  So
    DuckietownEnv.update_physics([l, r]) must equal Simulator.update_physics([l, r]), also beyond [-1, 1]."""
    from gym_duckietown.envs import DuckietownEnv
    env = DuckietownEnv(map_name="small_loop", domain_rand=False, seed=4, camera_width=64, camera_height=48)
    o = make_oracle("small_loop", domain_rand=False, seed=4)
    assert np.array_equal(env.cur_pos, o.cur_pos)
    for a in (np.array([0.6, 0.2]), np.array([1.4, -1.2]), np.array([-0.3, 0.9])):
        for _ in range(4):
            env.update_physics(a)
            o.update_physics(a)
        assert np.allclose(env.cur_pos, o.cur_pos, atol=1e-9) and abs(env.cur_angle - o.cur_angle) < 1e-9, a
        assert np.array_equal(env.last_action, a) and np.allclose(env.wheelVels, a * env.robot_speed)
        assert np.allclose(env._sim.read(_ffi.FIELD_WHEELS)[0], a)
    env.close()


def test_undistort_property_skips_the_fisheye_like_the_reference():
    """simulator.py:1968-1970: env.undistort = True (set by UndistortWrapper, wrappers.py:209) makes render_obs return
    the rectilinear image; the wrapper then remaps that.  Same bytes as the fisheye-free render / as the folded
    BatchedSimulator(undistort=True) path."""
    from gym_duckietown.simulator import Simulator
    from gym_duckietown.wrappers import UndistortWrapper
    kw = dict(map_name="small_loop_only_duckies", domain_rand=False, seed=5, camera_width=160, camera_height=120)
    env = Simulator(distortion=True, **kw)
    plain = Simulator(distortion=False, **kw)
    fish = env.render_obs().copy()
    rect = plain.render_obs().copy()
    assert not np.array_equal(fish, rect)
    env.undistort = True
    assert np.array_equal(env.render_obs(), rect)
    env.undistort = False
    assert np.array_equal(env.render_obs(), fish)
    w = UndistortWrapper(env)
    assert env.undistort is True
    folded = BatchedSimulator(kw["map_name"], 1, domain_rand=False, seed=5, camera_width=160, camera_height=120,
                              distortion=True, undistort=True, per_env_camera=True)   # (the render path the facade uses: its light is per env)
    folded.render()
    assert np.array_equal(w.observation(env.render_obs()), folded.frames_host()[0])
    env.close(); plain.close()


def test_actuation_delay_follows_the_frame_rate():
    """The DB18 delay is 0.15 s of simulated time (simulator.py:745-755): 5 steps at 30 Hz, 2 at 10 Hz, 9 at 60 Hz --
    not a fixed step count."""
    for fr, k in ((30, 5), (10, 2), (20, 3), (60, 9)):
        sim = BatchedSimulator("small_loop", 2, domain_rand=False, seed=1, frame_rate=fr, render=False, actions_f64=True)
        assert sim.delay_steps == k
        o = make_oracle("small_loop", domain_rand=False, seed=1, frame_rate=fr)
        assert o.delay_steps == k
        assert np.array_equal(sim.read(_ffi.FIELD_POS)[0], o.cur_pos)
        for i in range(12):
            a = np.array([0.6, 0.2 + 0.05 * i])
            sim.step(np.tile(a, (2, 1)))
            o.step(a)
            assert np.allclose(sim.read(_ffi.FIELD_POS)[0], o.cur_pos, atol=1e-9)
    with pytest.raises(ValueError):
        BatchedSimulator("small_loop", 1, frame_rate=120, render=False)     # 18 steps > DTSIM_MAX_DELAY = 16


def test_vel_steer_env_reports_the_wheel_duties_as_last_action():
    """DuckietownEnv.step hands [u_l, u_r] to Simulator.step (envs/duckietown_env.py:61), which keeps them as
    last_action / wheelVels (simulator.py:1555,1564) and reports them in info['Simulator']['action']."""
    from gym_duckietown.envs import DuckietownEnv
    env = DuckietownEnv(map_name="small_loop", domain_rand=False, seed=2, camera_width=64, camera_height=48)
    o = make_oracle("small_loop", domain_rand=False, seed=2)
    act = np.array([0.7, -1.3])
    _, _, _, info = env.step(act)
    wheels = np.clip(o.wheels_from_vel_steer(act), -1, 1)
    assert np.allclose(info["Simulator"]["action"], wheels, atol=1e-12)
    assert np.allclose(env.last_action, wheels, atol=1e-12) and np.allclose(env.wheelVels, wheels * env.robot_speed, atol=1e-12)
    env.close()


def test_read_agent_matches_the_field_reads():
    """dtsim_read_agent (one transfer) against the per-field reads it replaces in the 1-env gym loop."""
    from dtsim import BatchedSimulator, _ffi
    sim = BatchedSimulator("loop_dyn_duckiebots", 5, render=False, domain_rand=False, seed=3)
    sim.step(np.random.default_rng(1).uniform(0.1, 0.9, (7, 5, 2)).astype(np.float32), n_steps=7)
    for e in (0, 3, 4):
        a = sim.read_agent(e)
        assert np.array_equal(np.array(a.pos), sim.read(_ffi.FIELD_POS)[e])
        assert a.angle == sim.read(_ffi.FIELD_ANGLE)[e] and a.speed == sim.read(_ffi.FIELD_SPEED)[e]
        assert a.timestamp == sim.read(_ffi.FIELD_TIMESTAMP)[e] and a.step_count == sim.read(_ffi.FIELD_STEP_COUNT)[e]
        assert np.array_equal(np.array(a.wheels), sim.read(_ffi.FIELD_WHEELS)[e])
        assert np.array_equal(np.array(a.lane), sim.read(_ffi.FIELD_LANE)[e])
        assert a.prox == sim.read(_ffi.FIELD_PROX)[e] and a.reward == sim.read(_ffi.FIELD_REWARD)[e]
        assert np.array_equal(np.array(a.tile), sim.read(_ffi.FIELD_TILE)[e])
        assert (a.in_lane, a.done, a.done_code) == (sim.read(_ffi.FIELD_IN_LANE)[e], sim.read(_ffi.FIELD_DONE)[e], sim.read(_ffi.FIELD_DONE_CODE)[e])
    with pytest.raises(Exception):
        sim.read_agent(5)
    sim.close()


def test_allgather_frames_through_the_c_abi_one_rank():
    """SURVEY 8(b)'s dtsim_allgather_frames: the RCCL all-gather of the frame batch enqueued by the library on its own stream,
    with a communicator the caller made (here: a 1-rank ncclComm_t from librccl through ctypes -- the N > 1 path is the same
    call with more ranks; unmeasured on hardware).  Checks the frame batch and an observe() buffer come back byte for byte."""
    import ctypes as C
    import torch
    from dtsim import BatchedSimulator
    try:
        rccl = C.CDLL("librccl.so.1")
    except OSError:
        pytest.skip("librccl.so.1 not present")

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    N = 6
    sim = BatchedSimulator("small_loop", N, camera_width=160, camera_height=120, distortion=False, domain_rand=False, seed=3)
    sim.step(np.full((N, 2), 0.4, np.float32))
    sim.render()
    recv = torch.zeros((1, N, 120, 160, 3), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()                             # `recv` must be ready before the library's stream writes it (dtsim.h)
    sim.allgather_frames(comm.value, recv)
    sim.sync()
    assert np.array_equal(recv[0].cpu().numpy(), sim.frames_host())
    obs = sim.observe(60, 80)
    recv2 = torch.zeros((1, N, 60, 80, 3), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    sim.allgather_frames(comm.value, recv2, send=obs)
    sim.sync()
    assert np.array_equal(recv2[0].cpu().numpy(), torch.as_tensor(obs, device="cuda:0").cpu().numpy())
    sim.close()
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    rccl.ncclCommDestroy(comm)


def test_sharded_exchange_on_the_device_one_rank():
    """ShardedSimulator.step_render_gather on the real simulator (world_size 1: no transfer, but the whole produce path -- the render
    pass bound into the slot's slice through dtsim_bind_frames, dtsim_observe into the slot, slot rotation, flush): every batch handed
    out for step t equals what a second simulator with the same seed renders / observes at step t."""
    import torch
    from dtsim import BatchedSimulator
    from dtsim.sharding import ShardedSimulator
    N, W, H, T = 6, 160, 120, 4
    kw = dict(camera_width=W, camera_height=H, distortion=True, domain_rand=False, seed=11)
    acts = np.random.default_rng(2).uniform(0.2, 0.8, (T, N, 2)).astype(np.float32)
    ref = BatchedSimulator("small_loop", N, **kw)
    want_f, want_o = [], []
    for t in range(T):
        ref.step(acts[t]); ref.render()
        want_f.append(ref.frames_host().copy())
        o = ref.observe(60, 80)
        ref.sync()                                        # dtsim_observe runs on the library's stream, the copy below on torch's
        want_o.append(torch.as_tensor(o, device="cuda:0").cpu().numpy().copy())
    ref.close()
    for what, want, obs in (("frames", want_f, None), ("observe", want_o, (60, 80))):
        ss = ShardedSimulator.wrap(BatchedSimulator("small_loop", N, **kw), N, 0, 1)
        got = {}
        for t in range(T):
            k, batch = ss.step_render_gather(acts[t], overlap=True, dst=0, local_actions=True, what=what, obs=obs)
            if t == 0:
                assert k is None and batch is None
            else:
                assert k == t - 1
                got[k] = batch.cpu().numpy().copy()
        k, batch = ss.flush_gather(dst=0)
        assert k == T - 1
        got[k] = batch.cpu().numpy().copy()
        for t in range(T):
            assert np.array_equal(got[t], want[t]), (what, t)
        # the blocking form delivers the step it just made
        k, batch = ss.step_render_gather(acts[0], overlap=False, dst=0, local_actions=True, what=what, obs=obs)
        assert batch is not None and batch.shape[0] == N
        ss.sim.close()


def test_sharded_gather_frames_uses_the_library_allgather_one_rank():
    """ShardedSimulator.gather_frames() on RCCL goes through dtsim_allgather_frames (one exchange implementation on the all-gather
    path): enqueued on the simulator's stream behind the render pass, torch's stream ordered behind it by an event, no host wait.
    Exercised here with a 1-rank communicator (force_collective; the N > 1 call is the same one with more ranks -- unmeasured on
    hardware): the gathered batch is the frame batch, byte for byte, and a held batch survives the next render + gather."""
    import torch
    from dtsim import BatchedSimulator
    from dtsim.sharding import ShardedSimulator
    N = 6
    sim = BatchedSimulator("small_loop", N, camera_width=160, camera_height=120, distortion=True, domain_rand=False, seed=3)
    ss = ShardedSimulator.wrap(sim, N, 0, 1)
    ss.force_collective = True
    try:
        ss._rccl_comm(None)
    except OSError:
        pytest.skip("librccl not present")
    acts = np.random.default_rng(5).uniform(0.2, 0.8, (3, N, 2)).astype(np.float32)
    kept = []
    for t in range(3):
        sim.step(acts[t]); sim.render()
        out = ss.gather_frames()                          # no sim.sync() in between: the ordering is the streams'
        assert tuple(out.shape) == (N, 120, 160, 3)
        kept.append((out, None))
        host = out.cpu().numpy()                          # torch's stream: behind the library's all-gather by the event
        assert np.array_equal(host, sim.frames_host()), t
        kept[-1] = (out, host)
    for out, host in kept:                                # every batch is its own tensor
        assert np.array_equal(out.cpu().numpy(), host)
    sim.close()


def _two_rank_worker(rank, world, port, n, q):
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    try:
        from dtsim import BatchedSimulator
        from dtsim.sharding import ShardedSimulator
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        total, W, H, T = world * n, 160, 120, 5
        kw = dict(camera_width=W, camera_height=H, distortion=True, domain_rand=False)
        ss = ShardedSimulator("small_loop", total, seed=40, device=0, rank=rank, world=world, **kw)
        acts = np.random.default_rng(2).uniform(0.2, 0.8, (T, total, 2)).astype(np.float32)
        got = {}
        for what, obs in (("frames", None), ("observe", (60, 80))):
            held = None
            for t in range(T):
                k, batch = ss.step_render_gather(acts[t:t + 1], overlap=True, dst=0, what=what, obs=obs)
                if rank == 0 and held is not None:
                    assert np.array_equal(held[1].numpy() if not held[1].is_cuda else held[1].cpu().numpy(), got[(what, held[0])])   # still intact one call later
                if batch is not None:
                    got[(what, k)] = batch.cpu().numpy().copy()
                    held = (k, batch)
            k, batch = ss.flush_gather(dst=0)
            if rank == 0:
                got[(what, k)] = batch.cpu().numpy().copy()
            ss.sim.render()                               # (the loop rendered into the slots: the same state into the library's own buffer)
            allf = ss.gather_frames()                     # all-gather on gloo: host tensors
            if rank == 0:
                got[(what, "all")] = allf.cpu().numpy().copy()
        if rank == 0:
            np.savez(q, **{f"{a}_{b}": v for (a, b), v in got.items()})
        ss.sim.close()
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_two_ranks_on_one_gpu_exchange_the_real_batches(tmp_path):
    """world_size 2 on ONE GPU (gloo moves the payload through pinned host memory: the staged path of dtsim/sharding.py): both ranks run
    the real simulator on their half of the envs and the overlapped gather-to-root; what rank 0 receives for step t equals what ONE
    simulator with all the envs renders / observes at step t -- env ranges, per-env seeds, action slices, slot rotation and the bytes.
    (The RCCL transport itself needs two GPUs: unmeasured on hardware.)"""
    import socket
    import torch
    import torch.multiprocessing as mp
    from dtsim import BatchedSimulator
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    world, n = 2, 5
    out = str(tmp_path / "root.npz")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_two_rank_worker, args=(r, world, port, n, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    got = np.load(out)
    total, T = world * n, 5
    ref = BatchedSimulator("small_loop", total, camera_width=160, camera_height=120, distortion=True, domain_rand=False, seed=40)
    acts = np.random.default_rng(2).uniform(0.2, 0.8, (T, total, 2)).astype(np.float32)
    for what in ("frames", "observe"):
        for t in range(T):
            ref.step(acts[t]); ref.render()
            if what == "frames":
                want = ref.frames_host()
            else:
                o = ref.observe(60, 80)
                ref.sync()                                # dtsim_observe runs on the library's stream, the copy below on torch's
                want = torch.as_tensor(o, device="cuda:0").cpu().numpy()
            assert np.array_equal(got[f"{what}_{t}"], want), (what, t)
        assert np.array_equal(got[f"{what}_all"], ref.frames_host()), what
    ref.close()


def test_reset_captures_the_light_through_the_last_frame_s_model_view():
    """GL transforms GL_POSITION by the model-view current at the glLightfv call; reset() (simulator.py:565-584) issues it with what the last
    _render_img left: the identity at the first reset (inside __init__), the last frame's camera of the previous episode afterwards.  The facade
    hands the device the eye-space light accordingly; frames of the second episode against the oracle lit the same way.  Domain randomisation:
    the drawn DIRECTION is rotated only.  gl_light_capture=False keeps the light as given."""
    from gym_duckietown.simulator import Simulator
    from oracle import raster
    from test_gpu_render import _camera, _scene, _stats
    W, H = 320, 240
    scene = _scene("small_loop")
    for dr in (False, True):
        env = Simulator(map_name="small_loop", domain_rand=dr, camera_width=W, camera_height=H, seed=6, distortion=False)
        first = [float(v) for v in env._sim.init_states[0].light_pos]
        if not dr:
            assert first == [0.0, 3.0, 0.0, 1.0]                               # nothing drawn before the first reset: the identity
        for _ in range(7):
            env.step(np.array([0.6, 0.3]))
        cam_prev = _camera(env._sim, 0, W, H, dr)                              # the camera of the last frame of episode 1
        obs = env.reset()
        st = env._sim.init_states[0]
        got = [float(v) for v in st.light_pos]
        raw = [0.0, 3.0, 0.0, 1.0] if not dr else [float(v) for v in env.randomization_settings["light_pos"]] + [0.0]
        want = cam_prev.to_eye(np.asarray(raw[:3])) if raw[3] else cam_prev.normal_to_eye(np.asarray(raw[:3]))
        assert got[3] == raw[3] and np.allclose(got[:3], want, rtol=0, atol=1e-9), (got, want)
        assert not np.allclose(got[:3], raw[:3], atol=1e-3)                    # it did move
        col = env._sim.read(_ffi.FIELD_COLORS)[0]
        assert np.allclose(col[12:16], got, rtol=1e-6, atol=1e-6)
        ref = raster.render_obs(_camera(env._sim, 0, W, H, dr), scene, __import__("util").oracle_mode(env._sim), None)   # (reads the init state's light)
        s = _stats(obs, ref)
        assert s["mean"] <= 0.05 and s["frac_gt2"] <= 1e-3, (dr, s)
        env.close()
    env = Simulator(map_name="small_loop", domain_rand=False, camera_width=W, camera_height=H, seed=6, distortion=False, gl_light_capture=False)
    env.step(np.array([0.6, 0.3])); env.reset()
    assert [float(v) for v in env._sim.init_states[0].light_pos] == [0.0, 3.0, 0.0, 1.0]
    env.close()
