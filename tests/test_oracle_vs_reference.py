"""`not gpu`, build container only: pin the oracle against the reference's OWN code, imported
through oracle/refstub.py (skipped where /root/reference does not exist, e.g. the GPU box)."""
import copy

import numpy as np
import pytest

from dtsim import assets
from oracle import refstub, sim as osim
from util import EXT, junction_map

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


def test_segmentation_rules_equal_the_reference(tmp_path):
    """graphics.py:59-67 should_segment_out and objmesh.py:260-266 gen_segmentation_color (a closure inside
    ObjMesh.__init__, observed through the colour it hands to load_texture) against dtsim/assets.py."""
    ns = refstub.load()
    paths = ["tiles-processed/photos/straight/texture.jpg", "tiles-processed/photos/curve_left/texture.jpg",
             "tiles-processed/photos/curve_right/texture.jpg", "tiles-processed/photos/3way_left/texture.jpg",
             "tiles-processed/photos/4way/texture.jpg", "tiles-processed/photos/asphalt/texture.jpg",
             "tiles-processed/photos/grass/texture.jpg", "tiles-processed/photos/floor/texture.jpg",
             "tiles-processed/synthetic/calibration/texture.png", "sign_left_T_intersect.png", "sign_4_way_intersect.png",
             "trafficlight_card0.jpg", "duckie.png", "black_tile.png", "/data/highway/left.png", "bus.png", "/home/asphalt/straight.png"]
    for p in paths:
        assert assets.should_segment_out(p) == ns.graphics.should_segment_out(p), p
    obj = "v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvn 0 0 1\nf 1/1/1 2/1/1 3/1/1\n"
    for name in ["duckie", "duckiebot", "sign_generic", "trafficlight", "tree", "house", "bus", "cone", "truck", "barrier",
                 "building", "object"]:
        f = tmp_path / f"{name}.obj"
        f.write_text(obj)
        (tmp_path / "black_tile.png").write_bytes(b"")
        res = {"black_tile.png": str(tmp_path / "black_tile.png")}
        r = refstub.ref_objmesh(str(f), name, lambda bn, _r=res: _r.get(bn), segment=True)
        (path, seg, col), = r["textures"]                          # untextured chunk -> the black_tile hack with the name's colour
        assert seg is True and path.endswith("black_tile.png")
        assert list(col) == assets.gen_segmentation_color(name), name


def test_every_tile_kind_and_orientation_against_the_reference():
    """Curves, drivable flags, lane pose, valid pose, collision and proximity on a map holding every drivable tile kind
    in every orientation -- the loop fixtures only have straights and curves."""
    md = junction_map()
    r, ns = _ref("junctions", md=md)
    o = osim.OracleSim(copy.deepcopy(md), EXT, do_reset=False)
    # curve tables: same control points, tile by tile
    for j in range(r.grid_height):
        for i in range(r.grid_width):
            t = r._get_tile(i, j)
            ot = o.map.get_tile(i, j)
            if t is None:
                assert ot is None
                continue
            assert t["drivable"] == ot["drivable"] and t["kind"] == ot["kind"] and t["angle"] == ot["angle"]
            if t["drivable"]:
                assert np.array_equal(np.asarray(t["curves"]), np.asarray(ot["curves"])), (i, j, t["kind"], t["angle"])
    rng = np.random.default_rng(5)
    n_lane = 0
    for _ in range(3000):
        pos = np.array([rng.uniform(-0.3, r.grid_width * 0.585 + 0.3), 0, rng.uniform(-0.3, r.grid_height * 0.585 + 0.3)])
        a = rng.uniform(-7, 7)
        assert r._drivable_pos(pos) == o._drivable_pos(pos)
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
        n_lane += lp is not None
        cp, ct = r.closest_curve_point(pos, a)
        ocp, oct = o.closest_curve_point(pos, a)
        assert (cp is None) == (ocp is None)
        if cp is not None:
            assert np.array_equal(cp, ocp) and np.array_equal(ct, oct)
    assert n_lane > 1000


def test_wrappers_equal_the_reference():
    """src/gym_duckietown/wrappers.py run through the stub loader: the action wrappers' arithmetic, the observation
    transpose, and the calibration constants UndistortWrapper / Distortion carry (cv2 itself is absent: the remaps stay
    restatements)."""
    import gym_duckietown.wrappers as mine
    from dtsim import distortion as pdist
    ns = refstub.load()
    ref = ns.wrappers

    class Env:
        wheel_dist = 0.1093
        distortion = True
        undistort = False
        observation_space = None
        action_space = None

        @property
        def unwrapped(self):
            return self

    rng = np.random.default_rng(4)
    for a in range(3):
        assert list(ref.DiscreteWrapper(Env()).action(a)) == list(mine.DiscreteWrapper(Env()).action(a))
    for kw in ({}, dict(gain=1.3, trim=0.07, radius=0.03, k=25.0, limit=0.8)):
        r, m = ref.SteeringToWheelVelWrapper(Env(), **kw), mine.SteeringToWheelVelWrapper(Env(), **kw)
        for _ in range(200):
            act = rng.uniform(-1.5, 1.5, 2)
            assert np.array_equal(np.asarray(r.action(act), dtype=np.float64), np.asarray(m.action(act), dtype=np.float64))
    obs = rng.integers(0, 256, (6, 8, 3), dtype=np.uint8)
    po = ref.PyTorchObsWrapper.__new__(ref.PyTorchObsWrapper)
    assert np.array_equal(po.observation(obs), mine.PyTorchObsWrapper.observation(None, obs))
    e = Env()
    u = ref.UndistortWrapper(e)
    assert e.undistort is True
    assert np.array_equal(u.camera_matrix, mine.UndistortWrapper.K) and np.array_equal(u.camera_matrix, pdist.CAMERA_MATRIX)
    assert np.array_equal(np.ravel(u.distortion_coefs), mine.UndistortWrapper.D) and np.array_equal(np.ravel(u.distortion_coefs), pdist.DIST_COEFS)
    assert np.array_equal(u.projection_matrix, mine.UndistortWrapper.P) and np.array_equal(u.projection_matrix[:, :3], pdist.UNDISTORT_P)
    ns.distortion.cv2.getOptimalNewCameraMatrix = lambda **kw: (None, None)      # cv2 is a stub here
    d = ns.distortion.Distortion()
    assert np.array_equal(d.camera_matrix, pdist.CAMERA_MATRIX) and np.array_equal(np.ravel(d.distortion_coefs), pdist.DIST_COEFS)
    assert (d.W, d.H) == (640, 480)


def test_duckietown_env_kinematics_equal_the_reference():
    """envs/duckietown_env.py:36-61: (vel, steering) -> [left, right] duty handed to Simulator.step, and the
    `DuckietownEnv` info block; against the oracle's wheels_from_vel_steer (which the HIP k_step is checked against)."""
    ns = refstub.load()
    DE = ns.duckietown_env.DuckietownEnv
    seen = {}

    def fake_step(self, vels):
        seen["vels"] = np.array(vels, dtype=np.float64)
        return None, 0.0, False, {}

    orig = ns.simulator.Simulator.step
    ns.simulator.Simulator.step = fake_step
    try:
        rng = np.random.default_rng(8)
        for kw in (dict(gain=1.0, trim=0.0, radius=0.0318, k=27.0, limit=1.0), dict(gain=1.4, trim=-0.05, radius=0.03, k=24.0, limit=0.7)):
            env = DE.__new__(DE)
            for k, v in kw.items():
                setattr(env, k, v)
            env.wheel_dist = 0.102 * 1.04
            env.unwrapped = env                    # gym.Env.unwrapped (the stubbed gym.Env has none)
            o = osim.OracleSim(assets.get_map("small_loop"), EXT, do_reset=False, **kw)
            o.wheel_dist = env.wheel_dist
            for _ in range(300):
                act = rng.uniform(-1.6, 1.6, 2)
                _, _, _, info = env.step(act)
                mine = o.wheels_from_vel_steer(act)
                assert np.array_equal(seen["vels"], np.asarray(mine, dtype=np.float64))
                assert set(info["DuckietownEnv"]) == {"k", "gain", "train", "radius", "omega_r", "omega_l"}
    finally:
        ns.simulator.Simulator.step = orig


@pytest.mark.parametrize("dr", [False, True])
def test_reset_start_tile_branches(dr):
    """reset()'s tile / pose selection branches (simulator.py:659-688): `user_tile_start`, the map's `start_tile`,
    the map's `start_pose`; optional objects' visibility draw (:648-656) on the junction map."""
    base = assets.get_map("loop_only_duckies")
    cases = []
    cases.append((copy.deepcopy(base), dict(user_tile_start=(1, 0))))
    md = copy.deepcopy(base); md["start_tile"] = [2, 0]
    cases.append((md, {}))
    md = copy.deepcopy(base); md["start_tile"] = [1, 0]; md["start_pose"] = [[0.3, 0, 0.25], 1.2]
    cases.append((md, {}))
    jm = junction_map()                                    # has an `optional` object
    jm["tiles"] = [[("grass" if c == "empty" else c) for c in row] for row in jm["tiles"]]   # the reference's reset() cannot
    cases.append((jm, {}))                                  # iterate a grid with empty cells (simulator.py:634-637)
    for md, kw in cases:
        for seed in (3,):
            r, _ = _ref("case", dr, seed, md=copy.deepcopy(md))
            r.user_tile_start = kw.get("user_tile_start")
            o = osim.OracleSim(copy.deepcopy(md), EXT, domain_rand=dr, seed=seed, do_reset=False, **kw)
            for _ in range(3):
                r.reset(); o.reset()
                assert np.array_equal(r.cur_pos, o.cur_pos) and r.cur_angle == o.cur_angle
                assert [bool(ob.visible) for ob in r.objects] == [bool(ob.visible) for ob in o.map.objects]


@pytest.mark.parametrize("m,steps", [("loop_pedestrians", (250, 300)), ("loop_dyn_duckiebots", (60, 240))])
def test_collision_and_proximity_against_moving_objects(m, steps):
    """_collision (static batch + every object's check_collision with its *stored* obj_norm, simulator.py:1473-1492,
    objects.py:152-160, 272-281, 373-382), proximity_penalty2 and _inconvenient_spawn while DuckieObj pedestrians are
    mid-walk / DuckiebotObj followers have driven away from their spawn."""
    md = assets.get_map(m)
    r, ns = _ref(m, md=copy.deepcopy(md))
    if m == "loop_pedestrians":
        for ob in r.objects:
            ob.wiggle = np.pi / 15                       # the reference draws it from the global RNG (objects.py:362)
    o = osim.OracleSim(copy.deepcopy(md), EXT, do_reset=False)
    rng = np.random.default_rng(17)
    t = 0
    for t_end in steps:
        while t < t_end:
            for ob in r.objects:
                if ob.kind == "duckiebot" and not ob.static:
                    ob.step_duckiebot(1 / 30, r.closest_curve_point, r.objects)
                elif not ob.static:
                    ob.step(1 / 30)
            for ob in o.map.objects:
                if ob.kind == "duckiebot" and not ob.static:
                    ob.step_duckiebot(1 / 30, o.closest_curve_point)
                elif not ob.static:
                    ob.step(1 / 30)
            t += 1
        moved = 0
        for ro, oo in zip(r.objects, o.map.objects):
            assert np.array_equal(np.asarray(ro.pos, float), np.asarray(oo.pos, float))
            moved += int(not ro.static)
        assert moved > 0
        n_col = 0
        for _ in range(600):
            ob = r.objects[rng.integers(len(r.objects))]
            c = np.asarray(ob.pos, float)
            pos = np.array([c[0] + rng.uniform(-0.35, 0.35), 0, c[2] + rng.uniform(-0.35, 0.35)])
            a = rng.uniform(-7, 7)
            col = r._collision(ns.simulator.get_agent_corners(pos, a))
            assert col == o._collision(osim.get_agent_corners(pos, a))
            assert r._valid_pose(pos, a) == o._valid_pose(pos, a)
            assert r.proximity_penalty2(pos, a) == o.proximity_penalty2(pos, a)
            assert r._inconvenient_spawn(pos) == o._inconvenient_spawn(pos)
            n_col += int(col)
        assert n_col > 30


def test_dr_constants_match_the_reference():
    """The distribution table tests/test_gpu_device_reset.py checks the device sampler against IS the reference's:
    randomization/config/default_dr.json, randomizer.py DEFAULT_CONFIG, and the camera / wheel constants of simulator.py."""
    import ast, json, os, re
    import test_gpu_device_reset as T
    root = "/root/reference/src/gym_duckietown"
    with open(os.path.join(root, "randomization/config/default_dr.json")) as f:
        assert json.load(f) == T.DR_CONFIG
    src = open(os.path.join(root, "randomization/randomizer.py")).read()
    m = re.search(r"DEFAULT_CONFIG = (\{.*?\n\})", src, re.S)
    assert ast.literal_eval(m.group(1)) == T.DR_CONFIG
    sim = open(os.path.join(root, "simulator.py")).read()
    for name, val in (("CAMERA_FLOOR_DIST", T.CAMERA_FLOOR_DIST), ("CAMERA_ANGLE", T.CAMERA_ANGLE), ("CAMERA_FOV_Y", T.CAMERA_FOV_Y),
                      ("WHEEL_DIST", T.WHEEL_DIST)):
        mm = re.search(rf"^{name} = ([0-9.]+)", sim, re.M)
        assert mm and float(mm.group(1)) == val, name
    # the perturbation scales and base colours used in the device test (simulator.py:551-597)
    for frag in ("self._perturb(self.color_sky)", "self._perturb(WALL_COLOR)", "self._perturb([0.15, 0.15, 0.15], 0.4)",
                 "self._perturb([0.9, 0.9, 0.9], 0.4)", "self._perturb(ambient, 0.3)", "self._perturb(diffuse, 0.99)",
                 "self._perturb(np.array(self.color_ground), 0.3)", "self._perturb(WHEEL_DIST)", "DIM = 0.5",
                 "WALL_COLOR = np.array([0.64, 0.71, 0.28])", "BLUE_SKY = np.array([0.45, 0.82, 1])"):
        assert frag in sim, frag
    obj = open(os.path.join(root, "objects.py")).read()
    for frag in ("self.follow_dist = np.random.uniform(0.3, 0.4)", "self.velocity = np.random.uniform(0.05, 0.15)",
                 "self.gain = gain + np.random.uniform(-0.3, 0.3)", "self.trim = trim + np.random.uniform(-0.1, 0.1) + 2",
                 "self.radius = radius + 0.0002 * np.random.uniform(-1, 1)", "self.wheel_dist = wheel_dist + 0.01 * np.random.uniform(-1, 1)",
                 "self.robot_width = robot_width + 0.01 * np.random.uniform(-1, 1)",
                 "self.robot_length = robot_length + 0.01 * np.random.uniform(-1, 1)"):
        assert frag in obj, frag


def test_led_spheres_are_the_reference_s_draw_calls():
    """enable_leds: WorldObj.render_mesh (objects.py:68-121) run UNMODIFIED with recording GL mocks -- the translate of every LED, the
    glColor4f before each gluSphere and the sphere radii -- against the table oracle/raster.py (and gym_duckietown/simulator.py) state the
    spheres from: positions in the dict's order, DuckiebotObj.leds_color for a follower, blue for any other duckiebot-kind object, a
    1 cm sphere at alpha 1 and a halo of mean(colour) x 4 cm at alpha 0.2."""
    from oracle import raster
    r, ns = _ref("loop_dyn_duckiebots", False, 1)
    gl, glu = ns.objects.gl, ns.objects.gluSphere
    bot = [o for o in r.objects if o.kind == "duckiebot"][0]
    blue = copy.copy([o for o in r.objects if o.kind == "duckie"][0])
    blue.kind = "duckiebot"                              # a static WorldObj of kind duckiebot: not a DuckiebotObj
    for obj, colours in ((bot, raster.LED_FOLLOWER), (blue, raster.LED_STATIC)):
        gl.reset_mock(); glu.reset_mock()
        obj.render_mesh(segment=False, enable_leds=True)
        tr = [tuple(float(v) for v in c.args) for c in gl.glTranslatef.call_args_list]
        assert tr == [tuple(p) for p in raster.LED_POSITIONS]
        cols = [tuple(float(v) for v in c.args) for c in gl.glColor4f.call_args_list]
        radii = [float(c.args[1]) for c in glu.call_args_list]
        assert all(c.args[2:] == (10, 10) for c in glu.call_args_list)
        assert len(cols) == 15 and len(radii) == 10       # per LED: sphere colour, halo colour, reset to white
        for k, col in enumerate(colours):
            assert cols[3 * k] == (*col, 1.0) and cols[3 * k + 1] == (*col, 0.2) and cols[3 * k + 2] == (1.0, 1.0, 1.0, 1.0)
            assert radii[2 * k] == 0.01 and abs(radii[2 * k + 1] - float(np.mean(col)) * 0.04) < 1e-15
        assert gl.glBlendFunc.call_args_list[0].args == (gl.GL_SRC_ALPHA, gl.GL_ONE)
    gl.reset_mock(); glu.reset_mock()
    bot.render_mesh(segment=False, enable_leds=False)
    assert not glu.called                                 # off by default
    duck = [o for o in r.objects if o.kind == "duckie"][0]
    duck.render_mesh(segment=False, enable_leds=True)
    assert not glu.called                                 # duckiebot-kind objects only


def test_curve_overlay_segments_are_the_reference_s_bezier_draw_calls():
    """draw_curve: graphics.bezier_draw (graphics.py:336-349) run UNMODIFIED with recording GL mocks for every curve of every drivable tile,
    in the order and with the red / blue choice of simulator.py:1886-1904, against gym_duckietown.simulator.curve_overlay_segments (what
    the facade hands to dtsim_draw_lines)."""
    from gym_duckietown.simulator import curve_overlay_segments
    r, ns = _ref("loop_only_duckies", False, 3)
    gl = ns.graphics.gl
    for ang in (0.3, 2.0, -1.2):
        want = []
        dir_vec = ns.simulator.get_dir_vec(ang)
        for tile in r.grid:
            if tile is None or not tile["drivable"]:
                continue
            curves = tile["curves"]
            heads = curves[:, -1, :] - curves[:, 0, :]
            heads = heads / np.linalg.norm(heads).reshape(1, -1)                                  # simulator.py:1890-1893
            best = int(np.argmax(np.dot(heads, dir_vec)))
            for idx, red in [(best, True)] + [(i, False) for i in range(len(curves)) if i != best]:
                gl.reset_mock()
                ns.graphics.bezier_draw(curves[idx], n=20, red=red)
                col = tuple(float(v) for v in gl.glColor3f.call_args_list[0].args)
                pts = [tuple(float(v) for v in c.args) for c in gl.glVertex3f.call_args_list]
                assert len(pts) == 20 and gl.glBegin.call_args.args == (gl.GL_LINE_STRIP,)
                want += [[*a, *b, *col] for a, b in zip(pts[:-1], pts[1:])]
        got = np.asarray(curve_overlay_segments(r.grid, ang), dtype=np.float64)
        assert got.shape == (len(want), 9)
        assert np.array_equal(got, np.asarray(want, dtype=np.float64))
