"""GPU parity: HIP raster (dtsim_render through the C-ABI) vs the oracle rasteriser.

Tolerances (camera RGB "within stated float tolerance", BASELINE.json north_star; SURVEY
Appendix B): the HIP raster shades in float32 with per-fragment tile lighting, the oracle in
float64.  Against the oracle in the SAME lighting mode ("pixel"):
    >= 99.9 % of pixels identical within +-1/255, mean abs error <= 0.02/255, and no more
    than 0.05 % of pixels off by more than 2/255 (float32 coverage flips on silhouettes).
Against the oracle's GL-faithful per-vertex ("gouraud") tile lighting (SURVEY's proposal):
    >= 99 % of pixels within +-2/255, mean abs error <= 0.5/255.
"""
import os

import numpy as np
import pytest

from dtsim import BatchedSimulator, _ffi, assets
from dtsim import distortion as pdist
from oracle import raster, sim as osim
from util import EXT, oracle_mode

pytestmark = pytest.mark.gpu
ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "assets")


def _scene(map_name):
    om = osim.OracleMap(assets.get_map(map_name), EXT)
    kinds = {t["kind"] for t in om.grid if t is not None}
    tex = {k: assets.get_texture(k) for k in kinds}
    meshes = {"duckie": assets.get_mesh("duckie"), "*": assets.get_mesh("*")}
    return raster.Scene(om, tex, meshes)


def _camera(sim, e, W, H, dr):
    st = sim.init_states[e]
    pos = sim.read(_ffi.FIELD_POS)[e]
    ang = sim.read(_ffi.FIELD_ANGLE)[e]
    if not dr:
        return raster.Camera(pos, ang, width=W, height=H, horizon_color=list(st.horizon_color),
                             ground_color=list(st.ground_color), light_pos=list(st.light_pos))   # ((0, 3, 0, 1) unless the facade captured it through a model-view)
    return raster.Camera(pos, ang, cam_height=st.cam_height, cam_angle_deg=st.cam_angle_deg,
                         cam_fov_y_deg=st.cam_fov_y_deg, camera_noise=list(st.camera_noise), domain_rand=True,
                         horizon_color=list(st.horizon_color), ground_color=list(st.ground_color),
                         light_pos=list(st.light_pos), light_ambient=list(st.light_ambient),
                         light_diffuse=list(st.light_diffuse), width=W, height=H)


def _stats(a, b):
    d = np.abs(a.astype(np.int32) - b.astype(np.int32)).max(axis=-1)
    return dict(mean=float(np.abs(a.astype(np.int32) - b.astype(np.int32)).mean()), frac_gt1=float((d > 1).mean()),
                frac_gt2=float((d > 2).mean()), max=int(d.max()))


@pytest.mark.parametrize("map_name,W,H,distortion,dr", [
    ("small_loop", 640, 480, True, False),     # BASELINE config C3 geometry
    ("small_loop", 640, 480, False, False),
    ("small_loop", 84, 84, False, False),      # BASELINE config C1 geometry
    ("small_loop", 160, 120, True, False),
    ("small_loop", 640, 480, False, True),     # domain randomisation: per-env camera / light / colours
    ("small_loop", 640, 480, True, True),
])
def test_frames_match_oracle(map_name, W, H, distortion, dr):
    N = 5
    sim = BatchedSimulator(map_name, N, camera_width=W, camera_height=H, distortion=distortion, domain_rand=dr, seed=77)
    acts = np.random.default_rng(3).uniform(0.2, 0.8, (8, N, 2)).astype(np.float32)
    sim.step(acts, n_steps=8)                      # move away from the spawn pose
    sim.render()
    frames = sim.frames_host()
    assert frames.shape == (N, H, W, 3) and frames.dtype == np.uint8
    scene = _scene(map_name)
    rmap = pdist.distortion_maps(W, H) if distortion else None
    for e in range(N):
        cam = _camera(sim, e, W, H, dr)
        ref_px = raster.render_obs(cam, scene, "pixel", rmap)
        s = _stats(frames[e], ref_px)
        assert s["frac_gt1"] <= 1e-3 and s["frac_gt2"] <= 5e-4 and s["mean"] <= 0.02, (e, s)
        ref_g = raster.render_obs(cam, scene, "gouraud", rmap)
        g = _stats(frames[e], ref_g)
        assert g["frac_gt2"] <= 1e-2 and g["mean"] <= 0.5, (e, g)
        # run_tests.py:17-22 property: a sane image
        assert 0 < frames[e].mean() < 255
    sim.close()


def _obj_states(sim, e, scene):
    """Per-object render state of env e (static: map pose; DuckieObj: device centre / y_rot)."""
    cen, yrot = sim.read(_ffi.FIELD_OBJ_CENTER)[e], sim.read(_ffi.FIELD_OBJ_YROT)[e]
    cy = sim.read(_ffi.FIELD_OBJ_Y)[e]
    vis = sim.read(_ffi.FIELD_OBJ_VISIBLE)[e]
    out, slot = [], 0
    for k, o in enumerate(scene.m.objects):
        if o.static:
            out.append(dict(pos=o.pos, y_rot=o.y_rot, visible=bool(vis[k])))
        else:
            out.append(dict(pos=np.array([cen[slot, 0], cy[slot], cen[slot, 1]]), y_rot=float(yrot[slot]), visible=bool(vis[k])))
            slot += 1
    return out


@pytest.mark.parametrize("map_name,W,H,distortion,dr,steps", [
    ("small_loop_only_duckies", 640, 480, False, False, 4),
    ("loop_only_duckies", 320, 240, True, False, 4),
    ("loop_only_duckies", 640, 480, False, True, 4),
    ("loop_pedestrians", 320, 240, False, False, 255),    # duckies mid-walk (wiggling y_rot)
    ("loop_dyn_duckiebots", 320, 240, False, False, 120),  # follower bots (DuckiebotObj) + static duckies
])
def test_frames_with_mesh_objects_match_oracle(map_name, W, H, distortion, dr, steps):
    """Static duckies (WorldObj) and walking pedestrians (DuckieObj): z-buffered mesh triangles
    with per-vertex lighting and 4x MSAA coverage.  Same tolerances as the plane-only case, with the
    silhouette allowance scaled to the object edge length."""
    N = 6
    sim = BatchedSimulator(map_name, N, camera_width=W, camera_height=H, distortion=distortion, domain_rand=dr,
                           seed=123, max_steps=100000)
    zero = np.zeros((steps, N, 2), np.float32)
    sim.step(zero, n_steps=steps)
    sim.render()
    frames = sim.frames_host()
    scene = _scene(map_name)
    rmap = pdist.distortion_maps(W, H) if distortion else None
    n_obj_px = 0
    for e in range(N):
        cam = _camera(sim, e, W, H, dr)
        st = _obj_states(sim, e, scene)
        ref_px = raster.render_obs(cam, scene, "pixel", rmap, obj_states=st)
        no_obj = raster.render_obs(cam, scene, "pixel", rmap, obj_states=[dict(s_, visible=False) for s_ in st])
        n_obj_px += int((np.abs(ref_px.astype(int) - no_obj.astype(int)).max(-1) > 0).sum())
        s = _stats(frames[e], ref_px)
        assert s["frac_gt1"] <= 2e-3 and s["frac_gt2"] <= 1e-3 and s["mean"] <= 0.03, (e, s)
    assert n_obj_px > 200, n_obj_px          # the duckies are actually in view in this sample
    sim.close()


def _asset_scene():
    """Oracle scene of the real-asset fixture (tests/golden/assets): per-kind meshes with textures."""
    lib = assets.AssetLibrary(ASSETS)
    md = lib.map_data("test_town")
    meshes = {"*": assets.get_mesh("*")}
    for desc in md["objects"]:
        meshes[desc["kind"]] = lib.object_mesh(desc)[1]
    ext = {k: (m.min_coords, m.max_coords) for k, m in meshes.items()}
    om = osim.OracleMap(md, ext)
    kinds = {t["kind"] for t in om.grid if t is not None}
    scene = raster.Scene(om, {k: lib.tile_texture(k) for k in kinds}, meshes)
    scene.light_cards = lib.light_cards()              # TrafficLightObj.texs (objects.py:438-441)
    return scene, md, ext


@pytest.mark.parametrize("W,H,distortion,dr,steps", [(320, 240, False, False, 0), (640, 480, True, False, 0), (320, 240, False, True, 0),
                                                     (320, 240, False, False, 151)])   # 151 steps: the traffic light has switched
def test_real_assets_match_oracle(W, H, distortion, dr, steps):
    """MapFormat1 YAML + tile texture files + OBJ/MTL meshes (multi-material, textured
    chunks: GL_MODULATE of the material texture with the lit vertex colour, sign / duckiebot material
    overrides) loaded from an asset tree, rendered by the HIP raster and by the oracle."""
    scene, md, ext = _asset_scene()
    N = 8
    sim = BatchedSimulator("test_town", N, asset_root=ASSETS, camera_width=W, camera_height=H, distortion=distortion,
                           domain_rand=dr, seed=5, max_steps=100000)
    assert [o.mesh_kind for o in sim.maps[0].objects] == ["cone", "sign_stop", "tree", "duckiebot:blue", "duckie", "cone", "trafficlight"]
    # look at the objects: place the agents around them, facing them
    objs = sim.maps[0].objects
    for e in range(N):
        o = objs[e % len(objs)]
        a = 0.7 * e
        sim.init_states[e].pos[:] = [float(o.pos[0] - 0.45 * np.cos(a)), 0.0, float(o.pos[2] + 0.45 * np.sin(a))]
        sim.init_states[e].angle = float(a)
    sim.reset(states=sim.init_states)
    if steps:
        sim.step(np.zeros((steps, N, 2), np.float32), n_steps=steps)       # the agents stay put, the object clocks run
    light = sim.read(_ffi.FIELD_OBJ_LIGHT)
    assert (light[:, 6] == (1 if steps >= 150 else 0)).all() and (light[:, :6] == 0).all()
    sim.render()
    frames = sim.frames_host()
    rmap = pdist.distortion_maps(W, H) if distortion else None
    n_obj_px = 0
    for e in range(N):
        cam = _camera(sim, e, W, H, dr)
        st = _obj_states(sim, e, scene)
        for k_, s_ in enumerate(st):
            s_["light_pattern"] = int(light[e, k_])
        mode = oracle_mode(sim)                    # 128 x 128 tile images: the per-env path is the generic raster (llvmpipe's filter)
        ref_px = raster.render_obs(cam, scene, mode, rmap, obj_states=st)
        no_obj = raster.render_obs(cam, scene, mode, rmap, obj_states=[dict(s_, visible=False) for s_ in st])
        n_obj_px += int((np.abs(ref_px.astype(int) - no_obj.astype(int)).max(-1) > 0).sum())
        s = _stats(frames[e], ref_px)
        assert s["frac_gt1"] <= 2e-3 and s["frac_gt2"] <= 1e-3 and s["mean"] <= 0.03, (e, s)
    assert n_obj_px > 2000, n_obj_px
    sim.close()


@pytest.mark.parametrize("distortion", [False, True])
def test_file_textures_without_objects_match_oracle(distortion):
    """The asset tree's 128 x 128 tile images on a map without mesh objects: the quad-layout raster with the generic
    texture size (not the S = 256 specialisation) and the exact path inside the raster wavefronts."""
    import copy
    lib = assets.AssetLibrary(ASSETS)
    md = copy.deepcopy(lib.map_data("test_town"))
    md["objects"] = []
    W, H, N = 640, 480, 6
    sim = BatchedSimulator("test_town_bare", N, map_data=md, asset_root=ASSETS, camera_width=W, camera_height=H,
                           distortion=distortion, domain_rand=False, seed=11, max_steps=100000)
    sim.step(np.random.default_rng(5).uniform(0.2, 0.8, (6, N, 2)).astype(np.float32), n_steps=6)
    sim.render()
    frames = sim.frames_host()
    om = osim.OracleMap(md, {"*": (assets.get_mesh("*").min_coords, assets.get_mesh("*").max_coords)})
    kinds = {t["kind"] for t in om.grid if t is not None}
    assert {lib.tile_texture(k).shape[0] for k in kinds} == {128}
    scene = raster.Scene(om, {k: lib.tile_texture(k) for k in kinds}, {"*": assets.get_mesh("*")})
    rmap = pdist.distortion_maps(W, H) if distortion else None
    for e in range(N):
        cam = _camera(sim, e, W, H, False)
        s = _stats(frames[e], raster.render_obs(cam, scene, "pixel", rmap))
        assert s["frac_gt1"] <= 1e-3 and s["frac_gt2"] <= 5e-4 and s["mean"] <= 0.02, (e, s)
    sim.close()


def test_traffic_light_pattern_follows_the_reference_clock():
    """TrafficLightObj.step on the device vs the pattern sequence recorded from the reference's own code
    (tests/golden/ref_trafficlight.npz), including frame_skip sub-steps; the clock survives env resets."""
    g = np.load(os.path.join(os.path.dirname(ASSETS), "ref_trafficlight.npz"))
    for tag, kw in (("30hz", dict()), ("20hz", dict(frame_rate=20)), ("frame_skip_dt", dict(frame_skip=3))):
        sim = BatchedSimulator("test_town", 3, asset_root=ASSETS, render=False, domain_rand=False, seed=1, max_steps=10**6, **kw)
        ref = g[tag]
        per = 3 if tag == "frame_skip_dt" else 1
        pats = []
        T = 420
        for t in range(T):
            sim.step(np.zeros((1, 3, 2), np.float32))
            pats.append(sim.read(_ffi.FIELD_OBJ_LIGHT)[0, 6])
            if t == 200:
                sim.reset()                            # objects persist across resets (simulator.py:349,865)
        pats = np.array(pats)
        if tag == "frame_skip_dt":
            want = np.array([g["30hz"][per * (t + 1) - 1] for t in range(T)])     # 3 object steps of 1/30 s per env step
        else:
            want = ref[:T]
        assert np.array_equal(pats, want), tag
        sim.close()


def test_render_is_deterministic_and_per_env():
    N = 72   # > ENVS_PER_BLOCK (64): several env chunks
    sim = BatchedSimulator("small_loop", N, camera_width=160, camera_height=120, domain_rand=False, seed=5)
    sim.render()
    a = sim.frames_host().copy()
    sim.render()
    assert np.array_equal(a, sim.frames_host())
    assert len({a[e].tobytes() for e in range(N)}) == N        # every env has its own pose => frame
    sim.close()


def test_odd_frame_sizes_match_oracle():
    """Widths that are not a multiple of 4 take the byte-store path; partial 64x16 tiles on both axes."""
    for (W, H, N) in ((90, 62, 3), (66, 17, 2)):
        sim = BatchedSimulator("small_loop_only_duckies", N, camera_width=W, camera_height=H, distortion=False,
                               domain_rand=False, seed=4)
        sim.step(np.full((3, N, 2), 0.5, np.float32), n_steps=3)
        sim.render()
        frames = sim.frames_host()
        assert frames.shape == (N, H, W, 3)
        scene = _scene("small_loop_only_duckies")
        for e in range(N):
            cam = _camera(sim, e, W, H, False)
            ref_px = raster.render_obs(cam, scene, oracle_mode(sim), None, obj_states=_obj_states(sim, e, scene))
            s = _stats(frames[e], ref_px)
            assert s["frac_gt1"] <= 4e-3 and s["mean"] <= 0.05, (W, H, e, s)     # tiny frames: every silhouette pixel counts
        sim.close()


def test_checkerboard_renders_at_its_moving_centre():
    """CheckerboardObj: rendered at pos = centre, including the vertical part of its script (objects.py:531-587)."""
    md = assets.get_map("small_loop")
    md["objects"] = [dict(kind="checkerboard", pos=[2.5, 1.3], rotate=30, height=0.25, static=False)]
    N, W, H = 3, 320, 240
    sim = BatchedSimulator("cb", N, map_data=md, camera_width=W, camera_height=H, distortion=False, domain_rand=False,
                           seed=8, max_steps=10**6, do_reset=False)
    om = osim.OracleMap(md, EXT)
    scene = raster.Scene(om, {k: assets.get_texture(k) for k in {t["kind"] for t in om.grid if t is not None}},
                         {"duckie": assets.get_mesh("duckie"), "*": assets.get_mesh("*")})
    st = (_ffi.InitState * N)()
    o = om.objects[0]
    for e in range(N):
        a = 0.4 + 0.9 * e
        st[e].pos[:] = [float(o.pos[0] - 0.5 * np.cos(a)), 0.0, float(o.pos[2] + 0.5 * np.sin(a))]
        st[e].angle = float(a); st[e].wheel_dist = 0.102
        st[e].cam_height, st[e].cam_angle_deg, st[e].cam_fov_y_deg = 0.108, 19.15, 75.0
        st[e].horizon_color[:] = [0.45, 0.82, 1.0]; st[e].ground_color[:] = [0.15, 0.15, 0.15]
        st[e].light_pos[:] = [0.0, 3.0, 0.0, 1.0]; st[e].light_ambient[:] = [0.25] * 3; st[e].light_diffuse[:] = [0.35] * 3
    sim.init_states = st
    sim.reset(states=st)
    sim.step(np.zeros((170, N, 2), np.float32), n_steps=170)        # step counter 320: on its way up
    assert 0.02 < sim.read(_ffi.FIELD_OBJ_Y)[0, 0] < 0.2
    sim.render()
    frames = sim.frames_host()
    for e in range(N):
        cam = _camera(sim, e, W, H, False)
        stt = _obj_states(sim, e, scene)
        ref_px = raster.render_obs(cam, scene, "pixel", None, obj_states=stt)
        s_ = _stats(frames[e], ref_px)
        assert s_["frac_gt1"] <= 2e-3 and s_["mean"] <= 0.03, (e, s_)
    sim.close()


@pytest.mark.parametrize("map_name,W,H,distortion,dr", [
    ("loop_only_duckies", 320, 240, False, False),
    ("loop_dyn_duckiebots", 320, 240, True, True),
])
def test_segment_render_matches_oracle(map_name, W, H, distortion, dr):
    """`render_obs(segment=True)` (simulator.py:1730-1737,1753,1808,1879): unlit, magenta sky / ground, segmented
    tile textures, flat-coloured meshes; and the normal render afterwards is unchanged."""
    N = 3
    sim = BatchedSimulator(map_name, N, camera_width=W, camera_height=H, distortion=distortion, domain_rand=dr, seed=5)
    sim.step(np.random.default_rng(4).uniform(0.2, 0.8, (6, N, 2)).astype(np.float32), n_steps=6)
    sim.render()
    normal = sim.frames_host().copy()
    sim.render(segment=True)
    frames = sim.frames_host().copy()
    sim.render()
    assert np.array_equal(sim.frames_host(), normal)
    scene = _scene(map_name)
    seg_tex, rgb = sim.segment_assets()
    seg_by_kind = {kd: seg_tex[i] for i, kd in enumerate(sim.texture_kinds)}
    cols = {mk: rgb[i] for i, mk in enumerate(sim._mesh_order)}
    rmap = pdist.distortion_maps(W, H) if distortion else None
    n_obj_px = 0
    for e in range(N):
        cam, sc = raster.segment_view(_camera(sim, e, W, H, dr), scene, seg_by_kind, cols)
        st = _obj_states(sim, e, scene)
        ref_px = raster.render_obs(cam, sc, "pixel", rmap, obj_states=st)
        s_ = _stats(frames[e], ref_px)
        assert s_["frac_gt1"] <= 2e-3 and s_["frac_gt2"] <= 1e-3 and s_["mean"] <= 0.03, (e, s_)
        valid = np.ones((H, W), bool) if rmap is None else (ref_px.sum(-1) > 0) | (frames[e].sum(-1) > 0)
        magenta = (frames[e] == np.array([255, 0, 255], np.uint8)).all(-1)
        assert magenta.mean() > 0.2                      # the sky, at least
        obj = cols["duckie"].astype(int)
        n_obj_px += int((np.abs(frames[e].astype(int) - obj).max(-1) <= 60).sum())
    sim.close()


def test_undistort_wrapper_folded_into_the_raster():
    """UndistortWrapper (wrappers.py:145-227) = render without the fisheye, then cv2.remap(INTER_NEAREST) through
    initUndistortRectifyMap(K, D, I, P).  With `undistort=True` the wrapper's map is the raster's source map: the
    device frame must equal the host wrapper applied to the device's rectilinear frame, byte for byte."""
    from gym_duckietown.wrappers import UndistortWrapper
    N, W, H = 3, 640, 480
    kw = dict(camera_width=W, camera_height=H, domain_rand=False, seed=21)
    plain = BatchedSimulator("loop_only_duckies", N, distortion=False, **kw)
    und = BatchedSimulator("loop_only_duckies", N, distortion=True, undistort=True, **kw)
    acts = np.random.default_rng(2).uniform(0.2, 0.8, (5, N, 2)).astype(np.float32)
    for s_ in (plain, und):
        s_.step(acts, n_steps=5)
        s_.render()
    assert np.array_equal(plain.read(_ffi.FIELD_POS), und.read(_ffi.FIELD_POS))
    rect, got = plain.frames_host(), und.frames_host()
    w = UndistortWrapper.__new__(UndistortWrapper)
    w.mapx = w.mapy = None
    mx, my = pdist.undistort_wrapper_maps(W, H)
    assert np.array_equal(np.stack(w._maps(H, W)), np.stack([mx, my]))          # the two host statements agree
    for e in range(N):
        assert np.array_equal(w.observation(rect[e]), got[e]), e
    assert (got.reshape(N, -1, 3).sum(-1) == 0).mean() > 0.01                      # BORDER_CONSTANT corners
    plain.close(); und.close()


def test_domain_rand_with_a_positional_light_takes_the_exact_paths():
    """Domain randomisation draws a DIRECTIONAL light (simulator.py:565-584: light_pos has w = 0), which is what lets k_raster_v3dr fold the
    tile plane's light into a per-env constant.  A caller can still write a positional light (DTSIM_FIELD_COLORS, w = 1): those envs get an
    empty one-ray range, every tile pixel goes to the exact path, and k_resolve_dr evaluates the light per pixel from the EnvCam.  Frames
    against the oracle with the same light, plane-only thresholds; envs 0 / 2 positional, 1 / 3 as drawn."""
    N, W, H = 4, 320, 240
    for map_name, tol in (("small_loop", (1e-3, 5e-4, 0.02)), ("loop_only_duckies", (2e-3, 1e-3, 0.03))):
        sim = BatchedSimulator(map_name, N, camera_width=W, camera_height=H, distortion=True, domain_rand=True, seed=13)
        sim.step(np.random.default_rng(5).uniform(0.2, 0.8, (6, N, 2)).astype(np.float32), n_steps=6)
        col = sim.read(_ffi.FIELD_COLORS).copy()           # [N][16]: horizon, ground, ambient, diffuse, light xyzw
        lights = {0: (0.3, 2.5, -0.4, 1.0), 2: (-1.0, 1.5, 0.8, 1.0)}
        for e, L in lights.items():
            col[e, 12:16] = L
        sim.write(_ffi.FIELD_COLORS, col)
        sim.render()
        frames = sim.frames_host()
        scene = _scene(map_name)
        rmap = pdist.distortion_maps(W, H)
        for e in range(N):
            cam = _camera(sim, e, W, H, True)
            if e in lights:
                cam.L = np.asarray(lights[e], dtype=np.float64)
            ref = raster.render_obs(cam, scene, "pixel", rmap, obj_states=_obj_states(sim, e, scene) if scene.m.objects else None)
            s = _stats(frames[e], ref)
            assert s["frac_gt1"] <= tol[0] and s["frac_gt2"] <= tol[1] and s["mean"] <= tol[2], (map_name, e, s)
        sim.close()


def test_domain_rand_over_two_maps_in_the_render_order():
    """k_raster_v3dr / k_resolve_dr with more than one map resident (MultiMap slot alternation) and more than one chunk, so that the pass runs in
    k_env_sort's order: per-position records (EnvD, object masks, queue regions) against per-env ones (EnvCam, frames, triangles), the maps'
    column offsets in the LDS tile table.  Six envs -- both maps, both chunks, the partial tail chunk -- against the oracle."""
    N, W, H = 72, 320, 240                                 # (> 64 envs: two chunks of the render order, the second one partial)
    names = ["loop_only_duckies", "small_loop_only_duckies"]
    sim = BatchedSimulator(names, N, camera_width=W, camera_height=H, distortion=True, domain_rand=True, seed=23, map_cycle=True, max_steps=100000)
    sim.reset(mask=(np.arange(N) % 2 == 0))                # multimap_env.py:44-49: the two maps alternate over the slots
    sim.step(np.random.default_rng(6).uniform(0.2, 0.7, (5, N, 2)).astype(np.float32), n_steps=5)
    sim.render()
    frames = sim.frames_host()
    mid = sim.read(_ffi.FIELD_MAP_ID)
    assert set(np.unique(mid)) == {0, 1}
    rpos = sim.read(_ffi.FIELD_RENDER_POS)
    assert sorted(rpos.tolist()) == list(range(N)) and not np.array_equal(rpos, np.arange(N))     # a real permutation: the sorted order is in use
    scenes = [_scene(n) for n in names]
    rmap = pdist.distortion_maps(W, H)
    env_at = np.argsort(rpos)
    for e in sorted({0, 1, int(env_at[0]), int(env_at[63]), int(env_at[64]), int(env_at[N - 1])}):
        scene = scenes[int(mid[e])]
        ref = raster.render_obs(_camera(sim, e, W, H, True), scene, "pixel", rmap, obj_states=_obj_states(sim, e, scene))
        s = _stats(frames[e], ref)
        assert s["frac_gt1"] <= 2e-3 and s["frac_gt2"] <= 1e-3 and s["mean"] <= 0.03, (e, int(mid[e]), s)
    sim.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["shared", "dr"])
def test_a_state_renders_to_the_same_bytes_every_time(cfg):
    """One batch rendered four times from one state: byte-identical frames.  The envs reach the raster in a sorted order that is not the same
    from launch to launch, so an env is shaded by the peeled first iteration of the env loop one time and by the loop body the next, and a
    queued pixel by the exact path: all three must round alike.  (Round 6: with the coordinates snapped to 256ths of a texel the last bit of
    the float sums became visible -- -ffp-contract=fast had fused different products in the three copies: +-1/255 on ~ 10 of 157 M pixels.)"""
    N, W, H = 256, 640, 480
    sim = BatchedSimulator("small_loop", N, camera_width=W, camera_height=H, distortion=True, domain_rand=(cfg == "dr"), seed=5, max_steps=100000)
    acts = np.random.default_rng(9).uniform(0.2, 0.9, (4, N, 2)).astype(np.float32)
    sim.step(acts, n_steps=4)
    fr = []
    for _ in range(4):
        sim.render()
        fr.append(sim.frames_host().copy())
    sim.close()
    assert fr[0].std() > 10.0
    for i in range(1, 4):
        assert np.array_equal(fr[0], fr[i]), (cfg, i, int((fr[0] != fr[i]).any(axis=-1).sum()))


@pytest.mark.gpu
def test_k_raster_q_and_k_raster_v3_agree(monkeypatch):
    """k_raster_q -- the quad-record raster of the other square texture sizes, and of 256 x 256 textures on grids beyond 32 x 24 tiles -- against
    k_raster_v3 on the SAME batch (DTSIM_RASTER_OLD=1 at dtsim_create keeps k_raster_q for 256 x 256 textures): same records (4 x 2 cells to a
    line), same snapped coordinates, same byte-weight filter; the exact paths differ in form (k_raster_q's gives interior entries the one-ray
    colour, k_raster_v3's takes the four samples of every entry), which may move a rounding tie on a queued pixel."""
    N, W, H = 64, 640, 480
    out = []
    for old in ("0", "1"):
        monkeypatch.setenv("DTSIM_RASTER_OLD", old)
        sim = BatchedSimulator("small_loop", N, camera_width=W, camera_height=H, distortion=True, domain_rand=False, seed=11, max_steps=100000)
        acts = np.random.default_rng(3).uniform(0.2, 0.9, (4, N, 2)).astype(np.float32)
        sim.step(acts, n_steps=4)
        sim.render()
        out.append(sim.frames_host().copy())
        sim.close()
    d = np.abs(out[0].astype(np.int32) - out[1].astype(np.int32)).max(axis=-1)
    assert out[0].std() > 10.0
    assert (d > 0).mean() <= 2e-3 and (d > 1).mean() <= 1e-5, ((d > 0).mean(), (d > 1).mean(), int(d.max()))
