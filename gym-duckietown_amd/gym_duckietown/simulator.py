"""Drop-in `Simulator` (reference: src/gym_duckietown/simulator.py:188-2053) over the HIP library.

Same constructor keywords, `reset() / step() / render() / seed() / close()`, attributes
(`cur_pos`, `cur_angle`, `speed`, `step_count`, `timestamp`, `wheel_dist`, `grid`,
`drivable_tiles`, ...) and query methods (`closest_curve_point`, `get_lane_pos2`,
`_valid_pose`, `_collision`, `proximity_penalty2`, `compute_reward`, `_compute_done_reward`,
`get_grid_coords`, `_get_tile`, `_drivable_pos`, `get_agent_info`) plus the module-level
`_update_pos`, `get_agent_corners`, `get_dir_vec`, `get_right_vec`, `_actual_center`.
It is an N=1 view of dtsim.BatchedSimulator: every number comes from the GPU kernels
(`dtsim_step`, `dtsim_render`, `dtsim_query`); nothing is recomputed on the host.

`render(mode)` returns the 800x600 window image of every mode (agent camera, `free_cam`, `top_down`) from a second
handle that copies the state; `segment=True` is the segmentation render.

`draw_curve` / `draw_bbox` (round 5): the GL_LINE overlays as a post-pass on the resolved frame (`dtsim_draw_lines`; draw_bbox also switches
to the reference's debugging camera 0.8 m above the robot).  Not provided (out of scope): the pyglet window itself (nothing
is displayed), `camera_rand`'s carnivalmirror calibration sampling.  `enable_leds` blends the duckiebots' LED spheres
into the frame as a post-pass (`dtsim_draw_leds`: analytic spheres, front surfaces, after all opaque objects).
"""
from __future__ import annotations

import math
from collections import namedtuple
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import numpy as np

from dtsim import BatchedSimulator, _ffi
from dtsim.reset import BLUE_SKY

from . import logger
from .exceptions import InvalidMapException, NotInLane

try:  # gym is optional
    import gym
    from gym import spaces
    _EnvBase = gym.Env
except Exception:  # pragma: no cover
    gym = None
    _EnvBase = object

    class _Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low, self.high, self.dtype = low, high, dtype
            self.shape = tuple(shape) if shape is not None else np.asarray(low).shape

        def sample(self):
            return np.random.uniform(self.low, self.high, self.shape).astype(self.dtype)

    class spaces:  # noqa: N801
        Box = _Box

# constants of the reference (simulator.py:99-177)
WINDOW_WIDTH, WINDOW_HEIGHT = 800, 600
DEFAULT_CAMERA_WIDTH, DEFAULT_CAMERA_HEIGHT = 640, 480
CAMERA_ANGLE, CAMERA_FOV_Y, CAMERA_FLOOR_DIST, CAMERA_FORWARD_DIST = 19.15, 75, 0.108, 0.066
WHEEL_DIST = 0.102
ROBOT_WIDTH = 0.13 + 0.02
ROBOT_LENGTH = 0.18
ROBOT_HEIGHT = 0.12
SAFETY_RAD_MULT = 1.8
AGENT_SAFETY_RAD = (max(ROBOT_LENGTH, ROBOT_WIDTH) / 2) * SAFETY_RAD_MULT
MIN_SPAWN_OBJ_DIST = 0.25
DEFAULT_ROBOT_SPEED = 1.20
DEFAULT_FRAMERATE = 30
DEFAULT_MAX_STEPS = 1500
DEFAULT_MAP_NAME = "udem1"
DEFAULT_FRAME_SKIP = 1
DEFAULT_ACCEPT_START_ANGLE_DEG = 60
REWARD_INVALID_POSE = -1000
MAX_SPAWN_ATTEMPTS = 5000

LanePosition0 = namedtuple("LanePosition", "dist dot_dir angle_deg angle_rad")


class LanePosition(LanePosition0):
    def as_json_dict(self):
        return dict(dist=self.dist, dot_dir=self.dot_dir, angle_deg=self.angle_deg, angle_rad=self.angle_rad)


@dataclass
class DoneRewardInfo:
    done: bool
    done_why: str
    done_code: str
    reward: float


_DONE_MSG = {
    _ffi.DONE_IN_PROGRESS: "",
    _ffi.DONE_INVALID_POSE: "Stopping the simulator because we are at an invalid pose.",
}


class Simulator(_EnvBase):
    metadata = {"render.modes": ["human", "rgb_array", "app"], "video.frames_per_second": 30}
    _ACTION_MODE = "wheels"

    def __init__(self, map_name: str = DEFAULT_MAP_NAME, max_steps: int = DEFAULT_MAX_STEPS, draw_curve: bool = False,
                 draw_bbox: bool = False, domain_rand: bool = True, frame_rate: float = DEFAULT_FRAMERATE,
                 frame_skip: int = DEFAULT_FRAME_SKIP, camera_width: int = DEFAULT_CAMERA_WIDTH,
                 camera_height: int = DEFAULT_CAMERA_HEIGHT, robot_speed: float = DEFAULT_ROBOT_SPEED,
                 accept_start_angle_deg=DEFAULT_ACCEPT_START_ANGLE_DEG, full_transparency: bool = False,
                 user_tile_start=None, seed: int = None, distortion: bool = False, dynamics_rand: bool = False,
                 camera_rand: bool = False, randomize_maps_on_reset: bool = False, num_tris_distractors: int = 12,
                 color_ground: Sequence[float] = (0.15, 0.15, 0.15), color_sky: Sequence[float] = BLUE_SKY,
                 style: str = "photos", enable_leds: bool = False, device: int = 0, **env_kwargs):
        if camera_rand:
            raise NotImplementedError("camera_rand (carnivalmirror calibration sampling) is outside the path this backend implements")
        self.enable_leds = bool(enable_leds)
        self.gl_filter = bool(env_kwargs.pop("gl_filter", False))          # render with the reference renderer's GL_LINEAR arithmetic (DTSIM_RENDER_GL_FILTER: bit-faithful frames, slower)
        self.gl_light_capture = bool(env_kwargs.pop("gl_light_capture", True))   # reset()'s light through the last frame's model-view, as GL does (False: as given)
        if not domain_rand and (env_kwargs.get("device_reset") or env_kwargs.get("auto_reset")):
            self.gl_light_capture = False                # device-side resets do not come back through reset(): nothing to capture (and the per-env path would apply the sampler's camera noise)
        self._gl_modelview = None        # what the last frame left (None: the identity -- nothing drawn yet)
        self.seed_value = seed
        self.num_tris_distractors = num_tris_distractors
        self.color_ground, self.color_sky = color_ground, list(color_sky)
        self.full_transparency = full_transparency
        self.max_steps, self.draw_curve, self.draw_bbox, self.domain_rand = max_steps, draw_curve, draw_bbox, domain_rand
        self.frame_rate, self.delta_time, self.frame_skip = frame_rate, 1.0 / frame_rate, frame_skip
        self.graphics = True
        self.action_space = spaces.Box(low=-1, high=1, shape=(2,), dtype=np.float32)
        self.camera_width, self.camera_height = camera_width, camera_height
        self.robot_speed = robot_speed
        self.observation_space = spaces.Box(low=0, high=255, shape=(camera_height, camera_width, 3), dtype=np.uint8)
        self.reward_range = (-1000, 1000)
        self.window = None
        self.accept_start_angle_deg = accept_start_angle_deg
        self.distortion = distortion and not draw_bbox
        self.camera_rand = False
        self._undistort = False
        self.dynamics_rand = dynamics_rand
        self.user_tile_start = user_tile_start
        self.style = style
        self.randomize_maps_on_reset = bool(randomize_maps_on_reset)
        maps_arg = map_name
        if self.randomize_maps_on_reset:                 # simulator.py:373-378: every map but calibration* / regress*
            # (a list / tuple as map_name restricts the draw to those maps: the way out when the asset tree holds more map
            # files than the library keeps resident)
            self.map_names = list(map_name) if isinstance(map_name, (list, tuple)) else self._all_map_names(env_kwargs.get("asset_root"))
            maps_arg = list(self.map_names)
            env_kwargs = dict(env_kwargs, map_random=True)
        try:
            self._sim = BatchedSimulator(
                maps_arg, 1, max_steps=max_steps, domain_rand=domain_rand, frame_rate=frame_rate, frame_skip=frame_skip,
                camera_width=camera_width, camera_height=camera_height, robot_speed=robot_speed,
                accept_start_angle_deg=accept_start_angle_deg, user_tile_start=user_tile_start, seed=seed,
                distortion=self.distortion, dynamics_rand=dynamics_rand, num_tris_distractors=num_tris_distractors,
                color_ground=color_ground, color_sky=color_sky, action_mode=self._ACTION_MODE, actions_f64=True,
                device=device, style=style, do_reset=False, per_env_camera=self.gl_light_capture, **env_kwargs)
        except KeyError as e:
            raise InvalidMapException("Cannot load map data", map_name=map_name) from e
        except Exception as e:
            # every resident map's tables are staged into LDS by k_step (60 KB budget, dtsim_set_maps: DTSIM_E_LIMIT), and more
            # than 4 maps leave the quad-record rasters for the generic ones (slower, same frames): say so where it happens
            from dtsim import _ffi
            if self.randomize_maps_on_reset and getattr(e, "code", None) == _ffi.E_LIMIT:
                raise ValueError(f"randomize_maps_on_reset: the {len(maps_arg)} maps do not fit the library's resident-map budget ({e}); "
                                 "pass map_name=[...] with the maps to draw from") from e
            raise
        self._bind_map(0)
        self.cam_offset = np.array([0, 0, 0])
        self.reset()
        self.last_action = np.array([0, 0])
        self.wheelVels = np.array([0, 0])

    @staticmethod
    def _all_map_names(asset_root=None):
        """The map list randomize_maps_on_reset draws from (simulator.py:373-378), sorted, and capped at the
        library's DTSIM_MAX_MAPS resident maps."""
        import os
        from dtsim import assets
        lib = assets.AssetLibrary(asset_root)
        if lib.root:
            names = sorted({os.path.splitext(os.path.basename(p))[0] for p in lib._files
                            if p.endswith(".yaml") and os.path.basename(os.path.dirname(p)) == "maps"})
        else:
            names = sorted(assets.MAPS)
        names = [n for n in names if not n.startswith(("calibration", "regress"))]
        if len(names) > _ffi.MAX_MAPS:
            # the reference draws from EVERY map file (simulator.py:373-378, :547-549): a silent cut would change what reset() can
            # return.  32 resident maps cover the reference's own tree (~20); beyond that the caller has to choose.
            raise ValueError(f"randomize_maps_on_reset: {len(names)} map files under the asset root, the library keeps at most "
                             f"{_ffi.MAX_MAPS} resident (DTSIM_MAX_MAPS); pass map_name=[...] with the maps to draw from")
        return names

    def _bind_map(self, idx: int):
        """Point the map-dependent attributes (simulator.py:_load_map / _interpret_map) at map `idx` of the handle."""
        mt = self._sim.maps[idx]
        self._map_idx = idx
        self._mt = mt
        self.map_name = mt.name
        self.road_tile_size = mt.tile_size
        self.grid_width, self.grid_height = mt.grid_w, mt.grid_h
        self.grid = []
        for ti, kind in enumerate(mt.tile_kind_names):
            if kind is None:
                self.grid.append(None)
                continue
            t = {"coords": (ti % mt.grid_w, ti // mt.grid_w), "kind": kind, "angle": int(mt.tile_angle[ti]),
                 "drivable": bool(mt.tile_curve_off[ti] >= 0)}
            if t["drivable"]:
                o, c = int(mt.tile_curve_off[ti]), int(mt.tile_curve_cnt[ti])
                t["curves"] = mt.curves3[o:o + c]
            self.grid.append(t)
        self.drivable_tiles = [self.grid[j * mt.grid_w + i] for (i, j) in mt.drivable_tiles]
        self.objects = mt.objects
        self.start_tile = self._get_tile(*mt.start_tile) if mt.start_tile is not None else None
        self.start_pose = mt.start_pose

    # ---------------------------------------------------------------- state views --
    # fields of dtsim_agent_info: one transfer per state version instead of one dtsim_read per value
    _SNAP = {
        _ffi.FIELD_POS: lambda a: np.array(a.pos, np.float64), _ffi.FIELD_ANGLE: lambda a: np.float64(a.angle),
        _ffi.FIELD_SPEED: lambda a: np.float64(a.speed), _ffi.FIELD_TIMESTAMP: lambda a: np.float64(a.timestamp),
        _ffi.FIELD_WHEELS: lambda a: np.array(a.wheels, np.float64), _ffi.FIELD_LANE: lambda a: np.array(a.lane, np.float64),
        _ffi.FIELD_PROX: lambda a: np.float64(a.prox), _ffi.FIELD_REWARD: lambda a: np.float64(a.reward),
        _ffi.FIELD_TILE: lambda a: np.array(a.tile, np.int32), _ffi.FIELD_STEP_COUNT: lambda a: np.int32(a.step_count),
        _ffi.FIELD_IN_LANE: lambda a: np.uint8(a.in_lane), _ffi.FIELD_DONE: lambda a: np.uint8(a.done),
        _ffi.FIELD_DONE_CODE: lambda a: np.uint8(a.done_code),
    }

    # fields the reference derives from cur_pos / cur_angle ON DEMAND (get_agent_info :1586, _compute_done_reward :1685)
    _DERIVED = (_ffi.FIELD_LANE, _ffi.FIELD_PROX, _ffi.FIELD_REWARD, _ffi.FIELD_TILE, _ffi.FIELD_IN_LANE, _ffi.FIELD_DONE,
                _ffi.FIELD_DONE_CODE)

    def _f(self, field):
        get = self._SNAP.get(field)
        if get is None:
            return self._sim.read(field)[0]
        snap = getattr(self, "_snapshot", None)
        if snap is None or snap[0] != self._sim.state_version:
            snap = self._snapshot = (self._sim.state_version, self._sim.read_agent(0))
        if field in self._DERIVED and getattr(self, "_pose_written", False):
            return self._derived_at_written_pose(snap[1])[field]
        return get(snap[1])

    def _derived_at_written_pose(self, a):
        """`env.cur_pos = ...` / `env.cur_angle = ...` (the reference's `_update_pos` call pattern, :1558) only writes the
        pose; the snapshot's tile / lane / proximity / reward / done are those of the last step.  The reference evaluates
        them from the current pose whenever asked, so after a pose write they come from dtsim_query at that pose
        (_compute_done_reward :1685-1705: invalid pose, then max_steps, then compute_reward)."""
        cache = getattr(self, "_derived_cache", None)
        if cache is not None and cache[0] == self._sim.state_version:
            return cache[1]
        pr = self._probe(np.array(a.pos, np.float64), float(a.angle))
        in_lane = bool(pr["in_lane"])
        lane = np.array([pr["dist"], pr["dot_dir"], pr["angle_deg"], pr["angle_rad"]], np.float64) if in_lane else np.zeros(4)
        if not pr["valid"]:
            done, code, reward = 1, 1, REWARD_INVALID_POSE
        elif int(a.step_count) >= self.max_steps:
            done, code, reward = 1, 2, 0.0
        else:
            done, code, reward = 0, 0, float(pr["reward"])
        d = {_ffi.FIELD_LANE: lane, _ffi.FIELD_PROX: np.float64(pr["prox"]), _ffi.FIELD_REWARD: np.float64(reward),
             _ffi.FIELD_TILE: np.array([pr["tile_i"], pr["tile_j"]], np.int32), _ffi.FIELD_IN_LANE: np.uint8(in_lane),
             _ffi.FIELD_DONE: np.uint8(done), _ffi.FIELD_DONE_CODE: np.uint8(code)}
        self._derived_cache = (self._sim.state_version, d)
        return d

    if gym is None:
        @property
        def unwrapped(self):                          # gym.Env.unwrapped
            return self

    @property
    def undistort(self) -> bool:
        """simulator.py:1968-1970, 2001: when set (UndistortWrapper does, wrappers.py:209), render_obs / render skip
        camera_model.distort and return the rectilinear image."""
        return self._undistort

    @undistort.setter
    def undistort(self, flag):
        self._undistort = bool(flag)
        if hasattr(self, "_sim"):
            self._sim.skip_distort(self._undistort)

    @property
    def cur_pos(self):
        return self._f(_ffi.FIELD_POS).copy()

    @cur_pos.setter
    def cur_pos(self, pos):                           # `self.cur_pos, self.cur_angle = _update_pos(self, action)`
        arr = self._sim.read(_ffi.FIELD_POS).copy()
        arr[0] = np.asarray(pos, np.float64)
        self._sim.write(_ffi.FIELD_POS, arr)
        self._pose_written = True                      # derived views now come from dtsim_query at this pose

    @property
    def cur_angle(self):
        return float(self._f(_ffi.FIELD_ANGLE))

    @cur_angle.setter
    def cur_angle(self, angle):
        arr = self._sim.read(_ffi.FIELD_ANGLE).copy()
        arr[0] = float(angle)
        self._sim.write(_ffi.FIELD_ANGLE, arr)
        self._pose_written = True

    @property
    def speed(self):
        return float(self._f(_ffi.FIELD_SPEED))

    @property
    def step_count(self):
        return int(self._f(_ffi.FIELD_STEP_COUNT))

    @property
    def timestamp(self):
        return float(self._f(_ffi.FIELD_TIMESTAMP))

    @property
    def np_random(self):
        return self._sim.env_state[0].np_random

    # --------------------------------------------------------------------- gym API --
    def seed(self, seed=None):
        self._sim.env_state[0].np_random = np.random.default_rng(seed)     # simulator.py:1043-1045
        return [seed]

    def close(self):
        for v in getattr(self, "_viewers", {}).values():
            v.close()
        self._viewers = {}

    def reset(self, segment: bool = False):
        self._pose_written = False
        self._sim.reset()
        if int(self._sim.env_map[0]) != self._map_idx:   # randomize_maps_on_reset: _load_map(map_name) (simulator.py:541-544)
            self._bind_map(int(self._sim.env_map[0]))
            for v in getattr(self, "_viewers", {}).values():
                v.close()
            self._viewers = {}
        st = self._sim.init_states[0]
        es = self._sim.env_state[0]
        if self.gl_light_capture and self._gl_modelview is not None:
            # glLightfv(GL_POSITION) in reset() (simulator.py:565-584) is transformed by the model-view current at the call: the one the LAST frame of
            # the previous episode left (the identity at the first reset, inside __init__).  The light the device takes is eye-space: hand it over so.
            lp = gl_light_to_eye(self._gl_modelview, list(st.light_pos))
            st.light_pos[:] = lp
            col = self._sim.read(_ffi.FIELD_COLORS).copy()
            col[0, 12:16] = lp
            self._sim.write(_ffi.FIELD_COLORS, col)
        self.randomization_settings = es.settings
        self.horizon_color = np.array(list(st.horizon_color))
        self.ground_color = np.array(list(st.ground_color))
        self.wheel_dist = st.wheel_dist
        self.cam_height, self.cam_angle, self.cam_fov_y = st.cam_height, [st.cam_angle_deg, 0, 0], st.cam_fov_y_deg
        return self.render_obs(segment=segment)

    def step(self, action: np.ndarray):
        action = np.clip(action, -1, 1) if self._ACTION_MODE == "wheels" else np.asarray(action)
        action = np.array(action, dtype=np.float64)
        self._pose_written = False                     # the step recomputes every derived field on the device
        self._sim.step(action.reshape(1, 2))
        obs = self.render_obs()                      # launched right behind the step: one wait for both kernels
        # Simulator.step receives the clipped wheel duties [u_l, u_r] (DuckietownEnv.step computes them from
        # (vel, steering), envs/duckietown_env.py:36-61); update_physics keeps them as last_action / wheelVels (:1555, 1564)
        wheels = np.array(self._f(_ffi.FIELD_WHEELS), dtype=np.float64)
        self.last_action = wheels
        self.wheelVels = wheels * self.robot_speed
        misc = self.get_agent_info()
        d = self._compute_done_reward()
        misc["Simulator"]["msg"] = d.done_why
        return obs, d.reward, d.done, misc

    def render_obs(self, segment: bool = False) -> np.ndarray:
        """simulator.py:1953-1972; `segment=True` is the segmentation render (:1730-1737, 1753, 1808, 1879).
        draw_curve / draw_bbox (:1776-1778, 1886-1918): the GL_LINE overlays are a post-pass on the resolved frame
        (dtsim_draw_lines); with draw_bbox the view is the reference's debugging camera 0.8 m above the robot, looking down."""
        self._note_modelview(False, self.draw_bbox)
        if self.draw_bbox:
            v = self._viewer(False, (self.camera_width, self.camera_height))
            self._sync_viewer(v, top_down=False, bbox=True)
            v.render(segment=bool(segment), gl_filter=self.gl_filter)
            if self.enable_leds and not segment:
                v.draw_leds(self._led_spheres())
            v.draw_lines(self._overlay_lines())
            return v.frames_host()[0]
        self._sim.render(segment=bool(segment), gl_filter=self.gl_filter)
        if self.enable_leds and not segment:
            self._sim.draw_leds(self._led_spheres())
        if self.draw_curve:
            self._sim.draw_lines(self._overlay_lines())
        return self._sim.frames_host()[0]

    # ---------------------------------------------------------------- LEDs --
    def _led_spheres(self) -> np.ndarray:
        """World-space spheres [n, 8] = centre, radius, glColor, alpha of WorldObj.render_mesh's LEDs (objects.py:68-121), in draw order
        (BatchedSimulator.led_spheres); dtsim_draw_leds blends them into the rendered frame."""
        return self._sim.led_spheres([0])[0]

    # ---------------------------------------------------------------- GL_LINE overlays --
    def _overlay_lines(self) -> np.ndarray:
        """World-space segments [n, 9] of the reference's line overlays, in its draw order.
        draw_curve (simulator.py:1886-1904): per drivable tile (the tile loop's order) the curve whose chord has the largest dot product with
        the loop's `angle` first, red, then the tile's other curves, blue (curve_overlay_segments: the reference compares with the TILE's
        orientation index, not the heading) -- bezier_draw (graphics.py:336-349): 20 points, 19 segments, at the height of the control points.  draw_bbox (:1907-1918, objects.py:131-139): the collision rectangle of every visible object,
        then the agent's (with the reference's shadowed angle: agent_bbox_angle), at y = 0.01, red."""
        out = []
        ang = float(self.cur_angle)
        if self.draw_curve:
            out.extend(curve_overlay_segments(self.grid, self.grid_width, self.grid_height))
        if self.draw_bbox:
            vis = self._sim.read(_ffi.FIELD_OBJ_VISIBLE)[0]
            cen, yrot = self._sim.read(_ffi.FIELD_OBJ_CENTER)[0], self._sim.read(_ffi.FIELD_OBJ_YROT)[0]
            loops = []
            for k, o in enumerate(self.objects):
                if not vis[k]:
                    continue
                c = np.asarray(o.corners, dtype=np.float64).reshape(4, 2)
                if o.dyn_slot >= 0:
                    now = np.array([cen[o.dyn_slot, 0], cen[o.dyn_slot, 1]])
                    if o.dyn_kind == 1:                                    # DuckieObj: the rectangle moves with the centre (objects.py:400-403)
                        c = c + (now - np.array([o.pos[0], o.pos[2]]))
                    else:                                                  # DuckiebotObj / CheckerboardObj: regenerated at the new pose (collision.py:64-79)
                        th = math.radians(float(yrot[o.dyn_slot]))
                        mn, mx, sc = o.min_coords, o.max_coords, o.scale
                        raw = [(mn[0] * sc, mn[2] * sc), (mx[0] * sc, mn[2] * sc), (mx[0] * sc, mx[2] * sc), (mn[0] * sc, mx[2] * sc)]
                        c = np.array([[now[0] + x * math.cos(th) + z * math.sin(th), now[1] - x * math.sin(th) + z * math.cos(th)] for x, z in raw])
                loops.append(c)
            # the agent's own rectangle (:1910-1918) -- at the pose's position but with `angle` as the tile loop above it left it (:1862 rebinds the
            # name): the orientation INDEX (0..3) of the last tile drawn, read as radians.  Reproduced as the reference draws it.
            loops.append(get_agent_corners(self.cur_pos, agent_bbox_angle(self.grid, self.grid_width, self.grid_height, ang)))
            for c in loops:
                for i in range(4):
                    a, b = c[i], c[(i + 1) % 4]
                    out.append([a[0], 0.01, a[1], b[0], 0.01, b[1], 1.0, 0.0, 0.0])
        return np.asarray(out, dtype=np.float32).reshape(-1, 9)

    def render(self, mode: str = "human", close: bool = False, segment: bool = False):
        """simulator.py:1974-2053: the WINDOW_WIDTH x WINDOW_HEIGHT view of the current state -- the agent camera
        ("human", "rgb_array", "free_cam"; the last without the fisheye) or the map from above with the agent's mesh
        drawn at its pose ("top_down", :1786-1798, 1920-1927).  The image is returned for every mode; there is no
        pyglet window here ("human" displays nothing and draws no text label)."""
        assert mode in ["human", "top_down", "free_cam", "rgb_array"]
        if close:
            return
        v = self._viewer(self.distortion and mode != "free_cam")
        self._sync_viewer(v, top_down=(mode == "top_down"), bbox=self.draw_bbox and mode != "top_down")
        self._note_modelview(mode == "top_down", self.draw_bbox and mode != "top_down")
        v.render(segment=bool(segment), gl_filter=self.gl_filter)
        if self.enable_leds and not segment:             # (the stand-in for self.mesh in the top-down view is not a WorldObj: no LEDs, as in the reference)
            v.draw_leds(self._led_spheres())
        if self.draw_curve or self.draw_bbox:
            v.draw_lines(self._overlay_lines())
        # (Every mode but "rgb_array" goes on, in the reference, to blit the image into a pyglet WINDOW -- its own GL context, with glOrtho left on
        # that context's model-view stack (simulator.py:2007-2022) -- and leaves that context current: a reset() called before the next frame sets
        # its light THERE, not in the context the frames are drawn in.  There is no window here; the capture goes through the camera as after
        # any frame.  DESIGN.md section 4 lists it with the deviations.)
        return v.frames_host()[0]

    def _note_modelview(self, top_down: bool, bbox: bool):
        """Remember the model-view the frame being drawn leaves behind (what the next reset()'s glLightfv is transformed by): the composite
        Rx(cam_angle) T(0, 0, forward) LookAt as this backend's camera -- eye C, yaw (sa, ca), pitch (sth, cth)."""
        st = self._sim.init_states[0]
        pos = np.asarray(self.cur_pos, dtype=np.float64)
        if self.domain_rand and not top_down:
            pos = pos + np.asarray(list(st.camera_noise), dtype=np.float64)          # simulator.py:1768-1769
        vp, va, vh, vdeg = viewer_camera(top_down, bbox, pos, float(self.cur_angle), self.grid_width, self.grid_height, self.road_tile_size, st.cam_fov_y_deg)
        h = st.cam_height if vh is None else vh
        deg = st.cam_angle_deg if vdeg is None else vdeg
        y0 = 0.0 if top_down else float(pos[1])       # (gluLookAt ignores the agent only in the top-down view; the bbox view sits 0.8 m above pos, simulator.py:1776-1778)
        sa, ca = math.sin(va), math.cos(va)
        self._gl_modelview = dict(C=np.array([vp[0] + CAMERA_FORWARD_DIST * ca, y0 + h, vp[2] - CAMERA_FORWARD_DIST * sa]), sa=sa, ca=ca,
                                  sth=math.sin(math.radians(deg)), cth=math.cos(math.radians(deg)))

    # ------------------------------------------------------------------ viewer --
    def _viewer(self, distortion: bool, size=None):
        """A second one-env handle at the window size that renders copies of this env's state: same map and assets,
        per-env camera enabled (so a top-down pose can be given to it), plus one extra non-static duckiebot that
        stands for `self.mesh` in the top-down view (hidden otherwise)."""
        import copy
        size = (WINDOW_WIDTH, WINDOW_HEIGHT) if size is None else (int(size[0]), int(size[1]))
        vkey = (bool(distortion), size)
        v = self._viewers.get(vkey) if hasattr(self, "_viewers") else None
        if v is not None:
            return v
        if not hasattr(self, "_viewers"):
            self._viewers = {}
        md = copy.deepcopy(self._sim.map_datas[self._map_idx])
        objs = md.get("objects") or []
        objs = [o for o in (list(objs.values()) if isinstance(objs, dict) else list(objs)) if o["kind"] != "floor_tag"]
        n_dyn = sum(1 for o in objs if not o.get("static", True))
        self._agent_marker = None
        if n_dyn < _ffi.MAX_DYNAMIC and len(objs) < _ffi.MAX_OBJECTS:
            self._agent_marker = (len(objs), n_dyn)                   # (object index, dynamic slot)
            marker = {"kind": "duckiebot", "pos": [0.5, 0.5], "rotate": 0, "static": False, "color": "red"}
            if self._sim.library.root:
                marker["scale"] = 1.0                                 # self.mesh.render() is unscaled (:1923-1926)
            else:
                marker["height"] = 0.12                               # the stand-in meshes are unit-height blobs
            objs.append(marker)
        md["objects"] = objs
        v = BatchedSimulator(self._sim._ctor_map_names[self._map_idx], 1, map_data=md, camera_width=size[0], camera_height=size[1],
                             distortion=bool(distortion), domain_rand=True, seed=0, max_steps=self.max_steps,
                             frame_rate=self.frame_rate, device=self._sim._device, style=self.style,
                             asset_root=self._sim.library.root, do_reset=False)
        self._viewers[vkey] = v
        return v

    def _sync_viewer(self, v, top_down: bool, bbox: bool = False):
        import ctypes as C
        st = (_ffi.InitState * 1)()
        C.memmove(st, C.byref(self._sim.init_states[0]), C.sizeof(_ffi.InitState))
        s0 = st[0]
        pos, ang = self.cur_pos, self.cur_angle
        if not self.domain_rand:
            s0.camera_noise[:] = [0.0, 0.0, 0.0]                      # drawn, but only applied under domain_rand (:1768-1769)
        vp, va, vh, vdeg = viewer_camera(top_down, bbox, pos, float(ang), self.grid_width, self.grid_height, self.road_tile_size, s0.cam_fov_y_deg)
        s0.pos[:] = vp
        s0.angle = va
        if vh is not None:
            s0.cam_height, s0.cam_angle_deg = vh, vdeg
        if top_down:
            s0.camera_noise[:] = [0.0, 0.0, 0.0]
        v.init_states = st
        v.reset(states=st)
        for f in (_ffi.FIELD_OBJ_CENTER, _ffi.FIELD_OBJ_YROT, _ffi.FIELD_OBJ_Y, _ffi.FIELD_OBJ_ACTIVE, _ffi.FIELD_OBJ_VISIBLE,
                  _ffi.FIELD_OBJ_LIGHT):
            arr = self._sim.read(f).copy()
            if self._agent_marker is not None:
                k, slot = self._agent_marker
                if f == _ffi.FIELD_OBJ_CENTER:
                    arr[0, slot] = [pos[0], pos[2]]
                elif f == _ffi.FIELD_OBJ_YROT:
                    arr[0, slot] = math.degrees(ang)                  # glRotatef(cur_angle * 180 / pi, 0, 1, 0)  (:1924)
                elif f == _ffi.FIELD_OBJ_VISIBLE:
                    arr[0, k] = 1 if top_down else 0
            v.write(f, arr)

    # --------------------------------------------------------- device-side queries --
    def _probe(self, pos, angle, safety_factor=1.0):
        p = self._sim.query(np.zeros(1, np.int32), np.array([[pos[0], pos[2], angle]], np.float64), safety_factor)
        return p[0]

    def get_grid_coords(self, abs_pos) -> Tuple[int, int]:
        pr = self._probe(abs_pos, 0.0)
        return int(pr["tile_i"]), int(pr["tile_j"])

    def _get_tile(self, i, j):
        i, j = int(i), int(j)
        if i < 0 or i >= self.grid_width or j < 0 or j >= self.grid_height:
            return None
        return self.grid[j * self.grid_width + i]

    def _get_curve(self, i, j):
        """simulator.py:1151: the Bezier control points [C,4,3] of a drivable tile."""
        tile = self._get_tile(i, j)
        assert tile is not None
        return tile["curves"]

    def _perturb(self, val, scale=0.1):
        """simulator.py:1065-1085 (consumes the env's RNG exactly like the reference)."""
        from dtsim import reset as _R
        return _R._perturb(self.np_random, self.domain_rand, val, scale)

    def _drivable_pos(self, pos) -> bool:
        return bool(self._probe(pos, 0.0)["drivable"])

    def closest_curve_point(self, pos, angle):
        pr = self._probe(pos, angle)
        if not pr["in_lane"]:
            return None, None
        return (np.array([pr["point"][0], 0.0, pr["point"][1]]), np.array([pr["tangent"][0], 0.0, pr["tangent"][1]]))

    def get_lane_pos2(self, pos, angle):
        pr = self._probe(pos, angle)
        if not pr["in_lane"]:
            raise NotInLane(f"Point not in lane: {pos}")
        return LanePosition(dist=float(pr["dist"]), dot_dir=float(pr["dot_dir"]), angle_deg=float(pr["angle_deg"]),
                            angle_rad=float(pr["angle_rad"]))

    def proximity_penalty2(self, pos, angle) -> float:
        return float(self._probe(pos, angle)["prox"])

    def _valid_pose(self, pos, angle, safety_factor: float = 1.0) -> bool:
        return bool(self._probe(pos, angle, safety_factor)["valid"])

    def _inconvenient_spawn(self, pos) -> bool:
        return bool(self._probe(pos, 0.0)["inconvenient"])

    def _collision(self, agent_corners) -> bool:
        """simulator.py:1473: takes the 4 corners of get_agent_corners(pos, angle); the pose is
        recovered from them (rear-left, rear-right, front-right, front-left) and evaluated on the device."""
        c = np.asarray(agent_corners, dtype=np.float64)
        centre = c.mean(axis=0)
        fwd = (c[2] + c[3]) * 0.5 - (c[0] + c[1]) * 0.5
        angle = math.atan2(-fwd[1], fwd[0])
        d = np.array([math.cos(angle), -math.sin(angle)])
        p = centre - (CAMERA_FORWARD_DIST - ROBOT_LENGTH / 2) * d
        return bool(self._probe([p[0], 0.0, p[1]], angle)["collision"])

    def compute_reward(self, pos, angle, speed):
        pr = self._probe(pos, angle)
        if speed == self.robot_speed:
            return float(pr["reward"])
        if not pr["in_lane"]:
            return 40 * float(pr["prox"])
        return +1.0 * speed * float(pr["dot_dir"]) + -10 * abs(float(pr["dist"])) + +40 * float(pr["prox"])

    def _compute_done_reward(self) -> DoneRewardInfo:
        code = int(self._f(_ffi.FIELD_DONE_CODE))
        msg = _DONE_MSG.get(code, "Stopping the simulator because we reached max_steps = %s" % self.max_steps)
        return DoneRewardInfo(done=bool(self._f(_ffi.FIELD_DONE)), done_why=msg, reward=float(self._f(_ffi.FIELD_REWARD)),
                              done_code=_ffi.DONE_CODES[code])

    def update_physics(self, action, delta_time: float = None):
        """simulator.py:1551-1584: ONE physics update (frame_skip is Simulator.step's loop, :1674): the pose advances by
        `_update_pos`, step_count += 1, timestamp += delta_time, speed, last_action / wheelVels, every object stepped once.
        dtsim_step_ex(DTSIM_STEP_ONE_UPDATE); reward / done are refreshed for the new state (pure functions of it; the
        reference evaluates them on demand in _compute_done_reward).  `delta_time` other than the env's own is not
        supported: `_update_pos` ignores it in the reference too (:2084), only timestamp / speed / objects would differ."""
        if delta_time is not None and abs(delta_time - self.delta_time) > 1e-12:
            raise NotImplementedError("update_physics(delta_time != env.delta_time)")
        action = np.asarray(action, np.float64)
        self.wheelVels = action * self.robot_speed * 1
        self._pose_written = False
        self._sim.step(action.reshape(1, 2), flags=_ffi.STEP_ONE_UPDATE)   # the wheel pair as given (no kinematics, no clip)
        self.last_action = action

    def get_agent_info(self) -> dict:
        info = {"action": list(self.last_action)}
        if self.full_transparency:
            lane, in_lane = self._f(_ffi.FIELD_LANE), bool(self._f(_ffi.FIELD_IN_LANE))
            if in_lane:
                info["lane_position"] = LanePosition(*[float(v) for v in lane]).as_json_dict()
            pos = self.cur_pos
            w = self._f(_ffi.FIELD_WHEELS)
            info["robot_speed"] = self.speed
            info["proximity_penalty"] = float(self._f(_ffi.FIELD_PROX))
            info["cur_pos"] = [float(pos[0]), float(pos[1]), float(pos[2])]
            info["cur_angle"] = self.cur_angle
            info["wheel_velocities"] = [float(w[0]) * self.robot_speed, float(w[1]) * self.robot_speed]
            info["timestamp"] = self.timestamp
            info["tile_coords"] = [int(v) for v in self._f(_ffi.FIELD_TILE)]
        return {"Simulator": info}

    def cartesian_from_weird(self, pos, angle) -> np.ndarray:
        gx, gy, gz = pos
        cp = [gx, self.grid_height * self.road_tile_size - gz]
        c, s = math.cos(angle), math.sin(angle)
        return np.array([[c, -s, cp[0]], [s, c, cp[1]], [0, 0, 1.0]])

    def weird_from_cartesian(self, q):
        return [q[0, 2], 0, self.grid_height * self.road_tile_size - q[1, 2]], math.atan2(q[1, 0], q[0, 0])


# ---- module-level helpers of the reference (simulator.py:2056-2116) ------------------
def gl_light_to_eye(mv: dict, light_pos) -> list:
    """A GL_POSITION 4-vector through the model-view `mv` (eye C, yaw sa / ca, pitch sth / cth; the composite of _render_img's
    Rx(cam_angle) T(0, 0, forward) LookAt): a position (w != 0) is taken relative to the eye and rotated, a direction (w = 0) only rotated."""
    L = [float(v) for v in light_pos] + [0.0] * (4 - len(light_pos))
    w = L[3]
    rx, ry, rz = (L[0] / w - mv["C"][0], L[1] / w - mv["C"][1], L[2] / w - mv["C"][2]) if w != 0.0 else (L[0], L[1], L[2])
    xla = rx * mv["sa"] + rz * mv["ca"]
    zla = -(rx * mv["ca"] - rz * mv["sa"])
    return [xla, ry * mv["cth"] - zla * mv["sth"], ry * mv["sth"] + zla * mv["cth"], 1.0 if w != 0.0 else 0.0]


def viewer_camera(top_down: bool, bbox: bool, pos, ang: float, grid_width: int, grid_height: int, tile_size: float, fov_y_deg: float):
    """The window / debugging views of _render_img as parameters of THIS backend's camera model -- eye = pos + CAMERA_FORWARD_DIST * dir +
    (0, cam_height, 0), dir = (cos, 0, -sin), pitched down by cam_angle_deg -- returned as (pos, angle, cam_height or None, cam_angle_deg):
      top_down  gluLookAt((a, H, b), (a, 0, b - 0.01), +y), H = (max(a, b) + 0.1) / tan(fov_y / 2)   (simulator.py:1786-1798)
      bbox      draw_bbox: y += 0.8, glRotatef(90, 1, 0, 0), no forward offset                        (:1776-1778)
      else      the agent camera at the current pose (cam_height / cam_angle_deg stay the env's)."""
    if top_down:
        a, b = grid_width * tile_size / 2, grid_height * tile_size / 2
        h_from_floor = (max(a, b) + 0.1) / math.tan(math.radians(fov_y_deg) / 2)
        return [a, 0.0, b + CAMERA_FORWARD_DIST], math.pi / 2, h_from_floor, math.degrees(math.atan2(h_from_floor, 0.01))
    if bbox:
        d = get_dir_vec(float(ang))
        return [float(pos[0]) - CAMERA_FORWARD_DIST * d[0], 0.0, float(pos[2]) - CAMERA_FORWARD_DIST * d[2]], float(ang), 0.8, 90.0
    return [float(pos[0]), 0.0, float(pos[2])], float(ang), None, None


def agent_bbox_angle(grid, grid_width: int, grid_height: int, angle: float) -> float:
    """The angle _render_img hands to get_agent_corners for the agent's draw_bbox rectangle (simulator.py:1910-1911): its tile loop
    (`for i, j in itertools.product(range(grid_width), range(grid_height))`, :1853) rebinds `angle = tile["angle"]` (:1862), so the rectangle
    is drawn with the orientation index of the LAST non-empty tile in that order, as radians; the pose's angle only if the map has no tile."""
    for i in range(grid_width - 1, -1, -1):
        for j in range(grid_height - 1, -1, -1):
            t = grid[j * grid_width + i]
            if t is not None:
                return float(t["angle"])
    return float(angle)


def curve_overlay_segments(grid, grid_width: int, grid_height: int) -> list:
    """draw_curve (simulator.py:1853-1904) as _render_img draws it: per drivable tile, in the tile loop's order (`itertools.product(range(grid_width),
    range(grid_height))`: columns outer), first the curve whose chord has the largest dot product with get_dir_vec(angle), red, then the tile's other
    curves, blue -- where `angle` is NOT the agent's heading: the loop rebinds the name to the tile's orientation index (:1862), so the reference
    direction is get_dir_vec(0 .. 3 radians) per tile.  Reproduced as drawn.  Each curve as bezier_draw draws it (graphics.py:336-349): 20 points
    at t = i / 19, a line strip = 19 segments.  Rows (ax, ay, az, bx, by, bz, r, g, b)."""
    out = []
    ts = np.arange(20, dtype=np.float64) / 19.0
    for i in range(grid_width):
        for j in range(grid_height):
            tile = grid[j * grid_width + i]
            if tile is None or not tile["drivable"]:
                continue
            curves = np.asarray(tile["curves"], dtype=np.float64)
            heads = curves[:, -1, :] - curves[:, 0, :]
            heads = heads / np.linalg.norm(heads).reshape(1, -1)           # (the reference's scalar norm: the argmax is unaffected)
            best = int(np.argmax(np.dot(heads, get_dir_vec(float(tile["angle"])))))
            for idx in [best] + [k for k in range(len(curves)) if k != best]:
                cps = curves[idx]
                pts = np.stack([bezier_point(cps, t) for t in ts])
                col = (1.0, 0.0, 0.0) if idx == best else (0.0, 0.0, 1.0)
                for p0, p1 in zip(pts[:-1], pts[1:]):
                    out.append([*p0, *p1, *col])
    return out


def bezier_point(cps, t):
    """graphics.py:286-295: cubic Bezier point of the control points [4, 3] at t."""
    cps = np.asarray(cps, dtype=np.float64)
    return (1 - t) ** 3 * cps[0] + 3 * t * (1 - t) ** 2 * cps[1] + 3 * t ** 2 * (1 - t) * cps[2] + t ** 3 * cps[3]


def get_dir_vec(cur_angle: float) -> np.ndarray:
    return np.array([math.cos(cur_angle), 0, -math.sin(cur_angle)])


def get_right_vec(cur_angle: float) -> np.ndarray:
    return np.array([math.sin(cur_angle), 0, math.cos(cur_angle)])


def _actual_center(pos, angle):
    return pos + (CAMERA_FORWARD_DIST - (ROBOT_LENGTH / 2)) * get_dir_vec(angle)


def get_agent_corners(pos, angle):
    """simulator.py:2112 / collision.py:9 -- pure formatting of the pose for callers that pass
    the corners back into Simulator._collision (run_tests.py:50); no hot-path arithmetic."""
    c = _actual_center(np.asarray(pos, dtype=np.float64), angle)
    f, r = get_dir_vec(angle), get_right_vec(angle)
    hw, hl = 0.5 * ROBOT_WIDTH, 0.5 * ROBOT_LENGTH
    return np.array([c - hw * r - hl * f, c + hw * r - hl * f, c + hw * r + hl * f, c - hw * r + hl * f])[:, [0, 2]]


def _update_pos(self, action):
    """simulator.py:2076-2088: advance the dynamics state by one delta_time with the wheel pair `action` and return
    (pos, angle).  dtsim_step_ex(DTSIM_STEP_POSE_ONLY): step_count, timestamp, speed, the objects, reward and done are
    untouched.  The device pose fields already hold the returned pose (the reference leaves `self.cur_pos` to the
    caller's assignment, `self.cur_pos, self.cur_angle = _update_pos(self, action)`, which is accepted and a no-op here)."""
    self._sim.step(np.asarray(action, np.float64).reshape(1, 2), flags=_ffi.STEP_POSE_ONLY)
    return self.cur_pos, self.cur_angle
