"""Host half of Simulator.reset() (simulator.py:528-763): the RNG-ORDER logic only.

The reference's reset is a sequence of numpy-RNG draws whose order decides everything
(SURVEY.md 8a row 14).  The draws stay on the host, on the very generator class the
reference uses (gym>=0.22: Generator(PCG64)), in the reference's order.  All
*geometry* the reset needs -- _inconvenient_spawn, _valid_pose(safety_factor=1.3),
get_lane_pos2 -- is evaluated on the GPU through dtsim_query; there is no host copy of it.

Spawn rejection loop (simulator.py:692-738): each attempt consumes exactly three doubles
(x, z, angle).  We therefore draw a block of attempts from a *clone* of the generator,
evaluate the whole block on the device in one call, pick the first accepted attempt `a`
and consume 3*(a+1) draws from the real generator -- bit-identical to the sequential loop.
"""
from __future__ import annotations

import copy
import math
from typing import List, Optional, Sequence

import numpy as np

from . import _ffi

# simulator.py:107-177
BLUE_SKY = np.array([0.45, 0.82, 1])
WALL_COLOR = np.array([0.64, 0.71, 0.28])
DIM = 0.5
CAMERA_ANGLE, CAMERA_FOV_Y, CAMERA_FLOOR_DIST, WHEEL_DIST = 19.15, 75, 0.108, 0.102
MAX_SPAWN_ATTEMPTS = 5000
ATTEMPT_BLOCK = 32

# randomization/randomizer.py:8-16 + config/default_dr.json (identical content)
DR_CONFIG = {
    "horz_mode": {"type": "int", "low": 0, "high": 4},
    "light_pos": {"type": "uniform", "low": [-150, 170, -150], "high": [150, 220, 150], "size": 3},
    "camera_noise": {"type": "uniform", "low": -0.005, "high": 0.005, "size": 3},
    "trim": {"type": "normal", "loc": 0, "scale": 0.02},
    "camera_height": {"type": "uniform", "low": 0.92, "high": 1.08},
    "camera_angle": {"type": "uniform", "low": 0.8, "high": 1.2},
    "camera_fov_y": {"type": "uniform", "low": 0.8, "high": 1.2},
}
_DR_KEYS = sorted(DR_CONFIG)   # randomizer.py:34: parameters are generated in sorted-key order


def randomize(rng: np.random.Generator) -> dict:
    """Randomizer.randomize (randomizer.py:36-91): all 7 keys are always drawn."""
    out = {}
    for k in _DR_KEYS:
        d = DR_CONFIG[k]
        size = d.get("size", 1)
        if d["type"] == "int":
            out[k] = rng.integers(low=d["low"], high=d["high"], size=size)   # randint -> integers fallback :59-62
        elif d["type"] == "uniform":
            out[k] = rng.uniform(low=d["low"], high=d["high"], size=size)
        else:
            out[k] = rng.normal(loc=d["loc"], scale=d["scale"], size=size)
    return out


class EnvResetState:
    """Per-env host state of the reset logic: the RNG and what the last reset decided."""

    def __init__(self, seed):
        self.np_random = np.random.default_rng(seed)   # simulator.py:1043-1045
        self.settings: dict = {}
        self.tile_colors = None
        self.obj_colors = None
        self.spawn_attempts = 0
        self.map_slot = 0          # MultiMapEnv.cur_env_idx (envs/multimap_env.py:46)


def _perturb(rng, domain_rand, val, scale=0.1):
    """simulator.py:1065-1085"""
    val = np.array(val)
    if not domain_rand:
        return val
    noise = rng.uniform(low=1 - scale, high=1 + scale, size=val.shape)
    if val.size == 4:
        noise[3] = 1
    return val * noise


def draw_prefix(es: EnvResetState, mt, *, domain_rand, camera_rand, dynamics_rand, color_sky, color_ground,
                num_tris_distractors, n_visible_draw: Sequence[bool], user_tile_start):
    """Everything reset() draws before the spawn loop, in order (simulator.py:546-676).
    Returns (InitState with all DR fields filled, start tile (i,j), visible flags)."""
    rng = es.np_random
    rs = es.settings = randomize(rng)
    st = _ffi.InitState()
    if domain_rand:
        hm = int(rs["horz_mode"][0])
        horizon = (_perturb(rng, True, color_sky) if hm == 0 else
                   _perturb(rng, True, WALL_COLOR) if hm == 1 else
                   _perturb(rng, True, [0.15, 0.15, 0.15], 0.4) if hm == 2 else
                   _perturb(rng, True, [0.9, 0.9, 0.9], 0.4))
        light_pos = [float(v) for v in rs["light_pos"]] + [0.0]     # 3 components => w = 0 (glLightfv 4-vector)
    else:
        horizon = np.array(color_sky)
        light_pos = [0.0, 3.0, 0.0, 1.0]
    ambient = _perturb(rng, domain_rand, np.array([0.50 * DIM, 0.50 * DIM, 0.50 * DIM, 1]), 0.3)
    diffuse = _perturb(rng, domain_rand, np.array([0.70 * DIM, 0.70 * DIM, 0.70 * DIM, 1]), 0.99)
    ground = _perturb(rng, domain_rand, np.array(color_ground), 0.3)
    wheel_dist = _perturb(rng, domain_rand, WHEEL_DIST)
    cam_height, cam_angle, cam_fov = CAMERA_FLOOR_DIST, CAMERA_ANGLE, CAMERA_FOV_Y
    if domain_rand or camera_rand:
        cam_height = cam_height * float(rs["camera_height"][0])
        cam_angle = CAMERA_ANGLE * float(rs["camera_angle"][0])
        cam_fov = cam_fov * float(rs["camera_fov_y"][0])
    for _ in range(0, 3 * num_tris_distractors):       # never visible, but they consume RNG (:621-631)
        rng.uniform(low=[-20, -0.6, -20], high=[20, -0.3, 20], size=(3,))
        c = rng.uniform(low=0, high=0.9)
        _perturb(rng, domain_rand, [c, c, c], 0.1)
    n_tiles = sum(1 for k in mt.tile_kind_names if k is not None)
    es.tile_colors = [_perturb(rng, domain_rand, [1, 1, 1, 1], 0.2) for _ in range(n_tiles)]   # :634-645
    visible = []
    es.obj_colors = []
    for o in mt.objects:                                                                       # :648-656
        es.obj_colors.append(_perturb(rng, domain_rand, [1, 1, 1, 1], 0.3))
        if o.optional and domain_rand:
            visible.append(bool(rng.integers(0, 2) == 0))
        else:
            visible.append(True)
    if user_tile_start:
        tile = tuple(user_tile_start)
        ti = int(tile[1]) * mt.grid_w + int(tile[0])
        if not (0 <= tile[0] < mt.grid_w and 0 <= tile[1] < mt.grid_h) or mt.tile_kind_names[ti] is None:
            raise Exception("The tile specified does not exist.")
    elif mt.start_tile is not None:
        tile = mt.start_tile
    else:
        if not mt.drivable_tiles:
            raise Exception("There are no drivable tiles. Use start_tile or self.user_tile_start")
        tile = mt.drivable_tiles[int(rng.integers(0, len(mt.drivable_tiles)))]
    st.dynamics_trim_on = 1 if dynamics_rand else 0
    st.dynamics_trim = float(0 + rs["trim"][0])
    st.wheel_dist = float(wheel_dist)
    st.cam_height, st.cam_angle_deg, st.cam_fov_y_deg = float(cam_height), float(cam_angle), float(cam_fov)
    st.camera_noise[:] = [float(v) for v in rs["camera_noise"]]
    st.horizon_color[:] = [float(v) for v in horizon]
    st.ground_color[:] = [float(v) for v in ground]
    st.light_pos[:] = light_pos
    st.light_ambient[:] = [float(v) for v in ambient[:3]]
    st.light_diffuse[:] = [float(v) for v in diffuse[:3]]
    return st, tile, visible


def attempt_block(es: EnvResetState, tile, ts, k=ATTEMPT_BLOCK):
    """k spawn attempts drawn from a clone of the env's generator (simulator.py:695-701)."""
    clone = copy.deepcopy(es.np_random)
    u = clone.random(3 * k).reshape(k, 3)
    i, j = tile
    x = (i + ((i + 1) - i) * u[:, 0]) * ts       # Generator.uniform(low, high) = low + (high-low)*u
    z = (j + ((j + 1) - j) * u[:, 1]) * ts
    ang = 0 + (2 * math.pi - 0) * u[:, 2]
    return np.stack([x, z, ang], axis=1)


def commit_attempts(es: EnvResetState, n_attempts: int):
    # consume exactly the draws the sequential loop would have (PCG64.advance() would also
    # drop a buffered uint32 half-draw left by integers(), which the reference keeps)
    es.np_random.random(3 * n_attempts)
    es.spawn_attempts += n_attempts
