"""Oracle restatement of the reference's per-step simulation path (no render).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  numpy / float64, one env
per object, written to follow the reference's arithmetic *operation by
operation* (same numpy calls where rounding could matter).  All citations are
paths under /root/reference/.

Pinned against the reference's own code by tests/test_oracle_vs_reference.py
(build container) and tests/golden/ (everywhere).  The dynamics integrator
(duckietown_world, not in /root/reference) is restated from its published
model -- PARITY UNPINNED, see DynamicsDB18 below.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

# ---- constants: src/gym_duckietown/simulator.py:99-177 -------------------
CAMERA_ANGLE = 19.15
CAMERA_FOV_Y = 75
CAMERA_FLOOR_DIST = 0.108
CAMERA_FORWARD_DIST = 0.066
WHEEL_DIST = 0.102
ROBOT_WIDTH = 0.13 + 0.02
ROBOT_LENGTH = 0.18
SAFETY_RAD_MULT = 1.8
AGENT_SAFETY_RAD = (max(ROBOT_LENGTH, ROBOT_WIDTH) / 2) * SAFETY_RAD_MULT
MIN_SPAWN_OBJ_DIST = 0.25
DEFAULT_ROBOT_SPEED = 1.20
REWARD_INVALID_POSE = -1000
MAX_SPAWN_ATTEMPTS = 5000
BLUE_SKY = np.array([0.45, 0.82, 1])
WALL_COLOR = np.array([0.64, 0.71, 0.28])
DIM = 0.5
DRIVABLE_TILES = ["straight", "curve_left", "curve_right", "3way_left", "3way_right", "4way"]

DONE_IN_PROGRESS = 0  # "in-progress"        simulator.py:1704
DONE_INVALID_POSE = 1  # "invalid-pose"      simulator.py:1691
DONE_MAX_STEPS = 2  # "max-steps-reached"    simulator.py:1699
DONE_CODES = ["in-progress", "invalid-pose", "max-steps-reached"]


class NotInLane(Exception):
    """exceptions.py:14"""


# ---- graphics.py math ------------------------------------------------------
def rotate_point(px, py, cx, cy, theta):
    """graphics.py:254-265"""
    dx = px - cx
    dy = py - cy
    new_dx = dx * math.cos(theta) + dy * math.sin(theta)
    new_dy = dy * math.cos(theta) - dx * math.sin(theta)
    return cx + new_dx, cy + new_dy


def gen_rot_matrix(axis0, angle):
    """graphics.py:268-283"""
    axis = axis0 / math.sqrt(np.dot(axis0, axis0))
    a = math.cos(angle / 2.0)
    b, c, d = -axis * math.sin(angle / 2.0)
    return np.array(
        [
            [a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
            [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
            [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c],
        ]
    )


def bezier_point(cps, t):
    """graphics.py:286-297"""
    p = ((1 - t) ** 3) * cps[0, :]
    p = p + 3 * t * ((1 - t) ** 2) * cps[1, :]
    p = p + 3 * (t**2) * (1 - t) * cps[2, :]
    p = p + (t**3) * cps[3, :]
    return p


def bezier_tangent(cps, t):
    """graphics.py:300-313"""
    p = 3 * ((1 - t) ** 2) * (cps[1, :] - cps[0, :])
    p = p + 6 * (1 - t) * t * (cps[2, :] - cps[1, :])
    p = p + 3 * (t**2) * (cps[3, :] - cps[2, :])
    norm = np.linalg.norm(p)
    return p / norm


def bezier_closest(cps, p, t_bot=0, t_top=1, n=8):
    """graphics.py:316-333 -- 8-level endpoint-distance bisection, strict <."""
    for _ in range(n):
        mid = (t_bot + t_top) * 0.5
        d_bot = np.linalg.norm(bezier_point(cps, t_bot) - p)
        d_top = np.linalg.norm(bezier_point(cps, t_top) - p)
        if d_bot < d_top:
            t_top = mid
        else:
            t_bot = mid
    return (t_bot + t_top) * 0.5


# ---- free functions simulator.py:2056-2116 ---------------------------------
def get_dir_vec(angle):
    return np.array([math.cos(angle), 0, -math.sin(angle)])


def get_right_vec(angle):
    return np.array([math.sin(angle), 0, math.cos(angle)])


def actual_center(pos, angle):
    """simulator.py:2102-2109"""
    return pos + (CAMERA_FORWARD_DIST - (ROBOT_LENGTH / 2)) * get_dir_vec(angle)


def agent_boundbox(true_pos, width, length, f_vec, r_vec):
    """collision.py:9-34"""
    hwidth = 0.5 * width
    hlength = 0.5 * length
    return np.array(
        [
            true_pos - hwidth * r_vec - hlength * f_vec,
            true_pos + hwidth * r_vec - hlength * f_vec,
            true_pos + hwidth * r_vec + hlength * f_vec,
            true_pos - hwidth * r_vec + hlength * f_vec,
        ]
    )[:, [0, 2]]


def get_agent_corners(pos, angle):
    """simulator.py:2112-2116 (applies _actual_center itself)."""
    return agent_boundbox(actual_center(pos, angle), ROBOT_WIDTH, ROBOT_LENGTH,
                          get_dir_vec(angle), get_right_vec(angle))


# ---- collision.py ----------------------------------------------------------
def generate_corners(pos, min_coords, max_coords, theta, scale):
    """collision.py:64-79"""
    px = pos[0]
    pz = pos[-1]
    return np.array(
        [
            rotate_point(min_coords[0] * scale + px, min_coords[-1] * scale + pz, px, pz, theta),
            rotate_point(max_coords[0] * scale + px, min_coords[-1] * scale + pz, px, pz, theta),
            rotate_point(max_coords[0] * scale + px, max_coords[-1] * scale + pz, px, pz, theta),
            rotate_point(min_coords[0] * scale + px, max_coords[-1] * scale + pz, px, pz, theta),
        ]
    )


def generate_norm(corners):
    """collision.py:99-106 -- eigenvectors of the corner covariance."""
    ca = np.cov(corners, y=None, rowvar=False, bias=True)
    _, vect = np.linalg.eig(ca)
    return vect.T


def _sat(norm, corners):
    """collision.py:37-47"""
    dotval = np.matmul(norm, corners)
    return np.min(dotval, axis=-1), np.max(dotval, axis=-1)


def _overlaps(min1, max1, min2, max2):
    """collision.py:50-61 -- closed intervals."""
    return (min1 <= min2 <= max1) or (min2 <= min1 <= max2)


def intersects(duckie, objs_stacked, duckie_norm, norms_stacked):
    """collision.py:129-159"""
    dd_min, dd_max = _sat(duckie_norm, duckie.T)
    od_min, od_max = _sat(duckie_norm, objs_stacked)
    do_min, do_max = _sat(norms_stacked, duckie.T)
    oo_min, oo_max = _sat(norms_stacked, objs_stacked)
    for idx in range(od_min.shape[0]):
        if not _overlaps(dd_min[0], dd_max[0], od_min[idx][0], od_max[idx][0]):
            continue
        if not _overlaps(dd_min[1], dd_max[1], od_min[idx][1], od_max[idx][1]):
            continue
        if not _overlaps(do_min[idx][0], do_max[idx][0], oo_min[idx][0], oo_max[idx][0]):
            continue
        if not _overlaps(do_min[idx][1], do_max[idx][1], oo_min[idx][1], oo_max[idx][1]):
            continue
        return True
    return False


def intersects_single_obj(duckie, obj, duckie_norm, norm):
    """collision.py:162-186"""
    return intersects(duckie, obj[np.newaxis], duckie_norm, norm[np.newaxis])


def calculate_safety_radius(min_coords, max_coords, scale):
    """collision.py:214-220"""
    x, _, z = np.max([abs(min_coords), abs(max_coords)], axis=0)
    return np.linalg.norm([x, z]) * scale


# ---- map interpretation: simulator.py:788-1038, 1151-1335 -----------------
_CURVES = {
    "straight": [
        [[-0.20, 0, -0.50], [-0.20, 0, -0.25], [-0.20, 0, 0.25], [-0.20, 0, 0.50]],
        [[0.20, 0, 0.50], [0.20, 0, 0.25], [0.20, 0, -0.25], [0.20, 0, -0.50]],
    ],
    "curve_left": [
        [[-0.20, 0, -0.50], [-0.20, 0, 0.00], [0.00, 0, 0.20], [0.50, 0, 0.20]],
        [[0.50, 0, -0.20], [0.30, 0, -0.20], [0.20, 0, -0.30], [0.20, 0, -0.50]],
    ],
    "curve_right": [
        [[-0.20, 0, -0.50], [-0.20, 0, -0.20], [-0.30, 0, -0.20], [-0.50, 0, -0.20]],
        [[-0.50, 0, 0.20], [-0.30, 0, 0.20], [0.30, 0, 0.00], [0.20, 0, -0.50]],
    ],
    "3way": [
        [[-0.20, 0, -0.50], [-0.20, 0, -0.25], [-0.20, 0, 0.25], [-0.20, 0, 0.50]],
        [[-0.20, 0, -0.50], [-0.20, 0, 0.00], [0.00, 0, 0.20], [0.50, 0, 0.20]],
        [[0.20, 0, 0.50], [0.20, 0, 0.25], [0.20, 0, -0.25], [0.20, 0, -0.50]],
        [[0.50, 0, -0.20], [0.30, 0, -0.20], [0.20, 0, -0.20], [0.20, 0, -0.50]],
        [[0.20, 0, 0.50], [0.20, 0, 0.20], [0.30, 0, 0.20], [0.50, 0, 0.20]],
        [[0.50, 0, -0.20], [0.30, 0, -0.20], [-0.20, 0, 0.00], [-0.20, 0, 0.50]],
    ],
    "4way": [
        [[-0.20, 0, -0.50], [-0.20, 0, 0.00], [0.00, 0, 0.20], [0.50, 0, 0.20]],
        [[-0.20, 0, -0.50], [-0.20, 0, -0.25], [-0.20, 0, 0.25], [-0.20, 0, 0.50]],
        [[-0.20, 0, -0.50], [-0.20, 0, -0.20], [-0.30, 0, -0.20], [-0.50, 0, -0.20]],
    ],
}


def get_curve(kind, angle, i, j, ts):
    """simulator.py:1151-1335"""
    if kind.startswith("straight"):
        pts = np.array(_CURVES["straight"]) * ts
    elif kind == "curve_left":
        pts = np.array(_CURVES["curve_left"]) * ts
    elif kind == "curve_right":
        pts = np.array(_CURVES["curve_right"]) * ts
    elif kind.startswith("3way"):
        pts = np.array(_CURVES["3way"]) * ts
    elif kind.startswith("4way"):
        pts = np.array(_CURVES["4way"]) * ts
    else:
        raise ValueError(f"Cannot get bezier for kind {kind}")
    centre = np.array([(i + 0.5) * ts, 0, (j + 0.5) * ts])
    if kind.startswith("4way"):
        out = []
        for rot in np.arange(0, 4):
            mat = gen_rot_matrix(np.array([0, 1, 0]), rot * math.pi / 2)
            pts_new = np.matmul(pts, mat)
            pts_new += centre
            out.append(pts_new)
        return np.reshape(np.array(out), (12, 4, 3))
    mat = gen_rot_matrix(np.array([0, 1, 0]), angle * math.pi / 2)
    pts = np.matmul(pts, mat)
    pts += centre
    return pts


@dataclass
class OracleObj:
    """objects.py:23-66 (WorldObj) + :339-365 (DuckieObj state)."""
    kind: str
    pos: np.ndarray
    angle: float
    scale: float
    static: bool
    optional: bool
    min_coords: np.ndarray
    max_coords: np.ndarray
    safety_radius: float
    obj_corners: np.ndarray  # [4,2]
    obj_norm: np.ndarray  # [2,2]
    visible: bool = True
    y_rot: float = 0.0
    color: np.ndarray = field(default_factory=lambda: np.array([0.0, 0, 0, 1]))
    # DuckieObj dynamic state (objects.py:339-365)
    walk_distance: float = 0.0
    pedestrian_wait_time: float = 8
    vel: float = 0.02
    heading: Optional[np.ndarray] = None
    start: Optional[np.ndarray] = None
    center: Optional[np.ndarray] = None
    pedestrian_active: bool = False
    wiggle: float = math.pi / 15
    time: float = 0.0
    # DuckiebotObj parameters (objects.py:180-228, non-DR branch)
    follow_dist: float = 0.3
    velocity: float = 0.1
    gain: float = 2.0
    trim: float = 0.0
    # CheckerboardObj (objects.py:479-505)
    steps: int = -20
    reset_start: Optional[np.ndarray] = None
    # TrafficLightObj (objects.py:434-463, non-DR branch): freq 5 s, pattern 0
    light_freq: int = 0
    light_pattern: int = 0
    radius: float = 0.0318
    wheel_dist: float = WHEEL_DIST
    robot_width: float = ROBOT_WIDTH
    robot_length: float = ROBOT_LENGTH
    k: float = 27.0
    limit: float = 1.0
    max_iterations: int = 1000

    # objects.py:230-270 DuckiebotObj.step_duckiebot (pure pursuit on the lane curve)
    def step_duckiebot(self, delta_time, closest_curve_point):
        closest_point, closest_tangent = closest_curve_point(self.pos, self.angle)
        if closest_point is None or closest_tangent is None:
            raise Exception(f"Cannot find closest point/tangent from {self.pos}, {self.angle} ")
        iterations = 0
        lookup_distance = self.follow_dist
        curve_point = None
        while iterations < self.max_iterations:
            follow_point = closest_point + closest_tangent * lookup_distance
            curve_point, _ = closest_curve_point(follow_point, self.angle)
            if curve_point is not None:
                break
            iterations += 1
            lookup_distance *= 0.5
        point_vec = curve_point - self.pos
        point_vec /= np.linalg.norm(point_vec)
        dot = np.dot(get_right_vec(self.angle), point_vec)
        steering = self.gain * -dot
        self._update_pos_bot([self.velocity, steering], delta_time)

    # objects.py:283-336 DuckiebotObj._update_pos
    def _update_pos_bot(self, action, deltaTime):
        vel, angle = action
        k_r_inv = (self.gain + self.trim) / self.k
        k_l_inv = (self.gain - self.trim) / self.k
        omega_r = (vel + 0.5 * angle * self.wheel_dist) / self.radius
        omega_l = (vel - 0.5 * angle * self.wheel_dist) / self.radius
        u_r = omega_r * k_r_inv
        u_l = omega_l * k_l_inv
        u_r_limited = max(min(u_r, self.limit), -self.limit)
        u_l_limited = max(min(u_l, self.limit), -self.limit)
        if u_l_limited == u_r_limited:
            self.pos = self.pos + deltaTime * u_l_limited * get_dir_vec(self.angle)
            return
        w = (u_r_limited - u_l_limited) / self.wheel_dist
        r = (self.wheel_dist * (u_l_limited + u_r_limited)) / (2 * (u_l_limited - u_r_limited))
        rotAngle = w * deltaTime
        r_vec = get_right_vec(self.angle)
        px, py, pz = self.pos
        cx = px + r * r_vec[0]
        cz = pz + r * r_vec[2]
        npx, npz = rotate_point(px, pz, cx, cz, rotAngle)
        self.pos = np.array([npx, py, npz])
        self.angle += rotAngle
        self.y_rot += rotAngle * 180 / np.pi
        # corners refreshed, obj_norm deliberately NOT (objects.py:333-336 vs :270)
        self.obj_corners = agent_boundbox(self.pos, self.robot_width, self.robot_length,
                                          get_dir_vec(self.angle), get_right_vec(self.angle))

    # objects.py:384-431
    def step(self, delta_time):
        if self.kind == "trafficlight":               # objects.py:455-463
            self.time += delta_time
            if round(self.time, 3) % self.light_freq == 0:
                self.light_pattern ^= 1
            return
        if self.static or self.kind == "duckiebot":
            return
        if self.kind == "checkerboard":               # objects.py:531-587
            self.time += delta_time
            step, off, move = self.steps, 20 * 1.0 / 3000, True
            d = np.zeros(3)
            if step < 0:
                pass
            elif step < 40:
                d[0] = off
            elif step < 135:
                d[0] = -off
            elif step < 170:
                d[0] = off
            elif step < 200:
                d[2] = off
            elif step < 260:
                d[2] = -off
            elif step < 290:
                d[2] = off
            elif step < 310:
                d[1] = off
            elif step < 330:
                d[1] = -off
            elif step < 355:
                d[0] = -off
            elif step < 370:
                d[1] = -off
            elif step < 385:
                d[1] = off
            elif step < 420:
                d[0] = off
            else:
                self.center = np.copy(self.reset_start)
                self.steps = -20
                move = False
            if move:
                self.center = self.center + d
                self.steps += 2
            self.pos = self.center
            return
        self.time += delta_time
        if not self.pedestrian_active:
            self.pedestrian_wait_time -= delta_time
            if self.pedestrian_wait_time <= 0:
                self.pedestrian_active = True
            return
        vel_adjust = self.heading * self.vel
        self.center = self.center + vel_adjust
        self.obj_corners = self.obj_corners + vel_adjust[[0, -1]]
        distance = np.linalg.norm(self.center - self.start)
        if distance > self.walk_distance:
            # finish_walk objects.py:413-431 (non-DR branch)
            self.start = np.copy(self.center)
            self.angle += np.pi
            self.pedestrian_active = False
            self.vel *= -1
            self.pedestrian_wait_time = 8
        self.pos = self.center
        angle_delta = self.wiggle * math.sin(48 * self.time)
        self.y_rot = (self.angle + angle_delta) * (180 / np.pi)
        # obj_norm is re-derived from translated corners in the reference
        # (objects.py:411); translation-invariant up to eig noise, kept fixed
        # here: non-square footprints only.

    def check_collision(self, agent_corners, agent_norm):
        """objects.py:152-160 (static -> False), :367-371 (dynamic)."""
        if self.static:
            return False
        return intersects_single_obj(agent_corners, self.obj_corners.T, agent_norm, self.obj_norm)

    def proximity(self, agent_pos, agent_safety_rad):
        """objects.py:162-170 (static -> 0), :373-382 (dynamic)."""
        if self.static:
            return 0.0
        if self.kind == "duckiebot":     # objects.py:272-281 uses self.pos
            d = np.linalg.norm(agent_pos - self.pos)
            return min(0, d - agent_safety_rad - self.safety_radius)
        d = np.linalg.norm(agent_pos - self.center)
        score = d - agent_safety_rad - self.safety_radius
        return min(0, score)


class OracleMap:
    """simulator.py:788-1038 (_interpret_map, _load_objects, interpret_object)."""

    def __init__(self, map_data: dict, mesh_extents: dict, transform_uses_width: bool = False):
        self.tile_size = ts = map_data["tile_size"]
        tiles = map_data["tiles"]
        self.grid_height = len(tiles)
        self.grid_width = len(tiles[0])
        self.grid: List[Optional[dict]] = [None] * (self.grid_width * self.grid_height)
        self.drivable_tiles = []
        directions = ["S", "E", "N", "W"]
        for j, row in enumerate(tiles):
            assert len(row) == self.grid_width
            for i, tile in enumerate(row):
                tile = tile.strip()
                if tile == "empty":
                    continue
                if "/" in tile:
                    kind, orient = tile.split("/")
                    kind = kind.strip(" ")
                    angle = directions.index(orient.strip(" "))
                elif "4" in tile:
                    kind, angle = "4way", directions.index("E")
                else:
                    kind, angle = tile, directions.index("E")
                drivable = kind in DRIVABLE_TILES
                t = {"coords": (i, j), "kind": kind, "angle": angle, "drivable": drivable}
                self.grid[j * self.grid_width + i] = t
                if drivable:
                    t["curves"] = get_curve(kind, angle, i, j, ts)
                    self.drivable_tiles.append(t)
        self.start_tile = None
        if "start_tile" in map_data:
            self.start_tile = self.get_tile(*map_data["start_tile"])
        self.start_pose = map_data.get("start_pose")

        # objects
        self.objects: List[OracleObj] = []
        centers, corners, norms, radii = [], [], [], []
        objs = map_data.get("objects") or []
        if isinstance(objs, dict):
            objs = list(objs.values())
        for desc in objs:
            kind = desc["kind"]
            if kind == "floor_tag":  # simulator.py:971-972
                continue
            # get_transform [R] + weird_from_cartesian simulator.py:936-943,1640-1652
            Hc = self.grid_width if transform_uses_width else self.grid_height
            px, pz = desc["pos"][0], desc["pos"][1]
            rot = np.deg2rad(desc.get("rotate", 0.0))
            cpx, cpy = px * ts, (Hc - pz) * ts
            angle = float(np.arctan2(np.sin(rot), np.cos(rot)))
            pos = np.array([cpx, 0, self.grid_height * ts - cpy])
            mn, mx = mesh_extents.get(kind) or mesh_extents["*"]
            mn = np.asarray(mn, dtype=np.float32).astype(np.float64)
            mx = np.asarray(mx, dtype=np.float32).astype(np.float64)
            if "height" in desc:
                scale = desc["height"] / mx[1]
            else:
                scale = desc.get("scale", 1.0)
            static = desc.get("static", True)
            oc = generate_corners(pos, mn, mx, angle, scale)
            o = OracleObj(
                kind=kind, pos=pos, angle=angle, scale=scale, static=static,
                optional=desc.get("optional", False), min_coords=mn, max_coords=mx,
                safety_radius=SAFETY_RAD_MULT * calculate_safety_radius(mn, mx, scale),
                obj_corners=oc, obj_norm=generate_norm(oc), y_rot=float(np.rad2deg(angle)),
            )
            if static and kind == "trafficlight":
                o.light_freq, o.light_pattern = 5, 0  # objects.py:446-451
            if not static and kind == "duckiebot":
                pass                  # DuckiebotObj(obj_desc, ..., WHEEL_DIST, ROBOT_WIDTH, ROBOT_LENGTH) simulator.py:1005-1008
            elif not static and kind == "checkerboard":     # CheckerboardObj simulator.py:1013-1014
                o.walk_distance = ts + 0.25
                o.start = np.copy(pos); o.reset_start = np.copy(pos); o.center = pos; o.steps = -20
            elif not static:
                assert kind == "duckie", "oracle: DuckieObj / DuckiebotObj / CheckerboardObj dynamics restated"
                o.walk_distance = ts  # simulator.py:1010
                o.heading = np.array([math.cos(angle), 0, -math.sin(angle)])  # collision.py:223
                o.start = np.copy(pos)
                o.center = pos
                o.wiggle = math.pi / desc.get("wiggle_div", 15)
            self.objects.append(o)
            if static and kind != "trafficlight":  # simulator.py:1027-1038
                centers.append(pos)
                corners.append(oc.T)
                norms.append(o.obj_norm)
                radii.append(o.safety_radius)
        self.collidable_centers = np.array(centers)
        self.collidable_corners = np.stack(corners, axis=0) if corners else np.zeros((0, 2, 4))
        self.collidable_norms = np.stack(norms, axis=0) if norms else np.zeros((0, 2, 2))
        self.collidable_safety_radii = np.array(radii)

    def get_tile(self, i, j):
        """simulator.py:1053-1063"""
        i, j = int(i), int(j)
        if i < 0 or i >= self.grid_width or j < 0 or j >= self.grid_height:
            return None
        return self.grid[j * self.grid_width + i]

    def get_grid_coords(self, abs_pos):
        """simulator.py:1134-1149"""
        x, _, z = abs_pos
        return int(math.floor(x / self.tile_size)), int(math.floor(z / self.tile_size))


# ---- dynamics: duckietown_world [R], PARITY UNPINNED -----------------------
class DynamicsDB18:
    """Restatement of duckietown-world-daffy's DB18 dynamics as used at
    simulator.py:745-755 (get_DB18_nominal / get_DB18_uncalibrated, delay
    0.15 s) and :2076-2088 (integrate, TSE2_from_state).  The package is not
    under /root/reference and is unpinned in setup.py:34 -- PARITY UNPINNED.

    Published model (pwm_dynamics.py / dynamics_delay.py / generic_kinematics.py):
      parameters u1=5 u2=u3=0 w1=4 w2=w3=0 u_ar=u_al=1.5 w_ar=w_al=15
      (uncalibrated: w_ar=15(1+trim), w_al=15(1-trim));
      V = clip([R, L], -1, 1);  acc = f_dynamic + B V;  explicit Euler on
      (u, w);  q <- q * exp(dt * se2(u', 0, w'));
      ApplyDelay: the command applied at time t is the latest one issued at
      or before t - 0.15 s, (0,0) before that => delay_steps = 5 at dt = 1/30
      (k - 4.5 is never an integer, so there is no rounding ambiguity).
    """

    def __init__(self, x, y, angle, trim=None, delay_steps=5):
        self.u1, self.u2, self.u3 = 5.0, 0.0, 0.0
        self.w1, self.w2, self.w3 = 4.0, 0.0, 0.0
        self.uar = self.ual = 1.5
        if trim is None:
            self.war = self.wal = 15.0
        else:
            self.war = 15.0 * (1.0 + trim)
            self.wal = 15.0 * (1.0 - trim)
        self.x, self.y = float(x), float(y)
        self.c, self.s = math.cos(angle), math.sin(angle)
        self.u = 0.0
        self.w = 0.0
        self.ring = [(0.0, 0.0)] * delay_steps
        self.head = 0

    def integrate(self, dt, left, right):
        if len(self.ring) > 0:
            l_use, r_use = self.ring[self.head]
            self.ring[self.head] = (float(left), float(right))
            self.head = (self.head + 1) % len(self.ring)
        else:
            l_use, r_use = float(left), float(right)
        R = min(max(r_use, -1.0), 1.0)
        L = min(max(l_use, -1.0), 1.0)
        u, w = self.u, self.w
        acc_u = (-self.u1 * u - self.u2 * w + self.u3 * w * w) + (self.uar * R + self.ual * L)
        acc_w = (-self.w1 * w - self.w2 * u - self.w3 * u * w) + (self.war * R + (-self.wal) * L)
        u = u + dt * acc_u
        w = w + dt * acc_w
        th = dt * w
        vx = dt * u
        if abs(th) < 1e-12:
            cd, sd, tx, ty = 1.0, th, vx, 0.0
        else:
            sd, cd = math.sin(th), math.cos(th)
            A = sd / th
            B = (1.0 - cd) / th
            tx = A * vx
            ty = B * vx
        c0, s0 = self.c, self.s
        self.x = (c0 * tx + (-s0) * ty) + self.x
        self.y = (s0 * tx + c0 * ty) + self.y
        self.c = c0 * cd + (-s0) * sd
        self.s = s0 * cd + c0 * sd
        self.u, self.w = u, w

    def angle(self):
        return math.atan2(self.s, self.c)


# ---- Randomizer: randomization/randomizer.py:8-16,36-91 --------------------
_DR_CONFIG = {
    "horz_mode": {"type": "int", "low": 0, "high": 4},
    "light_pos": {"type": "uniform", "low": [-150, 170, -150], "high": [150, 220, 150], "size": 3},
    "camera_noise": {"type": "uniform", "low": -0.005, "high": 0.005, "size": 3},
    "trim": {"type": "normal", "loc": 0, "scale": 0.02},
    "camera_height": {"type": "uniform", "low": 0.92, "high": 1.08},
    "camera_angle": {"type": "uniform", "low": 0.8, "high": 1.2},
    "camera_fov_y": {"type": "uniform", "low": 0.8, "high": 1.2},
}


def randomize(rng):
    out = {}
    for k in sorted(_DR_CONFIG):
        d = _DR_CONFIG[k]
        size = d.get("size", 1)
        if d["type"] == "int":
            out[k] = rng.integers(low=d["low"], high=d["high"], size=size)
        elif d["type"] == "uniform":
            out[k] = rng.uniform(low=d["low"], high=d["high"], size=size)
        else:
            out[k] = rng.normal(loc=d["loc"], scale=d["scale"], size=size)
    return out


class OracleSim:
    """Simulator (simulator.py:188-2053) + DuckietownEnv.step
    (envs/duckietown_env.py:36-72), without rendering (see raster.py)."""

    def __init__(self, map_data, mesh_extents, *, max_steps=1500, domain_rand=False,
                 frame_rate=30, frame_skip=1, robot_speed=DEFAULT_ROBOT_SPEED,
                 accept_start_angle_deg=60, seed=None, dynamics_rand=False, camera_rand=False,
                 user_tile_start=None, num_tris_distractors=12, color_ground=(0.15, 0.15, 0.15),
                 color_sky=BLUE_SKY, delay_steps=None, transform_uses_width=False,
                 gain=1.0, trim=0.0, radius=0.0318, k=27.0, limit=1.0, do_reset=True):
        if delay_steps is None:   # 0.15 s of simulated time (get_DB18_nominal(delay=0.15)): smallest k with k * dt >= 0.15
            delay_steps = int(math.ceil(0.15 * frame_rate - 1e-9))
        self.map = OracleMap(map_data, mesh_extents, transform_uses_width)
        self._map_args = (map_data, mesh_extents, transform_uses_width)
        self.max_steps = max_steps
        self.domain_rand = domain_rand
        self.delta_time = 1.0 / frame_rate
        self.frame_skip = frame_skip
        self.robot_speed = robot_speed
        self.accept_start_angle_deg = accept_start_angle_deg
        self.dynamics_rand = dynamics_rand
        self.camera_rand = camera_rand
        self.user_tile_start = user_tile_start
        self.num_tris_distractors = num_tris_distractors
        self.color_ground = color_ground
        self.color_sky = list(color_sky)
        self.delay_steps = delay_steps
        self.gain, self.trim, self.radius, self.k, self.limit = gain, trim, radius, k, limit
        self.seed(seed)
        self.last_action = np.array([0, 0])
        self.wheelVels = np.array([0, 0])
        if do_reset:
            self.reset()

    def seed(self, seed=None):
        """simulator.py:1043-1045; gym>=0.22 => Generator(PCG64)."""
        self.np_random = np.random.default_rng(seed)
        return [seed]

    # -- simulator.py:1065-1085
    def _perturb(self, val, scale=0.1):
        val = np.array(val)
        if not self.domain_rand:
            return val
        noise = self.np_random.uniform(low=1 - scale, high=1 + scale, size=val.shape)
        if val.size == 4:
            noise[3] = 1
        return val * noise

    # -- simulator.py:528-763
    def reset(self):
        m = self.map
        self.step_count = 0
        self.timestamp = 0.0
        self.speed = 0.0
        rs = self.randomization_settings = randomize(self.np_random)
        if self.domain_rand:
            hm = rs["horz_mode"]
            if hm == 0:
                self.horizon_color = self._perturb(self.color_sky)
            elif hm == 1:
                self.horizon_color = self._perturb(WALL_COLOR)
            elif hm == 2:
                self.horizon_color = self._perturb([0.15, 0.15, 0.15], 0.4)
            elif hm == 3:
                self.horizon_color = self._perturb([0.9, 0.9, 0.9], 0.4)
            self.light_pos = list(rs["light_pos"])
        else:
            self.horizon_color = np.array(self.color_sky)
            self.light_pos = [0.0, 3.0, 0.0, 1.0]
        self.light_ambient = self._perturb(np.array([0.50 * DIM, 0.50 * DIM, 0.50 * DIM, 1]), 0.3)
        self.light_diffuse = self._perturb(np.array([0.70 * DIM, 0.70 * DIM, 0.70 * DIM, 1]), 0.99)
        self.ground_color = self._perturb(np.array(self.color_ground), 0.3)
        self.wheel_dist = self._perturb(WHEEL_DIST)
        self.cam_height = CAMERA_FLOOR_DIST
        self.cam_angle = [CAMERA_ANGLE, 0, 0]
        self.cam_fov_y = CAMERA_FOV_Y
        if self.domain_rand or self.camera_rand:
            self.cam_height *= rs["camera_height"]
            self.cam_angle = [CAMERA_ANGLE * rs["camera_angle"], 0, 0]
            self.cam_fov_y *= rs["camera_fov_y"]
        self.cam_offset = np.array([0, 0, 0])
        # distractor triangles: RNG consumed, never visible (simulator.py:621-631)
        for _ in range(0, 3 * self.num_tris_distractors):
            self.np_random.uniform(low=[-20, -0.6, -20], high=[20, -0.3, 20], size=(3,))
            c = self.np_random.uniform(low=0, high=0.9)
            self._perturb([c, c, c], 0.1)
        for tile in m.grid:  # simulator.py:634-645 (None tiles would raise there: Q5)
            if tile is None:
                continue
            tile["color"] = self._perturb([1, 1, 1, 1], 0.2)
        # fresh object state per reset is NOT what the reference does (objects
        # persist across resets); mirrored: objects keep their dynamic state.
        for obj in m.objects:
            obj.color = self._perturb([1, 1, 1, 1], 0.3)
            if obj.optional and self.domain_rand:
                obj.visible = self.np_random.integers(0, 2) == 0
            else:
                obj.visible = True
        if self.user_tile_start:
            tile = m.get_tile(*self.user_tile_start)
            if tile is None:
                raise Exception("The tile specified does not exist.")
        elif m.start_tile is not None:
            tile = m.start_tile
        else:
            tile_idx = self.np_random.integers(0, len(m.drivable_tiles))
            tile = m.drivable_tiles[tile_idx]
        ts = m.tile_size
        self.spawn_attempts = 0
        if m.start_pose is not None:
            i, j = tile["coords"]
            propose_pos = np.array([i * ts + m.start_pose[0][0], 0, j * ts + m.start_pose[0][2]])
            propose_angle = m.start_pose[1]
        else:
            for _ in range(MAX_SPAWN_ATTEMPTS):
                self.spawn_attempts += 1
                i, j = tile["coords"]
                x = self.np_random.uniform(i, i + 1) * ts
                z = self.np_random.uniform(j, j + 1) * ts
                propose_pos = np.array([x, 0, z])
                propose_angle = self.np_random.uniform(0, 2 * math.pi)
                if self._inconvenient_spawn(propose_pos):
                    continue
                if not self._valid_pose(propose_pos, propose_angle, safety_factor=1.3):
                    continue
                try:
                    lp = self.get_lane_pos2(propose_pos, propose_angle)
                except NotInLane:
                    continue
                M = self.accept_start_angle_deg
                if not (-M < lp[2] < +M):
                    continue
                break
            else:
                propose_pos = np.array([1, 0, 1])
                propose_angle = 1
        self.cur_pos = propose_pos
        self.cur_angle = propose_angle
        trim = (0 + rs["trim"][0]) if self.dynamics_rand else None
        # cartesian_from_weird simulator.py:1629-1638
        self.state = DynamicsDB18(self.cur_pos[0], m.grid_height * ts - self.cur_pos[2],
                                  self.cur_angle, trim=trim, delay_steps=self.delay_steps)

    # -- simulator.py:1461-1471
    def _inconvenient_spawn(self, pos):
        results = [
            np.linalg.norm(x.pos - pos) < max(x.max_coords) * 0.5 * x.scale + MIN_SPAWN_OBJ_DIST
            for x in self.map.objects if x.visible
        ]
        return bool(np.any(results))

    # -- simulator.py:1411-1428
    def _drivable_pos(self, pos):
        tile = self.map.get_tile(*self.map.get_grid_coords(pos))
        return tile is not None and tile["drivable"]

    # -- simulator.py:1473-1492
    def _collision(self, agent_corners):
        m = self.map
        agent_norm = generate_norm(agent_corners)
        if len(m.collidable_corners) > 0:
            if intersects(agent_corners, m.collidable_corners, agent_norm, m.collidable_norms):
                return True
        for obj in m.objects:
            if obj.check_collision(agent_corners, agent_norm):
                return True
        return False

    # -- simulator.py:1494-1534 (incl. the double centre shift)
    def _valid_pose(self, pos, angle, safety_factor=1.0):
        pos = actual_center(pos, angle)
        f_vec = get_dir_vec(angle)
        r_vec = get_right_vec(angle)
        l_pos = pos - (safety_factor * 0.5 * ROBOT_WIDTH) * r_vec
        r_pos = pos + (safety_factor * 0.5 * ROBOT_WIDTH) * r_vec
        f_pos = pos + (safety_factor * 0.5 * ROBOT_LENGTH) * f_vec
        all_drivable = (self._drivable_pos(pos) and self._drivable_pos(l_pos)
                        and self._drivable_pos(r_pos) and self._drivable_pos(f_pos))
        agent_corners = get_agent_corners(pos, angle)
        no_collision = not self._collision(agent_corners)
        return bool(no_collision and all_drivable)

    # -- simulator.py:1337-1369
    def closest_curve_point(self, pos, angle):
        m = self.map
        i, j = m.get_grid_coords(pos)
        tile = m.get_tile(i, j)
        if tile is None or not tile["drivable"]:
            return None, None
        curves = tile["curves"]
        curve_headings = curves[:, -1, :] - curves[:, 0, :]
        curve_headings = curve_headings / np.linalg.norm(curve_headings).reshape(1, -1)
        dot_prods = np.dot(curve_headings, get_dir_vec(angle))
        cps = curves[np.argmax(dot_prods)]
        t = bezier_closest(cps, pos)
        self._last_curve = (int(np.argmax(dot_prods)), t)
        return bezier_point(cps, t), bezier_tangent(cps, t)

    # -- simulator.py:1371-1409
    def get_lane_pos2(self, pos, angle):
        point, tangent = self.closest_curve_point(pos, angle)
        if point is None:
            raise NotInLane(f"Point not in lane: {pos}")
        dirVec = get_dir_vec(angle)
        dotDir = np.clip(np.dot(dirVec, tangent), -1.0, +1.0)
        posVec = pos - point
        rightVec = np.cross(tangent, np.array([0, 1, 0]))
        signedDist = np.dot(posVec, rightVec)
        angle_rad = math.acos(dotDir)
        if np.dot(dirVec, rightVec) < 0:
            angle_rad *= -1
        return (float(signedDist), float(dotDir), float(np.rad2deg(angle_rad)), float(angle_rad))

    # -- simulator.py:1430-1459 ; collision.py:189-211
    def proximity_penalty2(self, pos, angle):
        m = self.map
        pos = actual_center(pos, angle)
        if len(m.collidable_centers) == 0:
            static_dist = 0
        else:
            d = np.linalg.norm(m.collidable_centers - pos, axis=1)
            r1, r2 = AGENT_SAFETY_RAD, m.collidable_safety_radii
            intersect = np.logical_and(np.less_equal(np.power(r1 - r2, 2), np.power(d, 2)),
                                       np.less_equal(np.power(d, 2), np.power(r1 + r2, 2)))
            enveloped = np.less(d, abs(r1 - r2))
            if not (np.any(intersect) or np.any(enveloped)):
                static_dist = 0.0
            else:
                scores = d - r1 - r2
                static_dist = np.sum(scores[np.where(scores < 0)])
        total = static_dist
        for obj in m.objects:
            total += obj.proximity(pos, AGENT_SAFETY_RAD)
        return float(total)

    # -- simulator.py:1654-1667
    def compute_reward(self, pos, angle, speed):
        col_penalty = self.proximity_penalty2(pos, angle)
        try:
            lp = self.get_lane_pos2(pos, angle)
        except NotInLane:
            return 40 * col_penalty
        return +1.0 * speed * lp[1] + -10 * np.abs(lp[0]) + +40 * col_penalty

    # -- simulator.py:1551-1584, 2076-2088
    def update_physics(self, action):
        m = self.map
        self.wheelVels = action * self.robot_speed * 1
        prev_pos = self.cur_pos
        self.state.integrate(self.delta_time, action[0], action[1])
        ts = m.tile_size
        # weird_from_cartesian simulator.py:1640-1652
        self.cur_pos = np.asarray([self.state.x, 0, m.grid_height * ts - self.state.y])
        self.cur_angle = self.state.angle()
        self.step_count += 1
        self.timestamp += self.delta_time
        self.last_action = action
        self.speed = np.linalg.norm(self.cur_pos - prev_pos) / self.delta_time
        for obj in m.objects:
            if obj.kind == "duckiebot":
                if not obj.static:
                    obj.step_duckiebot(self.delta_time, self.closest_curve_point)
            else:
                obj.step(self.delta_time)

    # -- simulator.py:1685-1705
    def _compute_done_reward(self):
        if not self._valid_pose(self.cur_pos, self.cur_angle):
            return True, float(REWARD_INVALID_POSE), DONE_INVALID_POSE
        if self.step_count >= self.max_steps:
            return True, 0.0, DONE_MAX_STEPS
        return False, float(self.compute_reward(self.cur_pos, self.cur_angle, self.robot_speed)), DONE_IN_PROGRESS

    # -- simulator.py:1669-1683 (without render_obs)
    def step(self, action):
        action = np.clip(action, -1, 1)
        action = np.array(action, dtype=np.float64)
        for _ in range(self.frame_skip):
            self.update_physics(action)
        done, reward, code = self._compute_done_reward()
        return reward, done, code

    # -- envs/duckietown_env.py:36-72
    def wheels_from_vel_steer(self, action):
        vel, angle = float(action[0]), float(action[1])
        baseline = float(self.wheel_dist)
        k_r_inv = (self.gain + self.trim) / self.k
        k_l_inv = (self.gain - self.trim) / self.k
        omega_r = (vel + 0.5 * angle * baseline) / self.radius
        omega_l = (vel - 0.5 * angle * baseline) / self.radius
        u_r = omega_r * k_r_inv
        u_l = omega_l * k_l_inv
        u_r_limited = max(min(u_r, self.limit), -self.limit)
        u_l_limited = max(min(u_l, self.limit), -self.limit)
        return np.array([u_l_limited, u_r_limited])

    def step_vel_steer(self, action):
        return self.step(self.wheels_from_vel_steer(action))

    # convenience for tests / parity: everything the device reports per step
    def info(self):
        pos, ang = self.cur_pos, self.cur_angle
        i, j = self.map.get_grid_coords(pos)
        try:
            lp = self.get_lane_pos2(pos, ang)
            in_lane = True
        except NotInLane:
            lp, in_lane = (0.0, 0.0, 0.0, 0.0), False
        return dict(pos=np.array(pos, dtype=np.float64), angle=float(ang), tile=(i, j), in_lane=in_lane,
                    lane=lp, prox=self.proximity_penalty2(pos, ang), speed=float(self.speed),
                    step_count=self.step_count, timestamp=self.timestamp)

    def set_pose(self, pos, angle):
        """Oracle-fed state (dtsim_reset(states) parity mode): pose only."""
        m = self.map
        self.cur_pos = np.array(pos, dtype=np.float64)
        self.cur_angle = float(angle)
        self.step_count = 0
        self.timestamp = 0.0
        self.speed = 0.0
        self.state = DynamicsDB18(self.cur_pos[0], m.grid_height * m.tile_size - self.cur_pos[2],
                                  self.cur_angle, trim=None, delay_steps=self.delay_steps)
