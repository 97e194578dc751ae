"""Host-side, one-time map preparation: MapFormat1 dict -> flat tables for dtsim_set_maps.

This is the load-time half of Simulator._interpret_map / _load_objects / interpret_object
(simulator.py:788-1038): it runs once per map on the host, exactly as in the reference,
and produces the tables the HIP kernels consume.  The per-step geometry (tile lookup,
lane pose, SAT, proximity) is NOT here -- it lives only in csrc/physics.hip.

Numerics that feed bit-exact flags use the same numpy expressions as the reference so
that the tables are identical on the same machine:
  curves   np.matmul(template * tile_size, gen_rot_matrix([0,1,0], angle*pi/2)) + centre
           (simulator.py:1304-1335, graphics.py:268-283)
  heads    (P3 - P0) / np.linalg.norm(all chords of the tile)   (simulator.py:1355-1356)
  corners  rotate_point about the object position              (collision.py:64-79)
  norms    eigenvectors of the corner covariance (np.cov + np.linalg.eig; collision.py:99-106)
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

from . import _ffi
from .assets import MeshData, get_mesh

SAFETY_RAD_MULT = 1.8        # simulator.py:150
MIN_SPAWN_OBJ_DIST = 0.25    # simulator.py:156
ORIENTATIONS = ["S", "E", "N", "W"]  # simulator.py:823
DRIVABLE = ("straight", "curve_left", "curve_right", "3way_left", "3way_right", "4way")


class InvalidMapException(Exception):
    """exceptions.py:10"""


# Bezier templates in tile units, (x, z) only -- the data of simulator.py:1164-1299.
_A = (-0.20, -0.50)
_T = {
    "straight": [[_A, (-0.20, -0.25), (-0.20, 0.25), (-0.20, 0.50)],
                 [(0.20, 0.50), (0.20, 0.25), (0.20, -0.25), (0.20, -0.50)]],
    "curve_left": [[_A, (-0.20, 0.00), (0.00, 0.20), (0.50, 0.20)],
                   [(0.50, -0.20), (0.30, -0.20), (0.20, -0.30), (0.20, -0.50)]],
    "curve_right": [[_A, (-0.20, -0.20), (-0.30, -0.20), (-0.50, -0.20)],
                    [(-0.50, 0.20), (-0.30, 0.20), (0.30, 0.00), (0.20, -0.50)]],
    "3way": [[_A, (-0.20, -0.25), (-0.20, 0.25), (-0.20, 0.50)],
             [_A, (-0.20, 0.00), (0.00, 0.20), (0.50, 0.20)],
             [(0.20, 0.50), (0.20, 0.25), (0.20, -0.25), (0.20, -0.50)],
             [(0.50, -0.20), (0.30, -0.20), (0.20, -0.20), (0.20, -0.50)],
             [(0.20, 0.50), (0.20, 0.20), (0.30, 0.20), (0.50, 0.20)],
             [(0.50, -0.20), (0.30, -0.20), (-0.20, 0.00), (-0.20, 0.50)]],
    "4way": [[_A, (-0.20, 0.00), (0.00, 0.20), (0.50, 0.20)],
             [_A, (-0.20, -0.25), (-0.20, 0.25), (-0.20, 0.50)],
             [_A, (-0.20, -0.20), (-0.30, -0.20), (-0.50, -0.20)]],
}


def _template(kind: str) -> np.ndarray:
    key = "3way" if kind.startswith("3way") else ("4way" if kind.startswith("4way") else kind)
    if key not in _T:
        raise InvalidMapException(f"Cannot get bezier for kind {kind!r}")
    xz = np.array(_T[key], dtype=np.float64)                      # [C,4,2]
    pts = np.zeros(xz.shape[:2] + (3,))
    pts[..., 0], pts[..., 2] = xz[..., 0], xz[..., 1]
    return pts


def _rot_y(angle: float) -> np.ndarray:
    """graphics.py:268-283 gen_rot_matrix for axis (0,1,0), same operation order."""
    axis = np.array([0, 1, 0]) / math.sqrt(1.0)
    a = math.cos(angle / 2.0)
    b, c, d = -axis * math.sin(angle / 2.0)
    return np.array([
        [a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
        [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
        [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c],
    ])


def tile_curves(kind: str, angle: int, i: int, j: int, ts: float) -> np.ndarray:
    """[C,4,3] control points in the world frame (simulator.py:1151-1335)."""
    pts = _template(kind) * ts
    centre = np.array([(i + 0.5) * ts, 0, (j + 0.5) * ts])
    if kind.startswith("4way"):
        out = []
        for rot in np.arange(0, 4):
            p = np.matmul(pts, _rot_y(rot * math.pi / 2))
            p += centre
            out.append(p)
        return np.reshape(np.array(out), (12, 4, 3))
    p = np.matmul(pts, _rot_y(angle * math.pi / 2))
    p += centre
    return p


def _rotate_point(px, py, cx, cy, theta):
    dx, dy = px - cx, py - cy
    return (cx + (dx * math.cos(theta) + dy * math.sin(theta)),
            cy + (dy * math.cos(theta) - dx * math.sin(theta)))


@dataclass
class ObjectTable:
    kind: str
    mesh_kind: str
    pos: np.ndarray
    angle: float
    scale: float
    static: bool
    optional: bool
    collidable: bool
    corners: np.ndarray          # [4,2]
    norm: np.ndarray             # [2,2]
    safety_radius: float
    spawn_clear: float
    min_coords: np.ndarray
    max_coords: np.ndarray
    walk_distance: float = 0.0
    vel: float = 0.02            # objects.py:352-353 non-DR defaults
    wait_time: float = 8.0
    wiggle: float = math.pi / 15
    dyn_slot: int = -1
    dyn_kind: int = 0            # 0 static, 1 DuckieObj, 2 DuckiebotObj
    light_freq: int = 0          # TrafficLightObj (objects.py:434-453): seconds between pattern switches, 0 = not a light
    light_pattern: int = 0
    light_tris: int = 0          # triangles of the mesh's first material chunk (mesh.textures[0])


@dataclass
class MapTables:
    name: str
    grid_w: int
    grid_h: int
    tile_size: float
    tile_kind_names: List[Optional[str]]
    tile_kind: np.ndarray
    tile_angle: np.ndarray
    tile_tex: np.ndarray
    tile_curve_off: np.ndarray
    tile_curve_cnt: np.ndarray
    curves: np.ndarray           # [n_curves,4,2]
    curves3: np.ndarray          # [n_curves,4,3] (for the facade's tile["curves"])
    curve_heads: np.ndarray      # [n_curves,2]
    objects: List[ObjectTable]
    drivable_tiles: List[tuple]  # (i, j) in the reference's append order (row-major)
    start_tile: Optional[tuple] = None
    start_pose: Optional[list] = None
    texture_kinds: List[str] = field(default_factory=list)
    _keep: list = field(default_factory=list, repr=False)

    @property
    def n_dynamic(self):
        return sum(1 for o in self.objects if not o.static)

    def to_ffi(self, mesh_ids: Dict[str, int], light_tex=(-1, -1)) -> _ffi.Map:
        m = _ffi.Map()
        m.grid_w, m.grid_h, m.tile_size = self.grid_w, self.grid_h, float(self.tile_size)

        def ptr(a, ct):
            self._keep.append(a)
            return a.ctypes.data_as(C.POINTER(ct))

        m.tile_kind = ptr(self.tile_kind, C.c_uint8)
        m.tile_angle = ptr(self.tile_angle, C.c_uint8)
        m.tile_tex = ptr(self.tile_tex, C.c_int16)
        m.tile_curve_off = ptr(self.tile_curve_off, C.c_int16)
        m.tile_curve_cnt = ptr(self.tile_curve_cnt, C.c_uint8)
        m.n_curves = self.curves.shape[0]
        m.curves = ptr(np.ascontiguousarray(self.curves), C.c_double)
        m.curve_heads = ptr(np.ascontiguousarray(self.curve_heads), C.c_double)
        arr = (_ffi.Object * max(len(self.objects), 1))()
        for k, o in enumerate(self.objects):
            f = arr[k]
            f.mesh_id = mesh_ids.get(o.mesh_kind, -1)
            f.dynamic = 0 if o.static else o.dyn_kind
            f.collidable = 1 if o.collidable else 0
            f.optional = 1 if o.optional else 0
            f.pos[:] = [float(v) for v in o.pos]
            f.angle, f.scale = float(o.angle), float(o.scale)
            f.corners[:] = [float(v) for v in o.corners.reshape(-1)]
            f.norm[:] = [float(v) for v in o.norm.reshape(-1)]
            f.safety_radius, f.spawn_clear = float(o.safety_radius), float(o.spawn_clear)
            f.walk_distance, f.vel, f.wait_time, f.wiggle = (float(o.walk_distance), float(o.vel),
                                                            float(o.wait_time), float(o.wiggle))
            lit = o.light_freq > 0 and light_tex[0] >= 0 and light_tex[1] >= 0 and f.mesh_id >= 0
            f.light_freq, f.light_pattern = int(o.light_freq), int(o.light_pattern)
            f.light_tex[0], f.light_tex[1] = (int(light_tex[0]), int(light_tex[1])) if lit else (-1, -1)
            f.light_tris = int(o.light_tris) if lit else 0
        self._keep.append(arr)
        m.n_objects = len(self.objects)
        m.objects = C.cast(arr, C.POINTER(_ffi.Object))
        return m


def interpret_map(map_data: dict, name: str = "map", meshes: Optional[Dict[str, MeshData]] = None,
                  transform_uses_width: bool = False, texture_ids: Optional[Dict[str, int]] = None,
                  library=None) -> MapTables:
    """MapFormat1 dict -> MapTables.  `texture_ids` maps tile kind -> texture index
    (filled in by the caller after it decided which textures to upload)."""
    if "tile_size" not in map_data:
        raise InvalidMapException("Must now include explicit tile_size in the map data.")
    ts = map_data["tile_size"]
    rows = map_data["tiles"]
    if not rows or not rows[0]:
        raise InvalidMapException("empty tile grid")
    H, W = len(rows), len(rows[0])
    n = W * H
    kind_names: List[Optional[str]] = [None] * n
    tile_kind = np.zeros(n, np.uint8)
    tile_angle = np.zeros(n, np.uint8)
    tile_tex = np.full(n, -1, np.int16)
    coff = np.full(n, -1, np.int16)
    ccnt = np.zeros(n, np.uint8)
    curves3, heads, drivable = [], [], []
    tex_kinds: List[str] = []
    for j, row in enumerate(rows):
        if len(row) != W:
            raise InvalidMapException("each row of tiles must have the same length")
        for i, cell in enumerate(row):
            cell = cell.strip()
            if cell == "empty":
                continue
            if "/" in cell:                       # simulator.py:826-830
                kind, orient = (s.strip(" ") for s in cell.split("/"))
                ang = ORIENTATIONS.index(orient)
            elif "4" in cell:
                kind, ang = "4way", ORIENTATIONS.index("E")
            else:
                kind, ang = cell, ORIENTATIONS.index("E")
            idx = j * W + i
            kind_names[idx] = kind
            tile_kind[idx] = _ffi.TILE_KINDS.get(kind, _ffi.TILE_OTHER)
            tile_angle[idx] = ang
            if kind not in tex_kinds:
                tex_kinds.append(kind)
            if kind in DRIVABLE:
                c3 = tile_curves(kind, ang, i, j, ts)
                coff[idx], ccnt[idx] = len(curves3), c3.shape[0]
                ch = c3[:, -1, :] - c3[:, 0, :]                    # simulator.py:1355
                ch = ch / np.linalg.norm(ch).reshape(1, -1)        # simulator.py:1356
                curves3.extend(c3)
                heads.extend(ch[:, [0, 2]])
                drivable.append((i, j))
    if texture_ids is not None:
        for idx, k in enumerate(kind_names):
            if k is not None:
                tile_tex[idx] = texture_ids.get(k, -1)
    curves3 = np.array(curves3, dtype=np.float64).reshape(-1, 4, 3)
    heads = np.array(heads, dtype=np.float64).reshape(-1, 2)

    objs: List[ObjectTable] = []
    raw = map_data.get("objects") or []
    if isinstance(raw, dict):
        raw = list(raw.values())
    n_dyn = 0
    for desc in raw:
        kind = desc["kind"]
        if kind == "floor_tag":                   # simulator.py:971-972
            continue
        if library is not None:                   # real assets: one mesh per object kind / variant
            mesh_kind, mesh = library.object_mesh(desc)
            if meshes is not None:
                meshes.setdefault(mesh_kind, mesh)
        else:
            mesh_kind = "duckie" if kind == "duckie" else "*"
            mesh = (meshes or {}).get(mesh_kind) or get_mesh(kind)
        # get_transform [R] (README.md:239) + weird_from_cartesian (simulator.py:1640-1652)
        Hc = W if transform_uses_width else H
        px, pz = desc["pos"][0], desc["pos"][1]
        rot = np.deg2rad(desc.get("rotate", 0.0))
        cpx, cpy = px * ts, (Hc - pz) * ts
        angle = float(np.arctan2(np.sin(rot), np.cos(rot)))
        pos = np.array([cpx, 0, H * ts - cpy])
        # extents: float32 values, float64 arithmetic (numpy<=1.20 promotion, setup.py:28)
        mn = np.asarray(mesh.min_coords, np.float32).astype(np.float64)
        mx = np.asarray(mesh.max_coords, np.float32).astype(np.float64)
        if "height" in desc and "scale" in desc:
            raise InvalidMapException("cannot specify both height and scale")
        scale = desc["height"] / mx[1] if "height" in desc else desc.get("scale", 1.0)
        static = desc.get("static", True)
        x0, x1, z0, z1 = mn[0] * scale + pos[0], mx[0] * scale + pos[0], mn[2] * scale + pos[2], mx[2] * scale + pos[2]
        corners = np.array([_rotate_point(x, z, pos[0], pos[2], angle)
                            for x, z in ((x0, z0), (x1, z0), (x1, z1), (x0, z1))])
        ca = np.cov(corners, y=None, rowvar=False, bias=True)       # collision.py:104-106
        _, vect = np.linalg.eig(ca)
        ex, _, ez = np.max([abs(mn), abs(mx)], axis=0)              # collision.py:219-220
        o = ObjectTable(
            kind=kind, mesh_kind=mesh_kind, pos=pos, angle=angle, scale=float(scale), static=static,
            optional=desc.get("optional", False), collidable=bool(static and kind != "trafficlight"),
            corners=corners, norm=vect.T, safety_radius=float(SAFETY_RAD_MULT * (np.linalg.norm([ex, ez]) * scale)),
            spawn_clear=float(max(mx) * 0.5 * scale + MIN_SPAWN_OBJ_DIST), min_coords=mn, max_coords=mx)
        if static and kind == "trafficlight":      # TrafficLightObj, non-DR values (objects.py:446-451)
            o.light_freq, o.light_pattern = 5, 0
            o.light_tris = int(getattr(mesh, "chunk_sizes", [0])[0]) if getattr(mesh, "texture_files", None) is not None else 0
        if not static:
            if kind == "duckie":
                o.dyn_kind = 1
                o.walk_distance = ts              # DuckieObj(..., walk_distance=road_tile_size) simulator.py:1010
            elif kind == "duckiebot":
                o.dyn_kind = 2                    # DuckiebotObj, non-DR defaults (objects.py:208-216)
                o.walk_distance, o.vel, o.wait_time, o.wiggle = 0.3, 0.1, 2.0, 0.0   # follow_dist, velocity, gain, trim
            elif kind == "checkerboard":
                o.dyn_kind = 3                    # CheckerboardObj (objects.py:479-505): scripted motion, step counter starts at -20
                o.walk_distance, o.vel, o.wait_time, o.wiggle = ts + 0.25, -20.0, 0.0, 0.0
            else:
                raise InvalidMapException("Object kind unknown.")      # simulator.py:1016-1018
            o.dyn_slot = n_dyn
            n_dyn += 1
        objs.append(o)
    st = map_data.get("start_tile")
    return MapTables(
        name=name, grid_w=W, grid_h=H, tile_size=float(ts), tile_kind_names=kind_names, tile_kind=tile_kind,
        tile_angle=tile_angle, tile_tex=tile_tex, tile_curve_off=coff, tile_curve_cnt=ccnt,
        curves=np.ascontiguousarray(curves3[:, :, [0, 2]]), curves3=curves3, curve_heads=heads, objects=objs,
        drivable_tiles=drivable, start_tile=tuple(st) if st is not None else None,
        start_pose=map_data.get("start_pose"), texture_kinds=tex_kinds)
