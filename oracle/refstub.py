"""Stub loader: import the *reference's own* numpy code in this container.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Works only where
/root/reference exists (this build container); the GPU box never has it, so
nothing under tests/ -m gpu, smoke() or bench.py may call this module.  It is
used by oracle/make_golden.py (to generate tests/golden/*.npz) and by the
`not gpu` tests that pin oracle/sim.py against the reference's code.

The reference imports gym, pyglet, cv2, duckietown_world, geometry,
carnivalmirror, zuper_commons, zmq -- all absent here.  They are replaced by
MagicMock modules, except for the handful of pure functions the hot path
actually calls:

  geometry.SE2_from_translation_angle / translation_angle_from_SE2 /
  se2_from_linear_angular            (PyGeometry closed forms)
  duckietown_world...map_loading.get_transform (README.md:239 semantics)
  simulator.get_mesh / get_duckiebot_mesh -> stand-in mesh extents
"""
from __future__ import annotations

import importlib
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np

REFERENCE_ROOT = "/root/reference"
REFERENCE_SRC = os.path.join(REFERENCE_ROOT, "src")


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_SRC, "gym_duckietown"))


class _Mesh:
    """Stand-in for objmesh.ObjMesh: only the extents feed the hot path
    (collision.py:214-220, objects.py:50-63, simulator.py:977)."""

    def __init__(self, min_coords, max_coords):
        # ObjMesh keeps float32 extents (objmesh.py:181,230-232).  The reference
        # pins numpy<=1.20 (setup.py:28) where float32-scalar (op) python-float
        # promotes to float64; under this container's numpy 2 (NEP 50) the same
        # expressions would stay float32.  Supplying the float32-rounded values
        # as float64 reproduces the pinned-numpy arithmetic on both.
        self.min_coords = np.asarray(min_coords, dtype=np.float32).astype(np.float64)
        self.max_coords = np.asarray(max_coords, dtype=np.float32).astype(np.float64)

    def render(self, *a, **k):  # pragma: no cover
        pass


class _SE2Transform:
    def __init__(self, p, theta):
        self.p = np.asarray(p, dtype=np.float64)
        self.theta = float(theta)

    def as_SE2(self):
        return _SE2_from_translation_angle(self.p, self.theta)


def _SE2_from_translation_angle(t, theta):
    c, s = np.cos(theta), np.sin(theta)
    return np.array([[c, -s, t[0]], [s, c, t[1]], [0.0, 0.0, 1.0]])


def _translation_angle_from_SE2(q):
    return np.array([q[0, 2], q[1, 2]]), float(np.arctan2(q[1, 0], q[0, 0]))


def _se2_from_linear_angular(linear, angular):
    return np.array([[0.0, -angular, linear[0]], [angular, 0.0, linear[1]], [0.0, 0.0, 0.0]])


_MODULES = [
    "pyglet", "pyglet.gl", "pyglet.image", "pyglet.window", "pyglet.graphics", "pyglet.text",
    "cv2", "gym", "gym.spaces", "gym.utils", "gym.utils.seeding", "gym.envs", "gym.envs.registration",
    "duckietown_world", "duckietown_world.resources", "duckietown_world.gltf",
    "duckietown_world.gltf.export", "duckietown_world.world_duckietown",
    "duckietown_world.world_duckietown.map_loading",
    "zuper_commons", "zuper_commons.logs", "zuper_commons.types",
    "geometry", "carnivalmirror", "zmq", "PIL", "PIL.Image",
]

_loaded = None


def load(mesh_extents=None, transform_uses_width: bool = False):
    """Return a namespace with the reference modules
    (simulator, collision, graphics, objects, distortion).

    mesh_extents: dict kind -> (min_coords, max_coords); default = stand-in
    duckie extents shared with the fixtures.
    """
    global _loaded
    if not available():
        raise RuntimeError("reference tree not present: refstub cannot load")
    if _loaded is not None:
        if mesh_extents is not None:
            _loaded._mesh_extents.clear()
            _loaded._mesh_extents.update(mesh_extents)
        return _loaded

    saved = {}
    for name in _MODULES:
        saved[name] = sys.modules.get(name)
        m = MagicMock(name=name)
        m.__path__ = []  # looks like a package
        sys.modules[name] = m

    class _Env:  # gym.Env
        pass

    sys.modules["gym"].Env = _Env

    class _Wrapper:  # gym.Wrapper family: just enough for the reference's wrappers.py to be instantiated
        def __init__(self, env=None):
            self.env = env
            self.observation_space = getattr(env, "observation_space", None)
            self.action_space = getattr(env, "action_space", None)

        @property
        def unwrapped(self):
            return getattr(self.env, "unwrapped", self.env)

    for _n in ("Wrapper", "ActionWrapper", "ObservationWrapper", "RewardWrapper"):
        setattr(sys.modules["gym"], _n, type(_n, (_Wrapper,), {}))

    class _ZException(Exception):
        def __init__(self, msg="", **kw):
            super().__init__(msg)

    sys.modules["zuper_commons.types"].ZException = _ZException
    sys.modules["duckietown_world.resources"].list_maps2 = lambda: {}

    class _MF1C:
        KIND_DUCKIEBOT = "duckiebot"
        KIND_DUCKIE = "duckie"
        KIND_TRAFFICLIGHT = "trafficlight"
        KIND_CHECKERBOARD = "checkerboard"
        ObjectKind = str

    sys.modules["duckietown_world"].MapFormat1Constants = _MF1C
    geo = sys.modules["geometry"]
    geo.SE2_from_translation_angle = _SE2_from_translation_angle
    geo.translation_angle_from_SE2 = _translation_angle_from_SE2
    geo.se2_from_linear_angular = _se2_from_linear_angular

    state = types.SimpleNamespace(H=None)

    def get_transform(desc, W, tile_size):
        # README.md:239: pos is in tile units, [col, row]; rotate in degrees.
        # The simulator passes grid_width as W (simulator.py:936-938); the
        # README semantics need the grid *height*: with transform_uses_width
        # False we use the value captured from the Simulator instance.
        H = W if (transform_uses_width or state.H is None) else state.H
        pos = desc["pos"]
        rot = np.deg2rad(desc.get("rotate", 0.0))
        return _SE2Transform([pos[0] * tile_size, (H - pos[1]) * tile_size], rot)

    sys.modules["duckietown_world.world_duckietown.map_loading"].get_transform = get_transform

    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    for k in [k for k in sys.modules if k == "gym_duckietown" or k.startswith("gym_duckietown.")]:
        del sys.modules[k]
    ns = types.SimpleNamespace()
    try:
        ns.simulator = importlib.import_module("gym_duckietown.simulator")
        ns.collision = importlib.import_module("gym_duckietown.collision")
        ns.graphics = importlib.import_module("gym_duckietown.graphics")
        ns.objects = importlib.import_module("gym_duckietown.objects")
        ns.distortion = importlib.import_module("gym_duckietown.distortion")
        ns.objmesh = importlib.import_module("gym_duckietown.objmesh")
        ns.randomizer = importlib.import_module("gym_duckietown.randomization.randomizer")
        ns.wrappers = importlib.import_module("gym_duckietown.wrappers")
        ns.duckietown_env = importlib.import_module("gym_duckietown.envs.duckietown_env")
    finally:
        sys.path.remove(REFERENCE_SRC)
        # The reference package must not shadow the product's drop-in package
        # of the same name for the rest of the process.
        ns._ref_modules = {k: v for k, v in sys.modules.items()
                           if k == "gym_duckietown" or k.startswith("gym_duckietown.")}
        for k in ns._ref_modules:
            del sys.modules[k]
        for name in _MODULES:
            if saved[name] is None:
                sys.modules.pop(name, None)
            else:
                sys.modules[name] = saved[name]

    ns._mesh_extents = dict(mesh_extents or {})
    ns._state = state

    def get_mesh(kind, **kw):
        ext = ns._mesh_extents.get(kind) or ns._mesh_extents.get("*")
        if ext is None:
            ext = ((-0.5, 0.0, -0.35), (0.5, 1.0, 0.35))  # SURVEY App. A stand-in
        return _Mesh(*ext)

    ns.simulator.get_mesh = get_mesh
    ns.simulator.get_duckiebot_mesh = lambda color: get_mesh("duckiebot")
    ns.simulator.logger = MagicMock()
    _loaded = ns
    return ns


def ref_objmesh(obj_path: str, mesh_name: str, resolve, change_materials=None, segment=False):
    """Run the reference's own ObjMesh parser (objmesh.py:55-358) on `obj_path`.  pyglet's vertex
    lists are captured instead of created, `get_resource_path` is `resolve` (basename -> path or
    None => KeyError, like duckietown_world), textures are recorded by path.  Returns
    dict(verts, uvs, normals, colors [T,3,*] in draw order, chunk_sizes, textures, min_coords, max_coords)."""
    ns = load()
    om = ns.objmesh
    chunks = []

    def vertex_list(n, *attrs):
        chunks.append({a[0]: np.array(a[1], dtype=np.float32) for a in attrs})
        return len(chunks) - 1

    def get_resource_path(bn):
        p = resolve(bn)
        if p is None:
            raise KeyError(bn)
        return p

    om.pyglet.graphics.vertex_list = vertex_list
    om.get_resource_path = get_resource_path
    # segment=True (objmesh.py:255-292): record what every chunk asks load_texture for -- (path, segment flag,
    # gen_segmentation_color(mesh_name)) -- instead of the path alone
    om.load_texture = (lambda path, **kw: (path, kw.get("segment"), kw.get("segment_into_color"))) if segment else (lambda path, **kw: path)
    om.logger = MagicMock()
    mesh = om.ObjMesh(obj_path, mesh_name, bool(segment), change_materials)
    cat = lambda key, w: np.concatenate([c[key].reshape(-1, 3, w) for c in chunks], axis=0)
    return dict(verts=cat("v3f", 3), uvs=cat("t2f", 2), normals=cat("n3f", 3), colors=cat("c3f", 3),
                chunk_sizes=np.array([c["v3f"].size // 9 for c in chunks]), textures=list(mesh.textures),
                min_coords=np.asarray(mesh.min_coords), max_coords=np.asarray(mesh.max_coords))


def make_simulator(map_data: dict, domain_rand: bool = False, max_steps: int = 1500,
                   mesh_extents=None, transform_uses_width: bool = False):
    """Build a reference Simulator *without* running its GL constructor:
    Simulator.__new__ + _interpret_map (simulator.py:788), as in SURVEY 8c."""
    ns = load(mesh_extents)
    S = ns.simulator.Simulator
    sim = S.__new__(S)
    sim.domain_rand = domain_rand
    sim.max_steps = max_steps
    sim.robot_speed = ns.simulator.DEFAULT_ROBOT_SPEED
    sim.step_count = 0
    sim.enable_leds = False
    ns._state.H = None if transform_uses_width else len(map_data["tiles"])
    sim._init_vlists = lambda: None
    sim._interpret_map(map_data)
    ns._state.H = None
    return sim, ns
