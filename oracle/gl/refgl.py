"""Run the REFERENCE's Simulator -- constructor, reset(), step(), render_obs() / render() -- UNMODIFIED on real OpenGL.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Works only where /root/reference exists (the build container): used
by oracle/make_gl_golden.py to produce tests/golden/ref_gl_*.npz, by the `not gpu` tests that pin oracle/raster.py to GL
frames, and by oracle/time_reference_render.py.  Nothing under tests/ -m gpu, smoke() or bench.py may call it.

What is real and what is a stand-in when the reference runs here:

  real        every line of /root/reference/src/gym_duckietown (imported from where it lies, nothing patched after import);
              OpenGL = Mesa 23.2.1 llvmpipe, the reference CI's own renderer (.circleci/config.yml:10,26), through
              oracle/gl/gl_headless.c; numpy, yaml, PIL.
  shim        pyglet (oracle/gl/glshim.py: the calls pyglet 1.4/1.5 makes for the same API); gym (Env, spaces.Box,
              seeding.np_random -> numpy Generator, as gym >= 0.21 returns); zuper_commons (logger, ZException);
              duckietown_world.resources / get_texture_file (basename lookup in an asset directory, as the package does
              over its data/ tree); MapFormat1Constants.
  restated    duckietown_world dynamics (get_DB18_nominal / get_DB18_uncalibrated -> oracle/sim.py:DynamicsDB18 -- parity
              unpinned, see there: it moves the robot between frames and has no part in how a frame is drawn), get_transform
              (README.md:239 semantics, as oracle/refstub.py), PyGeometry's three SE2 closed forms,
              get_duckiebot_color_from_colorname (a colour table).
  absent      cv2 / carnivalmirror: no fisheye remap (frames are produced with distortion=False).  The eight cv2 calls of
              load_texture(segment=True) are stood in for by _cv2_stand_in (the restated 8-bit BGR <-> HSV conversions).
"""
from __future__ import annotations

import importlib
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np

from oracle import refstub
from oracle.gl import glshim

_ns = None


def available() -> bool:
    return refstub.available() and glshim.available()


class _Box:
    def __init__(self, low=None, high=None, shape=None, dtype=None):
        self.low, self.high, self.dtype = low, high, dtype
        self.shape = tuple(shape) if shape is not None else np.shape(low)


class _Dyn:
    """The slice of duckietown_world's PlatformDynamics the reference touches (simulator.py:745-755, 2083-2085) over
    oracle/sim.py:DynamicsDB18."""

    def __init__(self, trim=None):
        self.trim = trim

    def initialize(self, c0, t0=0, seed=None):
        from oracle.sim import DynamicsDB18
        q, _v = c0
        st = _Dyn(self.trim)
        st.m = DynamicsDB18(q[0, 2], q[1, 2], float(np.arctan2(q[1, 0], q[0, 0])), trim=self.trim)
        return st

    def integrate(self, dt, commands):
        self.m.integrate(dt, commands.motor_left, commands.motor_right)      # (in place: the reference rebinds self.state to the result)
        return self

    def TSE2_from_state(self):
        m = self.m
        return np.array([[m.c, -m.s, m.x], [m.s, m.c, m.y], [0.0, 0.0, 1.0]]), None


class AssetDir:
    """duckietown_world.resources over a plain directory tree: get_resource_path(basename) -> first file of that name
    (KeyError when there is none, as the package raises); get_texture_file(name) -> files `<name>.{jpg,jpeg,png}`."""

    def __init__(self, roots):
        self.files, self.by_name = [], {}
        for root in roots:
            for d, _sub, fs in sorted(os.walk(root)):
                for f in sorted(fs):
                    p = os.path.join(d, f)
                    self.files.append(p)
                    self.by_name.setdefault(f, p)

    def get_resource_path(self, basename):
        if basename not in self.by_name:
            raise KeyError(basename)
        return self.by_name[basename]

    def get_texture_file(self, name):
        out = [p for p in self.files if os.path.splitext(p)[1].lower() in (".jpg", ".jpeg", ".png") and os.path.splitext(p)[0].endswith(name)]
        if not out:
            raise KeyError(name)
        return out


def _cv2_stand_in():
    """The eight cv2 calls of load_texture(segment=True)'s lane-marking branch (graphics.py:100-126), so that the reference's segmentation
    render can run: imread / cvtColor(BGR2HSV, HSV2BGR) / inRange / bitwise_not / morphologyEx(MORPH_ERODE) / bitwise_and.  The 8-bit
    colour conversions are the restatement of OpenCV's integer algorithm the product uses too (dtsim.assets.bgr2hsv_u8 / hsv2bgr_u8;
    pinned against cv2 where it is installed, tests/test_thirdparty_pins.py): what the segmentation goldens pin is what GL does with the
    textures, not OpenCV."""
    from dtsim import assets as A
    cv2 = types.ModuleType("cv2")
    cv2.IMREAD_UNCHANGED, cv2.COLOR_BGR2HSV, cv2.COLOR_HSV2BGR, cv2.MORPH_ERODE = -1, 40, 54, 0
    cv2.INTER_NEAREST, cv2.INTER_LINEAR, cv2.INTER_CUBIC, cv2.INTER_AREA, cv2.BORDER_CONSTANT, cv2.CV_32FC1 = 0, 1, 2, 3, 0, 5   # (default arguments of distortion.py / wrappers.py)

    def imread(path, flags=None):
        from PIL import Image
        with Image.open(path) as im:
            a = np.asarray(im.convert("RGBA" if im.mode in ("RGBA", "LA", "P") else "RGB"), dtype=np.uint8)
        return np.ascontiguousarray(a[..., [2, 1, 0] + ([3] if a.shape[-1] == 4 else [])])

    def cvtColor(img, code):
        if code == cv2.COLOR_BGR2HSV:
            return A.bgr2hsv_u8(img[..., :3])
        if code == cv2.COLOR_HSV2BGR:
            return A.hsv2bgr_u8(img)
        raise NotImplementedError(code)

    def inRange(img, lo, hi):
        return (np.all((img >= np.asarray(lo)) & (img <= np.asarray(hi)), axis=-1) * 255).astype(np.uint8)

    def morphologyEx(src, op, kernel):                     # erode: minimum over the kernel's support, the border never lowers it
        assert op == cv2.MORPH_ERODE and kernel.shape == (3, 3)
        pad = np.pad(src, 1, constant_values=255)
        out = np.full_like(src, 255)
        H, W = src.shape
        for dy in range(3):
            for dx in range(3):
                if kernel[dy, dx]:
                    out = np.minimum(out, pad[dy:dy + H, dx:dx + W])
        return out

    def bitwise_and(a, b, mask=None):
        r = a & b
        if mask is not None:
            r = np.where((mask != 0)[..., None] if r.ndim == 3 else (mask != 0), r, 0).astype(a.dtype)
        return r

    cv2.imread, cv2.cvtColor, cv2.inRange, cv2.morphologyEx, cv2.bitwise_and = imread, cvtColor, inRange, morphologyEx, bitwise_and
    cv2.bitwise_not = lambda a: (255 - a).astype(a.dtype)
    return cv2


BOT_COLORS = {"red": (1.0, 0.0, 0.0), "green": (0.0, 0.5, 0.0), "blue": (0.0, 0.0, 1.0), "yellow": (1.0, 1.0, 0.0),
              "grey": (0.3, 0.3, 0.3), "gray": (0.3, 0.3, 0.3), "white": (1.0, 1.0, 1.0), "black": (0.0, 0.0, 0.0),
              "orange": (1.0, 0.5, 0.0), "purple": (0.5, 0.0, 0.5), "pink": (1.0, 0.4, 0.7), "cyan": (0.0, 1.0, 1.0)}   # = dtsim.assets.AssetLibrary.BOT_COLORS


def load(transform_uses_width: bool = False):
    """Import the reference package against the real-GL pyglet shim (once per process).  Returns a namespace with the
    reference modules (.simulator, .graphics, .objects, .objmesh, ...), .gl (the GL module they call), .assets (an AssetDir
    holder: set with use_assets()) and .state (H = grid height for get_transform's README reading)."""
    global _ns
    if _ns is not None:
        return _ns
    if not available():
        raise RuntimeError("needs /root/reference and Mesa's swrast_dri.so")
    mods = glshim.install()
    gl = mods["pyglet.gl"]
    state = types.SimpleNamespace(H=None, assets=None)

    def mock(name):
        m = MagicMock(name=name)
        m.__path__ = []
        return m

    names = ["cv2", "gym", "gym.spaces", "gym.utils", "gym.utils.seeding", "gym.envs", "gym.envs.registration",
             "duckietown_world", "duckietown_world.resources", "duckietown_world.gltf", "duckietown_world.gltf.export",
             "duckietown_world.world_duckietown", "duckietown_world.world_duckietown.map_loading",
             "zuper_commons", "zuper_commons.logs", "zuper_commons.types", "geometry", "carnivalmirror", "zmq"]
    shim = {n: mock(n) for n in names}
    shim.update(mods)
    shim["cv2"] = _cv2_stand_in()

    class _Env:
        metadata = {}

        @property
        def unwrapped(self):
            return self

        def close(self):
            pass

    gym = shim["gym"]
    gym.Env = _Env
    gym.spaces = shim["gym.spaces"]
    gym.spaces.Box = _Box
    shim["gym.utils"].seeding = shim["gym.utils.seeding"]
    shim["gym.utils.seeding"].np_random = lambda seed=None: (np.random.default_rng(seed), seed)
    for n in ("Wrapper", "ActionWrapper", "ObservationWrapper", "RewardWrapper"):
        setattr(gym, n, type(n, (), {"__init__": lambda self, env=None: setattr(self, "env", env)}))

    class _ZException(Exception):
        def __init__(self, msg="", **kw):
            super().__init__(f"{msg} {kw}" if kw else msg)

    shim["zuper_commons.types"].ZException = _ZException

    class _MF1C:
        KIND_DUCKIEBOT, KIND_DUCKIE, KIND_TRAFFICLIGHT, KIND_CHECKERBOARD = "duckiebot", "duckie", "trafficlight", "checkerboard"
        ObjectKind = str

    dw = shim["duckietown_world"]
    dw.MapFormat1Constants = _MF1C
    dw.get_DB18_nominal = lambda delay: _Dyn(None)
    dw.get_DB18_uncalibrated = lambda delay, trim: _Dyn(float(trim))
    dw.get_texture_file = lambda name: state.assets.get_texture_file(name)
    res = shim["duckietown_world.resources"]
    res.get_resource_path = lambda bn: state.assets.get_resource_path(bn)
    res.list_maps2 = lambda: {}
    shim["duckietown_world.gltf.export"].get_duckiebot_color_from_colorname = lambda c: list(BOT_COLORS.get(c, BOT_COLORS["red"])) + [1.0]
    geo = shim["geometry"]
    geo.SE2_from_translation_angle = refstub._SE2_from_translation_angle
    geo.translation_angle_from_SE2 = refstub._translation_angle_from_SE2
    geo.se2_from_linear_angular = refstub._se2_from_linear_angular

    def get_transform(desc, W, tile_size):
        H = W if (transform_uses_width or state.H is None) else state.H
        pos = desc["pos"]
        return refstub._SE2Transform([pos[0] * tile_size, (H - pos[1]) * tile_size], np.deg2rad(desc.get("rotate", 0.0)))

    shim["duckietown_world.world_duckietown.map_loading"].get_transform = get_transform

    saved = {n: sys.modules.get(n) for n in shim}
    saved_ref = {k: v for k, v in sys.modules.items() if k == "gym_duckietown" or k.startswith("gym_duckietown.")}
    for k in saved_ref:
        del sys.modules[k]
    sys.modules.update(shim)
    sys.path.insert(0, refstub.REFERENCE_SRC)
    ns = types.SimpleNamespace(gl=gl, state=state)
    try:
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):          # (gym_duckietown/__init__.py prints pyglet.options)
            for short in ("simulator", "collision", "graphics", "objects", "objmesh", "check_hw"):
                setattr(ns, short, importlib.import_module(f"gym_duckietown.{short}"))
            ns.duckietown_env = importlib.import_module("gym_duckietown.envs.duckietown_env")
    finally:
        sys.path.remove(refstub.REFERENCE_SRC)
        for k in [k for k in sys.modules if k == "gym_duckietown" or k.startswith("gym_duckietown.")]:
            del sys.modules[k]
        sys.modules.update(saved_ref)
        for n, m in saved.items():
            if m is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = m
    assert ns.simulator.gl is gl and ns.graphics.gl is gl and ns.objects.gl is gl and ns.objmesh.gl is gl
    _ns = ns
    return ns


def use_assets(roots, grid_height=None):
    """Point the reference's resource lookups at `roots` (directories) and forget what it cached from earlier ones
    (graphics.load_texture's lru_cache, Texture.tex_cache, ObjMesh.cache: all keyed by path)."""
    ns = load()
    ns.state.assets = AssetDir(list(roots))
    ns.state.H = grid_height
    ns.graphics.load_texture.cache_clear()
    ns.graphics.Texture.tex_cache.clear()
    ns.objmesh.ObjMesh.cache.clear()
    return ns


def make_simulator(map_name: str, roots, *, env_class: str = "Simulator", **kw):
    """Simulator(map_name=..., **kw) -- the reference's own constructor, run to the end (it resets and renders once)."""
    import yaml
    ns = load()
    use_assets(roots)
    with open(ns.state.assets.get_resource_path(f"{map_name}.yaml")) as f:
        ns.state.H = len(yaml.safe_load(f)["tiles"])
    cls = ns.simulator.Simulator if env_class == "Simulator" else ns.duckietown_env.DuckietownEnv
    # the reference draws the domain-randomised parameters of its objects (TrafficLightObj's frequency and pattern, DuckiebotObj's gains) from
    # numpy's GLOBAL generator (objects.py:180-228, 434-463), which nothing seeds: seed it here so that a recipe run is reproducible
    np.random.seed(20_000 + int(kw.get("seed") or 0))
    return cls(map_name=map_name, **kw), ns
