"""Fixture assets: maps, procedural tile textures and a stand-in duckie mesh.

The reference loads maps, textures and meshes from the third-party
`duckietown_world` package (simulator.py:638,779; objmesh.py:37), none of which is
available offline.  These deterministic stand-ins are the *inputs* shared
by the HIP path and the test oracle; parity is defined on identical inputs.

  * maps: `small_loop` (= the tile block of small_loop_only_duckies.yaml, no objects),
    `small_loop_only_duckies`, `loop_only_duckies` (the two YAML files in the reference
    root, restated as data), `loop_pedestrians` (= loop_only_duckies with static: False).
  * textures: 256x256 RGBA8 per tile kind, lane markings laid out in the tile-local frame
    the reference's Bezier curves use (simulator.py:1164-1225): road along local z for
    `straight`, arc about the (u,v)=(0,1) corner for `curve_right`, (1,1) for `curve_left`.
  * mesh: low-poly duckie, extents exactly x[-0.5,0.5] y[0,1] z[-0.34375,0.34375]
    (dyadic => exact in float32, so OBB arithmetic is dtype-independent).
"""
from __future__ import annotations

import copy
import math

import numpy as np

TEX_SIZE = 256

_G = "grass"
_SMALL_LOOP_TILES = [
    [_G, _G, _G, _G, _G],
    [_G, "curve_right/N", "straight/E", "curve_right/E", _G],
    [_G, "straight/S", _G, "straight/S", _G],
    [_G, "curve_right/W", "straight/E", "curve_right/S", _G],
    [_G, _G, _G, _G, _G],
]
_LOOP_TILES = [
    [_G] * 8,
    [_G, "curve_left/W", "straight/E", "straight/E", "straight/E", "straight/E", "curve_left/N", _G],
    [_G, "straight/S", _G, _G, _G, _G, "straight/S", _G],
    [_G, "straight/S", _G, _G, _G, _G, "straight/S", _G],
    [_G, "straight/S", _G, _G, "curve_left/W", "straight/E", "curve_left/E", _G],
    [_G, "curve_left/S", "straight/E", "straight/E", "curve_left/E", _G, _G, _G],
    [_G] * 8,
]


def _duckies(spec, static=True):
    return [dict(height=0.06, kind="duckie", optional=False, pos=list(p), rotate=r, static=static)
            for p, r in spec]


_SMALL_DUCKIES = [((2.5, 1.25), 30), ((1.75, 2.5), 60), ((2.5, 3.6), 120), ((3.4, 2.5), 170)]
_LOOP_DUCKIES = [((4.75, 1.25), 60), ((3.25, 1.75), 210), ((1.25, 2.9), 300), ((1.75, 4.1), 270),
                 ((3.25, 5.8), 60), ((2.75, 5.2), 240), ((6.2, 2.75), 310), ((6.8, 3.25), 50)]

def _bots(spec):
    return [dict(kind="duckiebot", pos=list(p), rotate=r, height=0.12, static=False) for p, r in spec]


# follower Duckiebots placed on the right-hand lanes of the loop (stand-in for `loop_dyn_duckiebots`)
_LOOP_BOTS = [((3.5, 1.7), 0), ((1.3, 3.0), -90), ((6.7, 2.5), 90), ((2.5, 5.3), 180)]

MAPS = {
    "small_loop": dict(tiles=_SMALL_LOOP_TILES, objects=[], tile_size=0.585),
    "small_loop_only_duckies": dict(tiles=_SMALL_LOOP_TILES, objects=_duckies(_SMALL_DUCKIES), tile_size=0.585),
    "loop_only_duckies": dict(tiles=_LOOP_TILES, objects=_duckies(_LOOP_DUCKIES), tile_size=0.585),
    "loop_pedestrians": dict(tiles=_LOOP_TILES, objects=_duckies(_LOOP_DUCKIES, static=False), tile_size=0.585),
    "loop_dyn_duckiebots": dict(tiles=_LOOP_TILES, objects=_bots(_LOOP_BOTS) + _duckies(_LOOP_DUCKIES[:3]), tile_size=0.585),
}


def get_map(name: str) -> dict:
    """Map data in the reference's MapFormat1 dict form (what yaml.load returns)."""
    import os
    if name in MAPS:
        return copy.deepcopy(MAPS[name])
    if os.path.isfile(name):
        import yaml
        with open(name) as f:
            return yaml.safe_load(f)
    raise KeyError(f"unknown map {name!r}; fixtures: {sorted(MAPS)} (or a path to a MapFormat1 YAML)")


def map_basename(name: str) -> str:
    import os
    if os.path.isfile(name):  # simulator.py:771-775
        return ".".join(os.path.basename(name).split(".")[:-1])
    return name


# ---------------------------------------------------------------- textures ----
def _noise(rng, n, amp):
    """Smooth-ish photo-like noise: coarse octaves upsampled + per-texel grain."""
    out = np.zeros((n, n))
    for cells, a in ((8, 0.5), (32, 0.3)):
        g = rng.uniform(-1, 1, size=(cells, cells))
        out += a * np.kron(g, np.ones((n // cells, n // cells)))
    out += 0.2 * rng.uniform(-1, 1, size=(n, n))
    return amp * out


def _paint(base, mask, color):
    for k in range(3):
        base[..., k] = np.where(mask, color[k], base[..., k])


def make_texture(kind: str, n: int = TEX_SIZE) -> np.ndarray:
    """RGBA8 [n,n,4]; row 0 = TOP of the image file (v=1).  Use gl_rows() for upload."""
    seed = sum(ord(ch) * (i + 1) for i, ch in enumerate(kind)) + 12345
    rng = np.random.default_rng(seed)
    v, u = np.meshgrid(1.0 - (np.arange(n) + 0.5) / n, (np.arange(n) + 0.5) / n, indexing="ij")
    img = np.zeros((n, n, 3))
    nz = _noise(rng, n, 14.0)
    WHITE, YELLOW, RED = (232, 232, 226), (238, 200, 36), (200, 40, 36)
    if kind == "grass":
        img[:] = (58, 132, 52)
        img[..., 1] += nz * 1.8
        img[..., 0] += nz
        img[..., 2] += nz * 0.6
    elif kind == "floor":
        img[:] = (196, 188, 170)
        img += nz[..., None] * 0.6
    else:
        img[:] = (62, 62, 66)
        img += nz[..., None]
        if kind == "straight":
            _paint(img, (np.abs(u - 0.06) < 0.022) | (np.abs(u - 0.94) < 0.022), WHITE)
            _paint(img, (np.abs(u - 0.5) < 0.012) & ((v * 6.0) % 1.0 < 0.55), YELLOW)
        elif kind in ("curve_right", "curve_left"):
            cu = 0.0 if kind == "curve_right" else 1.0
            r = np.hypot(u - cu, v - 1.0)
            ang = np.arctan2(np.abs(1.0 - v), np.abs(u - cu) + 1e-9)
            _paint(img, (np.abs(r - 0.06) < 0.022) | (np.abs(r - 0.94) < 0.022), WHITE)
            _paint(img, (np.abs(r - 0.5) < 0.012) & ((ang * 8.0 / math.pi * 2) % 1.0 < 0.55), YELLOW)
        elif kind.startswith("3way") or kind == "4way":
            corner = (np.minimum(u, 1 - u) < 0.08) & (np.minimum(v, 1 - v) < 0.08)
            _paint(img, corner, WHITE)
            stop = ((np.abs(v - 0.1) < 0.02) & (u > 0.5) & (u < 0.92)) | ((np.abs(v - 0.9) < 0.02) & (u < 0.5) & (u > 0.08))
            _paint(img, stop, RED)
        # asphalt / unknown: plain
    rgba = np.empty((n, n, 4), dtype=np.uint8)
    rgba[..., :3] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    rgba[..., 3] = 255
    return rgba


def gl_rows(tex: np.ndarray) -> np.ndarray:
    """Row 0 = bottom of the image (v=0): the order pyglet/GL store textures in
    (graphics.py:69-169) and the order dtsim_texture expects."""
    return np.ascontiguousarray(tex[::-1])


_TEX_CACHE: dict = {}


def get_texture(kind: str) -> np.ndarray:
    if kind not in _TEX_CACHE:
        _TEX_CACHE[kind] = gl_rows(make_texture(kind))
    return _TEX_CACHE[kind]


# -------------------------------------------------------------------- mesh ----
def _ellipsoid(center, radii, nu, nv, color):
    cx, cy, cz = center
    rx, ry, rz = radii
    tris, nrms = [], []

    def pt(i, j):
        th = math.pi * i / nv              # polar from +y
        ph = 2 * math.pi * j / nu
        n = np.array([math.sin(th) * math.cos(ph), math.cos(th), math.sin(th) * math.sin(ph)])
        p = np.array([cx + rx * n[0], cy + ry * n[1], cz + rz * n[2]])
        nn = np.array([n[0] / rx, n[1] / ry, n[2] / rz])
        return p, nn / np.linalg.norm(nn)

    for i in range(nv):
        for j in range(nu):
            a, b, c, d = pt(i, j), pt(i, j + 1), pt(i + 1, j + 1), pt(i + 1, j)
            if i > 0:
                tris.append((a[0], b[0], d[0])); nrms.append((a[1], b[1], d[1]))
            if i < nv - 1:
                tris.append((b[0], c[0], d[0])); nrms.append((b[1], c[1], d[1]))
    cols = [np.tile(np.asarray(color, dtype=np.float64), (3, 1)) for _ in tris]
    return tris, nrms, cols


def _cone(base_c, tip, r, n, color):
    bc, tip = np.asarray(base_c, float), np.asarray(tip, float)
    ax = tip - bc
    ax /= np.linalg.norm(ax)
    u = np.cross(ax, [0, 1, 0]); u /= np.linalg.norm(u)
    w = np.cross(ax, u)
    tris, nrms, cols = [], [], []
    for k in range(n):
        a0, a1 = 2 * math.pi * k / n, 2 * math.pi * (k + 1) / n
        p0 = bc + r * (math.cos(a0) * u + math.sin(a0) * w)
        p1 = bc + r * (math.cos(a1) * u + math.sin(a1) * w)
        nn = np.cross(p1 - p0, tip - p0)
        nn /= np.linalg.norm(nn)
        tris.append((p0, p1, tip)); nrms.append((nn, nn, nn))
        cols.append(np.tile(np.asarray(color, float), (3, 1)))
    return tris, nrms, cols


from .objmesh import MeshData, load_obj  # noqa: E402  (MeshData: what ObjMesh exposes to the hot path)


def _finish(parts, lo, hi):
    tris = np.array([t for p in parts for t in p[0]], dtype=np.float64)
    nrms = np.array([t for p in parts for t in p[1]], dtype=np.float64)
    cols = np.array([t for p in parts for t in p[2]], dtype=np.float64)
    mn, mx = tris.reshape(-1, 3).min(0), tris.reshape(-1, 3).max(0)
    lo, hi = np.asarray(lo, float), np.asarray(hi, float)
    tris = (tris - mn) / (mx - mn) * (hi - lo) + lo          # exact target extents
    tris = np.rint(tris * 1024.0) / 1024.0                    # dyadic grid: exact in float32
    return MeshData(tris, nrms, cols)


_MESH_CACHE: dict = {}


def get_mesh(kind: str) -> MeshData:
    """Stand-in meshes.  `duckie`: body + head + beak; anything else: a generic
    non-square-footprint blob (square footprints make generate_norm's eigenvectors
    arbitrary, collision.py:99-106 / SURVEY App. A)."""
    key = "duckie" if kind == "duckie" else "*"
    if key not in _MESH_CACHE:
        if key == "duckie":
            yellow, orange = (0.96, 0.80, 0.10), (0.93, 0.45, 0.08)
            parts = [
                _ellipsoid((0.0, 0.36, 0.0), (0.5, 0.36, 0.34375), 10, 6, yellow),
                _ellipsoid((0.27, 0.78, 0.0), (0.21, 0.22, 0.2), 8, 5, yellow),
                _cone((0.44, 0.76, 0.0), (0.5, 0.73, 0.0), 0.07, 5, orange),
            ]
            _MESH_CACHE[key] = _finish(parts, (-0.5, 0.0, -0.34375), (0.5, 1.0, 0.34375))
        else:
            parts = [_ellipsoid((0.0, 0.5, 0.0), (0.5, 0.5, 0.40625), 8, 5, (0.7, 0.7, 0.72))]
            _MESH_CACHE[key] = _finish(parts, (-0.5, 0.0, -0.40625), (0.5, 1.0, 0.40625))
    return _MESH_CACHE[key]


def mesh_kind(kind: str) -> str:
    """Which stand-in mesh a map object kind uses."""
    return "duckie" if kind == "duckie" else "*"


def mesh_extents(kinds=("duckie",)) -> dict:
    """kind -> (min_coords, max_coords) float32, plus '*' fallback."""
    out = {"*": (get_mesh("*").min_coords, get_mesh("*").max_coords)}
    for k in kinds:
        m = get_mesh(k)
        out[k] = (m.min_coords, m.max_coords)
    return out


# ------------------------------------------------------------ asset library ----
def load_image_rgba(path: str) -> np.ndarray:
    """Image file -> RGBA8 [h,w,4] in GL row order (row 0 = bottom, graphics.py:69-169).  Alpha is
    carried but unused (the reference never enables blending for tiles / meshes)."""
    from PIL import Image
    with Image.open(path) as im:
        arr = np.asarray(im.convert("RGBA"), dtype=np.uint8)
    return gl_rows(arr)


def _pow2_near(n: int) -> int:
    return 1 << max(0, int(round(math.log2(max(int(n), 1)))))


def to_pow2(tex: np.ndarray, size=None) -> np.ndarray:
    """The raster wraps texel indices with a mask: textures must be power-of-two sized (and all
    tile textures of one simulator share one size).  Other sizes are resampled (bilinear); the
    reference would upload them as they are, so this is the one place where a real asset can
    render differently -- power-of-two assets (the duckietown ones are 512 / 1024) are untouched."""
    h, w = tex.shape[:2]
    th, tw = (size, size) if size else (_pow2_near(h), _pow2_near(w))
    if (h, w) == (th, tw):
        return np.ascontiguousarray(tex)
    from PIL import Image
    im = Image.fromarray(tex[::-1]).resize((tw, th), Image.BILINEAR)
    return gl_rows(np.asarray(im, dtype=np.uint8))


class AssetLibrary:
    """Resolves maps, tile textures and meshes the way the reference asks `duckietown_world` for them
    (by basename / by `tiles-processed/<style>/<kind>/texture`, simulator.py:638,779; objmesh.py:37),
    from a directory tree `root` (e.g. a checkout of duckietown-world's `data/`), falling back to the
    deterministic fixtures of this module for anything that is not found.  `root=None` (and no
    `DTSIM_ASSET_ROOT` in the environment) => fixtures only."""

    IMG_EXT = (".jpg", ".jpeg", ".png")

    def __init__(self, root: str | None = None, style: str = "photos"):
        import os
        root = root if root is not None else os.environ.get("DTSIM_ASSET_ROOT")
        self.root = os.path.abspath(root) if root else None
        self.style = style
        self._by_name: dict = {}
        self._files: list = []
        if self.root:
            if not os.path.isdir(self.root):
                raise FileNotFoundError(f"asset root {self.root!r} is not a directory")
            for d, _sub, fs in sorted(os.walk(self.root)):
                for f in sorted(fs):
                    p = os.path.join(d, f)
                    self._files.append(p)
                    self._by_name.setdefault(f, p)          # first hit wins (get_resource_path)
        self._meshes: dict = {}
        self._tex: dict = {}

    # get_resource_path(basename)
    def resolve(self, basename: str):
        return self._by_name.get(basename)

    def map_data(self, name: str) -> dict:
        import os
        if name in MAPS or os.path.isfile(name):
            return get_map(name)
        p = self.resolve(f"{name}.yaml")
        if p is None:
            raise KeyError(f"unknown map {name!r}: not a fixture {sorted(MAPS)}, not a file, not under the asset root")
        import yaml
        with open(p) as f:
            return yaml.safe_load(f)

    def tile_texture_file(self, kind: str):
        """get_texture_file(f"tiles-processed/{style}/{kind}/texture")[0] (simulator.py:638)."""
        import os
        tail = os.path.join("tiles-processed", self.style, kind, "texture")
        for p in self._files:
            stem, ext = os.path.splitext(p)
            if ext.lower() in self.IMG_EXT and stem.endswith(tail):
                return p
        return None

    def tile_texture(self, kind: str) -> np.ndarray:
        if kind not in self._tex:
            p = self.tile_texture_file(kind) if self.root else None
            self._tex[kind] = to_pow2(load_image_rgba(p)) if p else get_texture(kind)
        return self._tex[kind]

    def mesh(self, kind: str) -> MeshData:
        """get_mesh(kind) (objmesh.py:28-52): `<kind>.obj` by basename; stand-in mesh otherwise."""
        if kind not in self._meshes:
            p = self.resolve(f"{kind}.obj") if self.root else None
            if p:
                m = load_obj(p, self.resolve, name=kind)
                m.textures = [to_pow2(load_image_rgba(t)) for t in m.texture_files]
            else:
                m = get_mesh(kind)
            self._meshes[kind] = m
        return self._meshes[kind]

    # chassis colours by name (duckietown_world.get_duckiebot_color_from_colorname is not available
    # offline; these are the usual RGB triples -- parity unpinned)
    BOT_COLORS = {"red": (1.0, 0.0, 0.0), "green": (0.0, 0.5, 0.0), "blue": (0.0, 0.0, 1.0), "yellow": (1.0, 1.0, 0.0),
                  "grey": (0.3, 0.3, 0.3), "gray": (0.3, 0.3, 0.3), "white": (1.0, 1.0, 1.0), "black": (0.0, 0.0, 0.0),
                  "orange": (1.0, 0.5, 0.0), "purple": (0.5, 0.0, 0.5), "pink": (1.0, 0.4, 0.7), "cyan": (0.0, 1.0, 1.0)}

    def object_mesh(self, desc: dict):
        """(cache key, mesh) of one map object, following simulator.py:958-974: duckiebots use the
        `duckiebot` mesh with the chassis materials recoloured, `sign*` kinds use `sign_generic` with
        the `April_Tag` material's texture replaced by `<kind>.png`, everything else `<kind>.obj`."""
        import os
        kind = desc["kind"]
        if not self.root:
            return ("duckie" if kind == "duckie" else "*"), get_mesh(kind)
        if kind == "duckiebot":
            cname = desc.get("color", "red")
            key, base, p = f"duckiebot:{cname}", "duckiebot", self.resolve("duckiebot.obj")
            col = np.array(self.BOT_COLORS.get(cname, self.BOT_COLORS["red"]))
            change = {"gkmodel0_chassis_geom0_mat_001-material": {"Kd": col},
                      "gkmodel0_chassis_geom0_mat_001-material.001": {"Kd": col}}
        elif kind.startswith("sign"):
            key, base, p = kind, "sign_generic", self.resolve("sign_generic.obj")
            change = {"April_Tag": {"map_Kd": f"{kind}.png"}}
        else:
            key, base, p, change = kind, kind, self.resolve(f"{kind}.obj"), None
        if p is None:
            return ("duckie" if kind == "duckie" else "*"), get_mesh(kind)
        if key not in self._meshes:
            m = load_obj(p, self.resolve, name=key, change_materials=change)
            m.textures = [to_pow2(load_image_rgba(t)) for t in m.texture_files]
            m.seg_name = base                      # get_mesh(mesh_name, segment=True) hashes the mesh name
            self._meshes[key] = m
        return key, self._meshes[key]

    def light_cards(self):
        """TrafficLightObj's two card textures (objects.py:438-441), or None when the tree has none."""
        ps = [self.resolve(f"trafficlight_card{k}.jpg") for k in (0, 1)] if self.root else [None, None]
        if None in ps:
            return None
        return [to_pow2(load_image_rgba(p)) for p in ps]

    def mesh_extents(self, kinds) -> dict:
        return {k: (self.mesh(k).min_coords, self.mesh(k).max_coords) for k in kinds}


# ---------------------------------------------------------------- segmentation assets --
# `render(segment=True)` swaps every texture for a segmented version (graphics.py:52-126) and every
# mesh for a flat-coloured one (objmesh.py:255-292).  The tile branch of load_texture() goes through
# OpenCV (8-bit BGR->HSV, inRange, erode, HSV->BGR); cv2 is absent offline, so its 8-bit colour
# conversions are restated below from OpenCV's documented algorithm -- parity unpinned.

def gen_segmentation_color(name: str):
    """objmesh.py:260-266: decimal character codes concatenated, cut into 3-digit groups, each % 255."""
    hashed = "".join(str(ord(ch)) for ch in name)
    col = [int(hashed[i:i + 3]) % 255 for i in range(0, len(hashed), 3)][:3]
    if len(col) != 3:
        raise ValueError(f"mesh name {name!r} is too short for gen_segmentation_color")
    return col


def should_segment_out(tex_path: str) -> bool:
    """graphics.py:59-67."""
    for yes in ("sign", "trafficlight", "asphalt"):
        if yes in tex_path:
            return True
    for no in ("left", "right", "way", "curve", "straight"):
        if no in tex_path:
            return False
    return True


def _round_half_even_div(num: int, den: float) -> int:
    return int(np.rint(num / den))


_HSV_SHIFT = 12
_SDIV = np.array([0] + [_round_half_even_div(255 << _HSV_SHIFT, 1.0 * i) for i in range(1, 256)], np.int64)
_HDIV180 = np.array([0] + [_round_half_even_div(180 << _HSV_SHIFT, 6.0 * i) for i in range(1, 256)], np.int64)


def bgr2hsv_u8(bgr: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(bgr, COLOR_BGR2HSV) for uint8 (OpenCV RGB2HSV_b, hrange 180): integer arithmetic with the
    12-bit reciprocal tables."""
    b, g, r = (bgr[..., k].astype(np.int64) for k in range(3))
    v = np.maximum(np.maximum(b, g), r)
    vmin = np.minimum(np.minimum(b, g), r)
    diff = v - vmin
    vr = np.where(v == r, -1, 0)
    vg = np.where(v == g, -1, 0)
    half = 1 << (_HSV_SHIFT - 1)
    s = (diff * _SDIV[v] + half) >> _HSV_SHIFT
    h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))))
    h = (h * _HDIV180[diff] + half) >> _HSV_SHIFT
    h = h + np.where(h < 0, 180, 0)
    return np.stack([np.clip(h, 0, 255), s & 255, v], axis=-1).astype(np.uint8)


def hsv2bgr_u8(hsv: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(hsv, COLOR_HSV2BGR) for uint8 (OpenCV HSV2RGB_b -> HSV2RGB_native in float32, hrange 180,
    saturate_cast<uchar>(x * 255) = round half to even)."""
    f = np.float32
    h = hsv[..., 0].astype(f) * f(6.0 / 180.0)
    s = hsv[..., 1].astype(f) * f(1.0 / 255.0)
    v = hsv[..., 2].astype(f) * f(1.0 / 255.0)
    h = np.fmod(h, f(6.0))
    sector = np.floor(h).astype(np.int64)
    h = h - sector.astype(f)
    bad = (sector < 0) | (sector >= 6)
    sector = np.where(bad, 0, sector)
    h = np.where(bad, f(0), h)
    one = f(1.0)
    tab = np.stack([v, v * (one - s), v * (one - s * h), v * (one - s * (one - h))], axis=-1)
    sector_data = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])
    idx = sector_data[sector]                                    # [..., 3] -> b, g, r
    bgr = np.take_along_axis(tab, idx, axis=-1)
    bgr = np.where((s == 0)[..., None], v[..., None], bgr)
    return np.clip(np.rint(bgr * f(255.0)), 0, 255).astype(np.uint8)


def segment_texture(tex: np.ndarray, tex_path: str, into_color=(0, 0, 0)) -> np.ndarray:
    """load_texture(tex_path, segment=True, segment_into_color) (graphics.py:91-126) applied to an RGBA8 texture in
    GL row order: either a flat fill, or (lane-marking tiles) everything that is not saturated / bright paint
    blacked out."""
    out = np.empty_like(tex)
    out[..., 3] = 255
    if should_segment_out(tex_path):
        out[..., :3] = np.asarray(into_color, np.int64).astype(np.uint8)
        return out
    bgr = tex[::-1, :, 2::-1]                                    # image row order, B G R
    hsv = bgr2hsv_u8(bgr)
    inr = (hsv[..., 0] <= 179) & (hsv[..., 1] <= 100) & (hsv[..., 2] <= 160)      # inRange([0,0,0],[179,100,160])
    keep = ~inr                                                  # bitwise_not; erode by the centre-only kernel = identity
    pad = np.pad(keep, 1, constant_values=True)                  # erode's border never lowers the minimum
    H, W = keep.shape
    ring = np.ones_like(keep)
    for dy in (0, 1, 2):
        for dx in (0, 1, 2):
            if (dy, dx) != (1, 1):
                ring &= pad[dy:dy + H, dx:dx + W]                # erode by the 8-neighbour ring kernel
    mask = keep & ring
    hsv = np.where(mask[..., None], hsv, 0).astype(np.uint8)
    res = hsv2bgr_u8(hsv)
    out[..., :3] = res[::-1, :, ::-1]
    return out
