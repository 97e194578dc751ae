"""Generate tests/golden/assets/: a small duckietown-world style asset tree (MapFormat1 YAML map,
tile textures, OBJ/MTL/PNG meshes) used to test real-asset ingestion.

    python tests/golden/make_assets.py

Deterministic; the generated files are committed (a few tens of KB).  Nothing here comes from
the reference or from duckietown-world: the geometry is procedural and only the *formats* (and the
material names the reference's simulator.py:958-974 / 2091-2099 looks for) are the real ones.
"""
from __future__ import annotations

import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "gym-duckietown_amd"))
OUT = os.path.join(HERE, "assets")


def save_png(path, rgb):
    from PIL import Image
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(np.asarray(rgb, np.uint8)).save(path, optimize=True)


class Obj:
    """Tiny OBJ writer: shared position / texcoord / normal pools, faces per material."""

    def __init__(self):
        self.v, self.vt, self.vn, self.lines = [], [], [], []

    def _idx(self, pool, val):
        val = tuple(round(float(x), 6) for x in val)
        if val not in pool:
            pool.append(val)
        return pool.index(val) + 1

    def usemtl(self, name):
        self.lines.append(f"usemtl {name}")

    def tri(self, ps, ns, ts=None):
        toks = []
        for k in range(3):
            vi, ni = self._idx(self.v, ps[k]), self._idx(self.vn, ns[k])
            toks.append(f"{vi}/{self._idx(self.vt, ts[k])}/{ni}" if ts is not None else f"{vi}//{ni}")
        self.lines.append("f " + " ".join(toks))

    def quad(self, ps, n, ts=None):
        self.tri([ps[0], ps[1], ps[2]], [n] * 3, None if ts is None else [ts[0], ts[1], ts[2]])
        self.tri([ps[0], ps[2], ps[3]], [n] * 3, None if ts is None else [ts[0], ts[2], ts[3]])

    def box(self, lo, hi, ts=False):
        x0, y0, z0 = lo; x1, y1, z1 = hi
        uv = [(0, 0), (1, 0), (1, 1), (0, 1)] if ts else None
        self.quad([(x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1)], (0, 0, 1), uv)
        self.quad([(x1, y0, z0), (x0, y0, z0), (x0, y1, z0), (x1, y1, z0)], (0, 0, -1), uv)
        self.quad([(x1, y0, z1), (x1, y0, z0), (x1, y1, z0), (x1, y1, z1)], (1, 0, 0), uv)
        self.quad([(x0, y0, z0), (x0, y0, z1), (x0, y1, z1), (x0, y1, z0)], (-1, 0, 0), uv)
        self.quad([(x0, y1, z1), (x1, y1, z1), (x1, y1, z0), (x0, y1, z0)], (0, 1, 0), uv)
        self.quad([(x0, y0, z0), (x1, y0, z0), (x1, y0, z1), (x0, y0, z1)], (0, -1, 0), uv)

    def write(self, path, mtllib=None):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write("# procedural test asset (tests/golden/make_assets.py)\n")
            if mtllib:
                f.write(f"mtllib {mtllib}\n")
            f.write("o mesh\n")
            for p in self.v:
                f.write("v %.6f %.6f %.6f\n" % p)
            for t in self.vt:
                f.write("vt %.6f %.6f\n" % t)
            for n in self.vn:
                f.write("vn %.6f %.6f %.6f\n" % n)
            for ln in self.lines:
                f.write(ln + "\n")


def write_mtl(path, mats):
    with open(path, "w") as f:
        f.write("# procedural test asset\n")
        for name, m in mats.items():
            f.write(f"\nnewmtl {name}\n")
            if "Kd" in m:
                f.write("Kd %.4f %.4f %.4f\n" % tuple(m["Kd"]))
            if "map_Kd" in m:
                f.write(f"map_Kd {m['map_Kd']}\n")


def cone(d):
    o, n = Obj(), 8
    o.usemtl("cone_mat")
    r0, r1, h = 0.14, 0.03, 0.42
    for k in range(n):
        a0, a1 = 2 * math.pi * k / n, 2 * math.pi * (k + 1) / n
        p = [(r0 * math.cos(a0) + 0.3, 0.02, r0 * math.sin(a0) * 0.8 - 0.1), (r0 * math.cos(a1) + 0.3, 0.02, r0 * math.sin(a1) * 0.8 - 0.1),
             (r1 * math.cos(a1) + 0.3, h, r1 * math.sin(a1) * 0.8 - 0.1), (r1 * math.cos(a0) + 0.3, h, r1 * math.sin(a0) * 0.8 - 0.1)]
        nm = (math.cos((a0 + a1) / 2), 0.25, math.sin((a0 + a1) / 2))
        o.quad(p, nm, [(k / n, 0), ((k + 1) / n, 0), ((k + 1) / n, 1), (k / n, 1)])
    o.usemtl("base_mat")
    o.box((0.3 - 0.18, 0.0, -0.1 - 0.15), (0.3 + 0.18, 0.02, -0.1 + 0.15))
    o.write(os.path.join(d, "cone.obj"), "cone.mtl")
    write_mtl(os.path.join(d, "cone.mtl"), {"cone_mat": {"Kd": (1.0, 0.9, 0.85), "map_Kd": "cone_stripes.png"},
                                           "base_mat": {"Kd": (0.12, 0.12, 0.12)}})
    v, u = np.meshgrid(np.arange(64), np.arange(64), indexing="ij")
    img = np.zeros((64, 64, 3), np.uint8)
    img[:] = (235, 96, 20)
    img[(v // 10) % 3 == 1] = (240, 240, 235)
    img[..., 1] = np.clip(img[..., 1].astype(int) + (u % 7) * 2, 0, 255)
    save_png(os.path.join(d, "cone_stripes.png"), img)


def sign_generic(d):
    o = Obj()
    o.usemtl("post")
    o.box((-0.012, 0.0, -0.012), (0.012, 0.30, 0.012))
    o.usemtl("April_Tag")
    o.quad([(-0.06, 0.30, 0.013), (0.06, 0.30, 0.013), (0.06, 0.42, 0.013), (-0.06, 0.42, 0.013)], (0, 0, 1),
           [(0, 0), (1, 0), (1, 1), (0, 1)])
    o.usemtl("plate")
    o.quad([(0.07, 0.28, -0.013), (-0.07, 0.28, -0.013), (-0.07, 0.44, -0.013), (0.07, 0.44, -0.013)], (0, 0, -1))
    o.write(os.path.join(d, "sign_generic.obj"), "sign_generic.mtl")
    write_mtl(os.path.join(d, "sign_generic.mtl"), {"post": {"Kd": (0.55, 0.55, 0.58)}, "plate": {"Kd": (0.8, 0.8, 0.8)},
                                                   "April_Tag": {"Kd": (1.0, 1.0, 1.0), "map_Kd": "april_default.png"}})
    rng = np.random.default_rng(5)
    tag = (rng.integers(0, 2, (8, 8)) * 255).astype(np.uint8)
    save_png(os.path.join(d, "april_default.png"), np.repeat(np.repeat(tag, 4, 0), 4, 1)[..., None].repeat(3, 2))
    v, u = np.meshgrid(np.linspace(-1, 1, 64), np.linspace(-1, 1, 64), indexing="ij")
    stop = np.zeros((64, 64, 3), np.uint8)
    stop[:] = (250, 250, 250)
    stop[(np.abs(u) + np.abs(v) < 1.35) & (np.maximum(np.abs(u), np.abs(v)) < 0.92)] = (196, 24, 30)
    stop[(np.abs(v) < 0.16) & (np.abs(u) < 0.6)] = (245, 245, 245)
    save_png(os.path.join(d, "sign_stop.png"), stop)


def duckiebot(d):
    o = Obj()
    o.usemtl("gkmodel0_chassis_geom0_mat_001-material")
    o.box((-0.09, 0.02, -0.06), (0.08, 0.07, 0.06))
    o.usemtl("gkmodel0_chassis_geom0_mat_001-material.001")
    o.box((-0.05, 0.07, -0.045), (0.06, 0.105, 0.045))
    o.usemtl("wheel")
    o.box((0.0, 0.0, 0.06), (0.066, 0.066, 0.075))
    o.box((0.0, 0.0, -0.075), (0.066, 0.066, -0.06))
    o.usemtl("camera")
    o.box((0.08, 0.09, -0.015), (0.095, 0.115, 0.015))
    o.write(os.path.join(d, "duckiebot.obj"), "duckiebot.mtl")
    write_mtl(os.path.join(d, "duckiebot.mtl"), {"gkmodel0_chassis_geom0_mat_001-material": {"Kd": (0.8, 0.1, 0.1)},
                                                "gkmodel0_chassis_geom0_mat_001-material.001": {"Kd": (0.8, 0.1, 0.1)},
                                                "wheel": {"Kd": (0.05, 0.05, 0.05)}, "camera": {"Kd": (0.1, 0.3, 0.1)}})


def tree(d):
    o, n = Obj(), 6
    o.usemtl("trunk")
    o.box((1.0 - 0.04, 0.5, 2.0 - 0.05), (1.0 + 0.04, 0.9, 2.0 + 0.05))
    o.usemtl("crown")
    for k in range(n):
        a0, a1 = 2 * math.pi * k / n, 2 * math.pi * (k + 1) / n
        b0 = (1.0 + 0.22 * math.cos(a0), 0.85, 2.0 + 0.3 * math.sin(a0))
        b1 = (1.0 + 0.22 * math.cos(a1), 0.85, 2.0 + 0.3 * math.sin(a1))
        nm = (math.cos((a0 + a1) / 2), 0.5, math.sin((a0 + a1) / 2))
        o.tri([b0, b1, (1.0, 1.5, 2.0)], [nm] * 3)
        o.tri([b1, b0, (1.0, 0.85, 2.0)], [(0, -1, 0)] * 3)
    o.write(os.path.join(d, "tree.obj"), "tree.mtl")
    write_mtl(os.path.join(d, "tree.mtl"), {"trunk": {"Kd": (0.35, 0.22, 0.1)}, "crown": {"Kd": (0.1, 0.45, 0.15)}})


def trafficlight(d):
    o = Obj()
    o.usemtl("pole")
    o.box((-0.015, 0.0, -0.015), (0.015, 0.36, 0.015))
    o.box((-0.05, 0.36, -0.05), (0.05, 0.50, 0.05))
    o.usemtl("card")                                   # sorts first: mesh.textures[0] is the LED card
    s_ = 0.0505
    o.quad([(-0.04, 0.37, s_), (0.04, 0.37, s_), (0.04, 0.49, s_), (-0.04, 0.49, s_)], (0, 0, 1), [(0, 0), (1, 0), (1, 1), (0, 1)])
    o.quad([(s_, 0.37, 0.04), (s_, 0.37, -0.04), (s_, 0.49, -0.04), (s_, 0.49, 0.04)], (1, 0, 0), [(0, 0), (1, 0), (1, 1), (0, 1)])
    o.quad([(0.04, 0.37, -s_), (-0.04, 0.37, -s_), (-0.04, 0.49, -s_), (0.04, 0.49, -s_)], (0, 0, -1), [(0, 0), (1, 0), (1, 1), (0, 1)])
    o.quad([(-s_, 0.37, -0.04), (-s_, 0.37, 0.04), (-s_, 0.49, 0.04), (-s_, 0.49, -0.04)], (-1, 0, 0), [(0, 0), (1, 0), (1, 1), (0, 1)])
    o.write(os.path.join(d, "trafficlight.obj"), "trafficlight.mtl")
    write_mtl(os.path.join(d, "trafficlight.mtl"), {"pole": {"Kd": (0.2, 0.2, 0.22)},
                                                   "card": {"Kd": (1.0, 1.0, 1.0), "map_Kd": "trafficlight_card0.jpg"}})
    from PIL import Image
    for k, (top, bot) in enumerate([((230, 30, 30), (40, 60, 40)), ((60, 40, 40), (40, 220, 60))]):
        img = np.zeros((64, 64, 3), np.uint8)
        img[:] = (25, 25, 28)
        v, u = np.meshgrid(np.arange(64), np.arange(64), indexing="ij")
        img[(u - 32) ** 2 + (v - 18) ** 2 < 120] = top
        img[(u - 32) ** 2 + (v - 46) ** 2 < 120] = bot
        Image.fromarray(img).save(os.path.join(d, f"trafficlight_card{k}.jpg"), quality=95)


def duckie(d):
    """The procedural stand-in duckie (dtsim.assets.get_mesh) as OBJ / MTL, so that the reference's parser and the product's
    see the same file (the recentring quirk of objmesh.py:214-226 then applies to both)."""
    from dtsim import assets
    m = assets.get_mesh("duckie")
    o = Obj()
    mats, cur = {}, None
    for t in range(m.n_tris):
        col = tuple(round(float(c), 4) for c in m.colors[t, 0])
        name = mats.setdefault(col, f"duckie_mat{len(mats)}")
        if name != cur:
            o.usemtl(name)
            cur = name
        o.tri([tuple(m.verts[t, k]) for k in range(3)], [tuple(m.normals[t, k]) for k in range(3)])
    o.write(os.path.join(d, "duckie.obj"), "duckie.mtl")
    write_mtl(os.path.join(d, "duckie.mtl"), {name: {"Kd": col} for col, name in mats.items()})


MAP = """# test map for real-asset ingestion (MapFormat1)
tiles:
- [grass, asphalt, floor, grass, grass]
- [curve_left/W, straight/E, 3way_left/E, straight/E, curve_left/N]
- [straight/S, grass, straight/S, grass, straight/S]
- [curve_left/S, straight/E, 4way, straight/E, curve_left/E]
objects:
- {kind: cone, pos: [1.5, 1.2], rotate: 20, height: 0.1}
- {kind: sign_stop, pos: [2.95, 1.95], rotate: 90, height: 0.18}
- {kind: tree, pos: [1.5, 2.5], rotate: 0, height: 0.3, optional: true}
- {kind: duckiebot, pos: [3.5, 1.3], rotate: 180, height: 0.12, static: true, color: blue}
- {kind: duckie, pos: [0.8, 2.5], rotate: 45, height: 0.06}
- {kind: cone, pos: [4.2, 2.5], rotate: 0, scale: 0.2}
- {kind: trafficlight, pos: [2.1, 3.2], rotate: 45, height: 0.3}
tile_size: 0.585
"""


def main():
    from dtsim import assets
    meshes = os.path.join(OUT, "meshes")
    os.makedirs(meshes, exist_ok=True)
    cone(meshes); sign_generic(meshes); duckiebot(meshes); tree(meshes); trafficlight(meshes); duckie(meshes)
    for kind in ("grass", "asphalt", "floor", "straight", "curve_left", "curve_right", "3way_left", "4way"):
        tex = assets.make_texture(kind, 128)[..., :3]
        save_png(os.path.join(OUT, "textures", "tiles-processed", "photos", kind, "texture.png"), tex)
    os.makedirs(os.path.join(OUT, "maps"), exist_ok=True)
    with open(os.path.join(OUT, "maps", "test_town.yaml"), "w") as f:
        f.write(MAP)
    print("assets written to", OUT)


if __name__ == "__main__":
    main()
