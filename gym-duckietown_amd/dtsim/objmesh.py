"""Wavefront OBJ / MTL ingestion with the reference's ObjMesh semantics (objmesh.py:55-358).

Host-side, one-time asset prep: produces the triangle soup the hot path consumes
(`dtsim_mesh`: float32 positions / normals / per-vertex Kd / texture coordinates, a texture
index per triangle) plus `min_coords` / `max_coords`, which feed the collision boxes
(objects.py:47-63).  What is reproduced, quirks included:

  * materials (objmesh.py:295-358): a default material "" with Kd = (1,1,1) and, when
    `<stem>.png` resolves, that image as its texture; `<stem>.mtl` (resolved by basename)
    contributes `newmtl` / `Kd` / `map_Kd` (the texture is later looked up by *basename*,
    objmesh.py:273-274);
  * faces (objmesh.py:105-158): triangles only, `v/t/n` or `v//n` (no texcoord -> (0,0)),
    `usemtl` of an unknown material falls back to "", then a *stable sort by material name*
    decides the draw order (objmesh.py:161) -- the z-buffer tie break depends on it;
  * recentring (objmesh.py:214-226): x and z are centred on `(min + m) / 2` where `m` is
    `verts.max(axis=0).min(axis=0)` -- the smallest of the three per-corner maxima, not the
    true maximum; y is shifted so the base sits at 0.  All of it in float32.

A material without `Kd` raises KeyError in the reference; here it renders white.
"""
from __future__ import annotations

import os
from typing import Callable, Dict, List, Optional

import numpy as np


class MeshData:
    """What ObjMesh exposes to the hot path: float32 triangle soup [T,3,3] verts / normals /
    per-vertex Kd colours, [T,3,2] texture coordinates, a per-triangle index into
    `texture_files` (-1 = untextured), and min_coords / max_coords (objmesh.py:230-232)."""

    def __init__(self, verts, normals, colors, uvs=None, tri_tex=None, texture_files=None, name="mesh", chunk_sizes=None):
        self.name = name
        self.verts = np.ascontiguousarray(verts, dtype=np.float32)
        self.normals = np.ascontiguousarray(normals, dtype=np.float32)
        self.colors = np.ascontiguousarray(colors, dtype=np.float32)
        T = self.verts.shape[0]
        self.uvs = np.zeros((T, 3, 2), np.float32) if uvs is None else np.ascontiguousarray(uvs, dtype=np.float32)
        self.tri_tex = np.full(T, -1, np.int32) if tri_tex is None else np.ascontiguousarray(tri_tex, dtype=np.int32)
        self.texture_files: List[str] = list(texture_files or [])
        self.textures: List[np.ndarray] = []          # RGBA8, GL row order; filled by the asset library
        self.chunk_sizes: List[int] = list(chunk_sizes) if chunk_sizes is not None else [T]   # triangles per material chunk, draw order
        self.min_coords = self.verts.min(axis=0).min(axis=0)
        self.max_coords = self.verts.max(axis=0).max(axis=0)

    @property
    def n_tris(self):
        return self.verts.shape[0]


def _tokens(line: str) -> List[str]:
    return [t for t in (tok.strip(" ") for tok in line.split(" ")) if t != ""]


def _lines(path: str):
    with open(path, "r") as f:
        for raw in f:
            line = raw.rstrip(" \r\n")
            if line == "" or line.startswith("#"):
                continue
            tk = _tokens(line)
            if tk:
                yield tk[0], tk[1:]


def load_materials(obj_path: str, resolve: Callable[[str], Optional[str]]) -> Dict[str, dict]:
    """objmesh.py:295-358.  `resolve(basename)` -> path or None (get_resource_path)."""
    model_dir, file_name = os.path.split(obj_path)
    stem = file_name.split(".")[0]
    default = {"Kd": np.array([1, 1, 1])}
    png = resolve(f"{stem}.png")
    if png is not None:
        default["map_Kd"] = png
    mats: Dict[str, dict] = {"": default}
    mtl_path = resolve(f"{stem}.mtl")
    if mtl_path is None:
        return mats
    cur = None
    for key, args in _lines(mtl_path):
        if key == "newmtl":
            cur = {}
            mats[args[0]] = cur
        elif key == "Kd":
            cur["Kd"] = np.array([float(v) for v in args])
        elif key == "map_Kd":
            cur["map_Kd"] = os.path.join(model_dir, args[-1])
    return mats


def load_obj(obj_path: str, resolve: Optional[Callable[[str], Optional[str]]] = None, name: Optional[str] = None,
             change_materials: Optional[Dict[str, dict]] = None) -> MeshData:
    """Parse `obj_path` the way ObjMesh.__init__ does (objmesh.py:55-232).  `change_materials`
    updates named materials after the MTL is read (objmesh.py:96-103: duckiebot chassis colour,
    sign textures)."""
    if resolve is None:
        base = os.path.dirname(obj_path)

        def resolve(bn, _base=base):
            p = os.path.join(_base, bn)
            return p if os.path.isfile(p) else None

    mats = load_materials(obj_path, resolve)
    for mname, upd in (change_materials or {}).items():
        if mname in mats:
            mats[mname].update(upd)
    pos, tcs, nrm = [], [], []
    faces = []                                   # (corner index triples, material name)
    cur = ""
    for key, args in _lines(obj_path):
        if key == "v":
            pos.append([float(v) for v in args])
        elif key == "vt":
            tcs.append([float(v) for v in args])
        elif key == "vn":
            nrm.append([float(v) for v in args])
        elif key == "usemtl":
            cur = args[0] if args[0] in mats else ""
        elif key == "f":
            if len(args) != 3:
                raise ValueError(f"{obj_path}: only triangle faces are supported")
            corners = []
            for tok in args:
                idx = [int(t) for t in tok.split("/") if t != ""]
                if len(idx) not in (2, 3):
                    raise ValueError(f"{obj_path}: face corner {tok!r} needs v/t/n or v//n")
                corners.append(idx)
            faces.append((corners, cur))
    if not faces:
        raise ValueError(f"{obj_path}: no faces")
    faces.sort(key=lambda f: f[1])               # stable: draw order = material name, then file order

    F = len(faces)
    V = np.zeros((F, 3, 3), np.float32)
    N = np.zeros((F, 3, 3), np.float32)
    T = np.zeros((F, 3, 2), np.float32)
    Cc = np.zeros((F, 3, 3), np.float32)
    tex_files: List[str] = []
    tri_tex = np.full(F, -1, np.int32)
    for fi, (corners, mname) in enumerate(faces):
        m = mats[mname]
        kd = m.get("Kd", np.array([1, 1, 1]))
        if "map_Kd" in m:
            tf = resolve(os.path.basename(m["map_Kd"]))        # objmesh.py:273-274
            if tf is not None:
                if tf not in tex_files:
                    tex_files.append(tf)
                tri_tex[fi] = tex_files.index(tf)
        for ci, idx in enumerate(corners):
            if len(idx) == 3:
                V[fi, ci] = pos[idx[0] - 1]; T[fi, ci] = tcs[idx[1] - 1][:2]; N[fi, ci] = nrm[idx[2] - 1]
            else:
                V[fi, ci] = pos[idx[0] - 1]; N[fi, ci] = nrm[idx[1] - 1]
            Cc[fi, ci] = kd[:3]
    lo = V.min(axis=0).min(axis=0)
    hi_q = V.max(axis=0).min(axis=0)             # sic (objmesh.py:217)
    mid = (lo + hi_q) / 2
    V[:, :, 1] -= lo[1]
    V[:, :, 0] -= mid[0]
    V[:, :, 2] -= mid[2]
    sizes, prev = [], None
    for _corners, mname in faces:                 # chunks = runs of equal material name (objmesh.py:163-173)
        if mname != prev:
            sizes.append(0); prev = mname
        sizes[-1] += 1
    return MeshData(V, N, Cc, T, tri_tex, tex_files, name=name or os.path.basename(obj_path).split(".")[0], chunk_sizes=sizes)
