"""Shared by tests/test_gl_golden.py (CPU) and tests/test_gpu_gl_golden.py: load a GL golden record (tests/golden/ref_gl_*.npz,
made by oracle/make_gl_golden.py from the reference's own render path on Mesa llvmpipe) and rebuild, from the state it carries,
the oracle's Camera / Scene / object states -- so that oracle/raster.py (and the HIP raster) can be held against real GL frames."""
from __future__ import annotations

import glob
import json
import os

import numpy as np

from dtsim import assets
from oracle import raster, sim as osim
from oracle.gl import asset_trees

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def cases():
    return sorted(os.path.basename(p)[len("ref_gl_"):-len(".npz")] for p in glob.glob(os.path.join(GOLDEN, "ref_gl_*.npz")))


def load(case):
    z = np.load(os.path.join(GOLDEN, f"ref_gl_{case}.npz"))
    d = {k: z[k] for k in z.files}
    d["meta"] = json.loads(str(d["meta"]))
    return d


_scenes = {}


def scene_for(meta):
    """(Scene, map data, AssetLibrary) of a golden case: the same files the reference read."""
    key = (meta["map_name"], meta["tree"])
    if key not in _scenes:
        lib = assets.AssetLibrary(asset_trees.tree(meta["tree"]))
        md = lib.map_data(meta["map_name"])
        meshes = {"*": assets.get_mesh("*")}
        for desc in md["objects"]:
            meshes[desc["kind"]] = lib.object_mesh(desc)[1]
        ext = {k: (m.min_coords, m.max_coords) for k, m in meshes.items()}
        om = osim.OracleMap(md, ext)
        kinds = {t["kind"] for t in om.grid if t is not None}
        scene = raster.Scene(om, {k: lib.tile_texture(k) for k in kinds}, meshes)
        scene.light_cards = lib.light_cards()
        _scenes[key] = (scene, md, lib)
    return _scenes[key]


def camera(d, k):
    m = d["meta"]
    return raster.Camera(d["pos"][k], float(d["angle"][k]), cam_height=float(d["cam_height"][k]), cam_angle_deg=float(d["cam_angle"][k]),
                         cam_fov_y_deg=float(d["cam_fov_y"][k]), camera_noise=list(d["camera_noise"][k]), domain_rand=bool(m["dr"]),
                         horizon_color=list(d["horizon"][k]), ground_color=list(d["ground"][k]), light_pos=list(d["light_eye"][k]),
                         light_ambient=list(d["light_ambient"][k]), light_diffuse=list(d["light_diffuse"][k]), width=int(m["W"]), height=int(m["H"]))


def obj_states(d, k):
    return [dict(pos=d["obj_pos"][k][i], y_rot=float(d["obj_yrot"][k][i]), visible=bool(d["obj_visible"][k][i]),
                 light_pattern=int(d["obj_pattern"][k][i])) for i in range(d["obj_pos"].shape[1])]


def oracle_frame(d, k, lighting="gouraud"):
    scene, _md, lib = scene_for(d["meta"])
    cam = camera(d, k)
    if d["meta"].get("segment"):                      # render_obs(segment=True): textures through load_texture(segment=True), lighting off, magenta clear / ground
        seg_tex = {kind: assets.segment_texture(t, lib.tile_texture_file(kind)) for kind, t in scene.textures.items()}
        cam, scene = raster.segment_view(cam, scene, seg_tex, {key: (0, 0, 0) for key in scene.meshes})
    return raster.render_obs(cam, scene, lighting, obj_states=obj_states(d, k))


def stats(a, b):
    e = np.abs(a.astype(np.int32) - b.astype(np.int32))
    m = e.max(axis=-1)
    return dict(mean=float(e.mean()), gt1=float((m > 1).mean()), gt2=float((m > 2).mean()), gt8=float((m > 8).mean()), max=int(m.max()))
