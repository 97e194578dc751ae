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


def view_scene(d, k):
    """(Camera, Scene, object states, lines) of a record of the "top_down" / "bbox" views (meta["view"]): the cameras through
    gym_duckietown.simulator.viewer_camera (what the facade gives this backend's camera model), the agent's own mesh as one more object in the
    top-down view (self.mesh.render(): the red duckiebot, unscaled), the collision rectangles as line segments in the bbox view."""
    import copy, math
    from gym_duckietown.simulator import agent_bbox_angle, viewer_camera
    m = d["meta"]
    scene, md, lib = scene_for(m)
    om = scene.m
    st = obj_states(d, k)
    top = m["view"] == "top_down"
    vp, va, vh, vdeg = viewer_camera(top, not top, d["pos"][k], float(d["angle"][k]), om.grid_width, om.grid_height, om.tile_size, float(d["cam_fov_y"][k]))
    cam = raster.Camera(vp, va, cam_height=vh, cam_angle_deg=vdeg, cam_fov_y_deg=float(d["cam_fov_y"][k]), width=int(m["W"]), height=int(m["H"]),
                        horizon_color=list(d["horizon"][k]), ground_color=list(d["ground"][k]), light_pos=list(d["light_eye"][k]),
                        light_ambient=list(d["light_ambient"][k]), light_diffuse=list(d["light_diffuse"][k]))
    lines = None
    if top:
        key = ("view", m["map_name"], m["tree"])
        if key not in _scenes:
            md2 = copy.deepcopy(md)
            md2["objects"] = list(md2["objects"]) + [{"kind": "duckiebot", "pos": [0.5, 0.5], "rotate": 0, "static": False, "scale": 1.0, "color": "red"}]
            meshes = dict(scene.meshes)
            meshes["duckiebot"] = lib.object_mesh(md2["objects"][-1])[1]
            om2 = osim.OracleMap(md2, {kk: (mm.min_coords, mm.max_coords) for kk, mm in meshes.items()})
            _scenes[key] = raster.Scene(om2, scene.textures, meshes)
        scene = _scenes[key]
        st = st + [dict(pos=d["pos"][k], y_rot=math.degrees(float(d["angle"][k])), visible=True)]
    else:
        segs = []
        for o, s_ in zip(om.objects, st):
            if s_["visible"]:
                c = o.obj_corners                                            # [4, 2]
                segs += [[c[i][0], 0.01, c[i][1], c[(i + 1) % 4][0], 0.01, c[(i + 1) % 4][1], 1.0, 0.0, 0.0] for i in range(4)]
        c = osim.get_agent_corners(d["pos"][k], agent_bbox_angle(om.grid, om.grid_width, om.grid_height, float(d["angle"][k])))
        segs += [[c[i][0], 0.01, c[i][1], c[(i + 1) % 4][0], 0.01, c[(i + 1) % 4][1], 1.0, 0.0, 0.0] for i in range(4)]
        lines = np.asarray(segs, dtype=np.float64)
    return cam, scene, st, lines


def oracle_frame(d, k, lighting="gouraud", with_lines=True):
    if d["meta"].get("view"):
        cam, scene, st, lines = view_scene(d, k)
        return raster.render_obs(cam, scene, lighting, obj_states=st, lines=lines if with_lines else None)
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


def line_mask(d, k, grow=1):
    """Pixels of a "bbox" record that the GL_LINE_LOOPs may touch (the oracle with its lines against the oracle without, grown by `grow` pixels):
    the reference leaves texturing and lighting on while it draws them, so their COLOUR is not something dtsim
    reproduces; everything else in the view is compared."""
    a, b = oracle_frame(d, k, "gouraud", True), oracle_frame(d, k, "gouraud", False)
    m = (a != b).any(axis=-1)
    for _ in range(grow):
        g = m.copy()
        g[1:] |= m[:-1]; g[:-1] |= m[1:]; g[:, 1:] |= m[:, :-1]; g[:, :-1] |= m[:, 1:]
        m = g
    return m


def stats_masked(a, b, mask):
    keep = ~mask
    e = np.abs(a.astype(np.int32) - b.astype(np.int32))[keep]
    m = e.max(axis=-1)
    return dict(mean=float(e.mean()), gt1=float((m > 1).mean()), gt2=float((m > 2).mean()), gt8=float((m > 8).mean()), max=int(m.max()))
