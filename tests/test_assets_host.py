"""Real-asset ingestion, host side: OBJ/MTL parser pinned against the reference's
own ObjMesh (golden recorded by oracle/make_golden.py; live comparison when /root/reference is
present), asset resolution by basename, texture preparation, MapFormat1 YAML -> tables."""
import os

import numpy as np
import pytest

from dtsim import assets, maps, objmesh

HERE = os.path.dirname(os.path.abspath(__file__))
ASSETS = os.path.join(HERE, "golden", "assets")
CASES = [
    ("cone", "cone", None),
    ("sign_stop", "sign_generic", {"April_Tag": {"map_Kd": "sign_stop.png"}}),
    ("tree", "tree", None),
    ("duckiebot_blue", "duckiebot", {"gkmodel0_chassis_geom0_mat_001-material": {"Kd": np.array([0.0, 0.0, 1.0])},
                                    "gkmodel0_chassis_geom0_mat_001-material.001": {"Kd": np.array([0.0, 0.0, 1.0])}}),
]


def _chunk_textures(m, chunk_sizes):
    out, i = [], 0
    for n in chunk_sizes:
        t = int(m.tri_tex[i])
        assert (m.tri_tex[i:i + n] == t).all()
        out.append("" if t < 0 else os.path.basename(m.texture_files[t]))
        i += n
    return out


@pytest.mark.parametrize("tag,stem,change", CASES)
def test_obj_parser_matches_reference_golden(tag, stem, change):
    lib = assets.AssetLibrary(ASSETS)
    g = np.load(os.path.join(HERE, "golden", "ref_objmesh.npz"))
    m = objmesh.load_obj(lib.resolve(stem + ".obj"), lib.resolve, change_materials=change)
    for k, mine in (("verts", m.verts), ("uvs", m.uvs), ("normals", m.normals), ("colors", m.colors),
                    ("min_coords", m.min_coords), ("max_coords", m.max_coords)):
        assert np.array_equal(g[f"{tag}_{k}"], mine), k          # bit-exact float32
    assert _chunk_textures(m, g[f"{tag}_chunk_sizes"]) == [str(t) for t in g[f"{tag}_textures"]]


def test_obj_parser_matches_reference_live():
    from oracle import refstub
    if not refstub.available():
        pytest.skip("reference tree not present")
    lib = assets.AssetLibrary(ASSETS)
    for tag, stem, change in CASES:
        r = refstub.ref_objmesh(lib.resolve(stem + ".obj"), stem, lib.resolve, change)
        m = objmesh.load_obj(lib.resolve(stem + ".obj"), lib.resolve, change_materials=change)
        assert np.array_equal(r["verts"], m.verts) and np.array_equal(r["colors"], m.colors)
        assert np.array_equal(r["min_coords"], m.min_coords) and np.array_equal(r["max_coords"], m.max_coords)


def test_recentring_quirk_is_reproduced():
    # objmesh.py:217 centres on (min + max(axis=0).min(axis=0)) / 2, not on the true mid-point
    m = assets.AssetLibrary(ASSETS).mesh("tree")
    assert m.min_coords[1] == 0.0
    assert abs(m.min_coords[0] + m.max_coords[0]) > 0.05


def test_library_resolution_and_fallback():
    lib = assets.AssetLibrary(ASSETS)
    assert lib.resolve("cone.obj").endswith(os.path.join("meshes", "cone.obj"))
    assert lib.resolve("nope.obj") is None
    assert lib.tile_texture_file("4way").endswith(os.path.join("photos", "4way", "texture.png"))
    assert lib.tile_texture("grass").shape == (128, 128, 4)
    key, m = lib.object_mesh({"kind": "sign_stop"})
    assert key == "sign_stop" and [os.path.basename(t) for t in m.texture_files] == ["sign_stop.png"]
    key, m = lib.object_mesh({"kind": "duckiebot", "color": "blue"})
    assert key == "duckiebot:blue" and (m.colors == np.array([0, 0, 1], np.float32)).all(-1).any()   # recoloured chassis
    key, m = lib.object_mesh({"kind": "duckie"})                    # no duckie.obj in the tree: stand-in
    assert key == "duckie" and m.n_tris == assets.get_mesh("duckie").n_tris
    none = assets.AssetLibrary(None)
    assert none.root is None and none.tile_texture("grass").shape == (256, 256, 4)
    assert none.object_mesh({"kind": "cone"})[0] == "*"


def test_map_yaml_to_tables():
    lib = assets.AssetLibrary(ASSETS)
    md = lib.map_data("test_town")
    meshes = {"duckie": assets.get_mesh("duckie"), "*": assets.get_mesh("*")}
    mt = maps.interpret_map(md, "test_town", meshes, library=lib)
    assert (mt.grid_w, mt.grid_h) == (5, 4) and len(mt.objects) == 7
    assert [o.mesh_kind for o in mt.objects] == ["cone", "sign_stop", "tree", "duckiebot:blue", "duckie", "cone", "trafficlight"]
    tl = mt.objects[6]
    assert (tl.light_freq, tl.light_pattern, tl.light_tris) == (5, 0, 8) and not tl.collidable     # objects.py:446-451, simulator.py:1027
    cone = lib.mesh("cone")
    assert np.isclose(mt.objects[0].scale, 0.1 / float(cone.max_coords[1])) and mt.objects[5].scale == 0.2
    assert mt.objects[2].optional and all(o.static for o in mt.objects)
    assert [c.shape for c in lib.light_cards()] == [(64, 64, 4)] * 2 and assets.AssetLibrary(None).light_cards() is None
    # the same YAML through a file path
    md2 = assets.get_map(os.path.join(ASSETS, "maps", "test_town.yaml"))
    assert md2 == md


def test_non_power_of_two_textures_are_resampled():
    t = np.zeros((100, 60, 4), np.uint8)
    t[..., 0] = np.arange(60)[None, :]
    r = assets.to_pow2(t)
    assert r.shape == (128, 64, 4)
    assert assets.to_pow2(r) is not None and assets.to_pow2(r).shape == r.shape
    assert assets.to_pow2(t, 256).shape == (256, 256, 4)


def test_obj_parser_fuzz_against_reference(tmp_path):
    """Random OBJ/MTL files (random material order, v/t/n and v//n faces, missing / unknown materials, an
    optional <stem>.png default texture) through both parsers: bit-identical arrays, extents and per-chunk
    textures.  Needs the reference tree (build container only)."""
    from oracle import refstub
    if not refstub.available():
        pytest.skip("reference tree not present")
    from PIL import Image
    rng = np.random.default_rng(123)
    for case in range(12):
        d = tmp_path / f"c{case}"
        d.mkdir()
        stem = f"mesh{case}"
        mats = [f"m{k}_{rng.integers(0, 99)}" for k in range(int(rng.integers(0, 4)))]
        with open(d / f"{stem}.mtl", "w") as f:
            for m in mats:
                f.write(f"newmtl {m}\nKd {rng.random():.4f} {rng.random():.4f} {rng.random():.4f}\n")
                if rng.random() < 0.5:
                    Image.fromarray(rng.integers(0, 255, (8, 8, 3), dtype=np.uint8)).save(d / f"{m}.png")
                    f.write(f"map_Kd  {m}.png\n")
        if case % 3 == 0:
            os.remove(d / f"{stem}.mtl")                       # no material library at all
        if case % 4 == 1:
            Image.fromarray(rng.integers(0, 255, (8, 8, 3), dtype=np.uint8)).save(d / f"{stem}.png")   # default-material texture
        nv, nt, nn = int(rng.integers(4, 12)), int(rng.integers(1, 6)), int(rng.integers(1, 5))
        with open(d / f"{stem}.obj", "w") as f:
            f.write("# fuzz\n\no thing\n")
            for _ in range(nv):
                f.write("v  %.5f %.5f  %.5f\n" % tuple(rng.uniform(-2, 3, 3)))
            for _ in range(nt):
                f.write("vt %.5f %.5f\n" % tuple(rng.uniform(0, 1, 2)))
            for _ in range(nn):
                f.write("vn %.5f %.5f %.5f\n" % tuple(rng.uniform(-1, 1, 3)))
            for _ in range(int(rng.integers(2, 14))):
                if rng.random() < 0.4:
                    f.write(f"usemtl {rng.choice(mats + ['nosuchmtl']) if mats else 'nosuchmtl'}\n")
                with_t = rng.random() < 0.6
                toks = []
                for _ in range(3):
                    v, t, n = int(rng.integers(1, nv + 1)), int(rng.integers(1, nt + 1)), int(rng.integers(1, nn + 1))
                    toks.append(f"{v}/{t}/{n}" if with_t else f"{v}//{n}")
                f.write("f " + " ".join(toks) + " \n")

        def resolve(bn, _d=d):
            p = os.path.join(_d, bn)
            return p if os.path.isfile(p) else None

        p = str(d / f"{stem}.obj")
        r = refstub.ref_objmesh(p, stem, resolve, None)
        m = objmesh.load_obj(p, resolve)
        for k, mine in (("verts", m.verts), ("uvs", m.uvs), ("normals", m.normals), ("colors", m.colors),
                        ("min_coords", m.min_coords), ("max_coords", m.max_coords)):
            assert np.array_equal(r[k], mine), (case, k)
        assert list(r["chunk_sizes"]) == m.chunk_sizes, case
        want = ["" if t is None else os.path.basename(t) for t in r["textures"]]
        assert _chunk_textures(m, r["chunk_sizes"]) == want, case


def test_segmentation_assets_follow_the_reference_rules():
    """graphics.py:59-126 / objmesh.py:260-266: which textures are blanked, which keep their paint, the name hash."""
    from dtsim import assets as A
    # gen_segmentation_color: decimal char codes, 3-digit groups, % 255 (checked by hand for "duckie":
    # "100" "117" "991" -> 100, 117, 991 % 255 = 226)
    assert A.gen_segmentation_color("duckie") == [100, 117, 226]
    assert A.gen_segmentation_color("duckiebot") == [100, 117, 226]          # same first three groups (reference quirk)
    with pytest.raises(ValueError):
        A.gen_segmentation_color("*")
    assert A.should_segment_out("tiles-processed/photos/asphalt/texture.jpg")
    assert A.should_segment_out("tiles-processed/photos/grass/texture.jpg")
    assert A.should_segment_out("sign_left_T_intersect.png")                    # "sign" wins over "left"
    assert not A.should_segment_out("tiles-processed/photos/curve_left/texture.jpg")
    assert not A.should_segment_out("tiles-processed/photos/4way/texture.jpg")
    # 8-bit BGR<->HSV known values (OpenCV documentation: H in [0,180), S, V in [0,255])
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 0, 0], [128, 128, 128], [0, 255, 255]]], np.uint8)
    hsv = A.bgr2hsv_u8(px)
    assert hsv.tolist() == [[[120, 255, 255], [60, 255, 255], [0, 255, 255], [0, 0, 255], [0, 0, 0], [0, 0, 128], [30, 255, 255]]]
    assert np.array_equal(A.hsv2bgr_u8(hsv), px)
    rng = np.random.default_rng(0)
    rnd = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    back = A.hsv2bgr_u8(A.bgr2hsv_u8(rnd))
    assert np.abs(back.astype(int) - rnd.astype(int)).max() <= 6                # 8-bit HSV is lossy, but only slightly
    # a lane-marking tile: dark asphalt goes black, yellow / white paint survives except its 1-px rim
    tex = np.zeros((32, 32, 4), np.uint8)
    tex[..., :3] = (60, 60, 64)
    tex[..., 3] = 255
    tex[:, 10:16, :3] = (240, 200, 40)                                          # yellow line, 6 px wide
    tex[:, 24:28, :3] = (235, 235, 230)                                         # white line, 4 px wide
    seg = A.segment_texture(tex, "tiles-processed/photos/straight/texture.png")
    assert (seg[:, :10, :3] == 0).all() and (seg[:, 16:24, :3] == 0).all() and (seg[:, 28:, :3] == 0).all()
    assert (seg[:, 10, :3] == 0).all() and (seg[:, 15, :3] == 0).all()         # rim eroded by the 8-neighbour kernel
    assert np.abs(seg[:, 11:15, :3].astype(int) - (240, 200, 40)).max() <= 4
    assert np.abs(seg[:, 25:27, :3].astype(int) - (235, 235, 230)).max() <= 4
    flat = A.segment_texture(tex, "tiles-processed/photos/grass/texture.png")
    assert (flat[..., :3] == 0).all() and (flat[..., 3] == 255).all()
    col = A.segment_texture(tex, "duckie.png", [100, 117, 226])
    assert (col[..., :3] == (100, 117, 226)).all()


def test_map_list_of_randomize_maps_on_reset_is_never_cut_silently(tmp_path):
    """simulator.py:373-378: randomize_maps_on_reset draws from every map file but calibration* / regress*.  The library keeps
    DTSIM_MAX_MAPS (32) maps resident: a tree with more map files is an error that names the way out, not a shorter list."""
    from dtsim import _ffi
    from gym_duckietown.simulator import Simulator
    d = tmp_path / "maps"
    d.mkdir()
    for i in range(_ffi.MAX_MAPS):
        (d / f"map_{i:02d}.yaml").write_text("tiles: []\n")
    (d / "calibration_map.yaml").write_text("tiles: []\n")
    (d / "regress_4way.yaml").write_text("tiles: []\n")
    names = Simulator._all_map_names(str(tmp_path))
    assert len(names) == _ffi.MAX_MAPS == 32 and names[0] == "map_00" and "calibration_map" not in names
    (d / "one_more.yaml").write_text("tiles: []\n")
    with pytest.raises(ValueError, match="DTSIM_MAX_MAPS"):
        Simulator._all_map_names(str(tmp_path))
