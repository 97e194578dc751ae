"""`not gpu`: oracle/raster.py against frames of the REFERENCE's own render path on real OpenGL.

tests/golden/ref_gl_*.npz hold frames the unmodified reference Simulator rendered on Mesa 23.2.1 llvmpipe -- the renderer of the
reference's CI (.circleci/config.yml:10,26) -- with the state that produced them (oracle/make_gl_golden.py, oracle/gl/).  The
oracle's GL-faithful mode ("gouraud" tile lighting, llvmpipe's fixed-point GL_LINEAR) is held to them here; the HIP raster is
held to the same frames in tests/test_gpu_gl_golden.py.

Tolerance, vs Mesa 23.2.1 llvmpipe: per frame >= 99.8 % of pixels identical within +-1/255 (in fact identical: the
fraction that differs AT ALL is printed), mean abs error <= 0.02/255.  What is left are single MSAA samples at silhouettes and
tile seams: llvmpipe snaps vertices to 1/256 pixel and clips the 100 m ground quad / tiles against the frustum before
rasterising, the oracle intersects rays analytically in float64.
"""
import numpy as np
import pytest

import gl_golden as G

CASES = G.cases()


def test_goldens_are_committed():
    assert len(CASES) >= 10, CASES
    for c in CASES:
        d = G.load(c)
        assert "llvmpipe" in d["meta"]["renderer"] and d["frame"].dtype == np.uint8 and d["frame"].shape[1:] == (d["meta"]["H"], d["meta"]["W"], 3)


@pytest.mark.parametrize("case", CASES)
def test_oracle_raster_matches_reference_gl(case):
    d = G.load(case)
    n = len(d["frame"])
    ks = range(n) if d["meta"]["W"] > 160 else range(0, n, 2)          # (every second small frame: keeps the CPU suite in minutes)
    worst = dict(mean=0.0, gt1=0.0, any=0.0)
    for k in ks:
        o = G.oracle_frame(d, k, "gouraud")
        if d["meta"].get("view") == "bbox":              # the debugging view: everything but the colour of its GL_LINE_LOOPs (see gl_golden.line_mask)
            mask = G.line_mask(d, k)
            assert 0.002 < mask.mean() < 0.08, mask.mean()
            s = G.stats_masked(o, d["frame"][k], mask)
            s["any"] = float(((o != d["frame"][k]).any(axis=-1) & ~mask).mean())
            lines_gl = (d["frame"][k][..., 0].astype(int) - d["frame"][k][..., 1] > 25) & mask     # the reference did draw reddish lines there
            assert lines_gl.sum() > 20, int(lines_gl.sum())
        else:
            s = G.stats(o, d["frame"][k])
            s["any"] = float((o != d["frame"][k]).any(axis=-1).mean())
        assert s["gt1"] <= 2e-3 and s["mean"] <= 0.02, (case, k, s)
        for key in worst:
            worst[key] = max(worst[key], s[key])
    print(f"\n{case}: worst frame of {len(ks)}: differing pixels {worst['any']:.5f}, beyond +-1 {worst['gt1']:.5f}, mean abs {worst['mean']:.5f} / 255")


def test_per_fragment_tile_light_is_within_one_level_of_gl():
    """The HIP raster lights tiles per fragment; GL lights the 8 x 8 vertices of a tile and interpolates.  Measured on the GL frames:
    the oracle in "pixel" mode stays within +-1/255 of GL on >= 99.8 % of the pixels (the price of that design choice)."""
    d = G.load("small_loop_t256_160")
    for k in (0, 5, 9):
        s = G.stats(G.oracle_frame(d, k, "pixel"), d["frame"][k])
        assert s["gt1"] <= 2e-3 and s["mean"] <= 0.2, (k, s)


def test_quad_filter_distance_to_gl():
    """The product's quad-record pipeline filters with ONE byte-weight multiply-accumulate per texel and channel (oracle mode "pixel" =
    per-fragment light + raster._dtsim8_shade, the restatement of csrc/render.hip quad_filter) where llvmpipe lerps twice with an 8-bit
    intermediate.  Its distance to the GL frames, measured: a +-1/255 difference on a fraction of the textured pixels, nothing beyond that
    except the silhouette samples -- these numbers are the tolerance tests/test_gpu_gl_golden.py holds the HIP raster to."""
    for case, ks in (("small_loop_t256_160", (0, 7)), ("small_loop_dr_t256_160", (3,)), ("small_loop_t256_640", (1,))):
        d = G.load(case)
        for k in ks:
            o = G.oracle_frame(d, k, "pixel")
            s = G.stats(o, d["frame"][k])
            differ = float((o != d["frame"][k]).any(axis=-1).mean())
            print(f"\n{case}[{k}] byte-weight filter vs GL: pixels that differ {differ:.3f}, beyond +-1 {s['gt1']:.5f}, beyond +-2 {s['gt2']:.5f}, mean abs {s['mean']:.4f} / 255")
            assert s["gt1"] <= 1e-2 and s["gt2"] <= 4e-3 and s["mean"] <= 0.35, (case, k, s)


@pytest.mark.parametrize("case", ["small_loop_t256_160", "small_loop_dr_t256_160", "town_t128_320", "town_dr_t128_320", "episode2_t256_160", "small_loop_t256_640"])
def test_gl_port_reproduces_the_reference_s_frames(case):
    """oracle/gl/glport.py -- the reference's GL call stream restated so that bench.py can time it where /root/reference does not exist --
    rendered from the state of every golden record on the same driver: the frames must be BYTE-IDENTICAL to what the unmodified
    reference produced.  (Skipped where Mesa's swrast driver is missing.)"""
    from oracle.gl import glport
    if not glport.available():
        pytest.skip("no swrast_dri.so / GL headers here")
    d = G.load(case)
    scene, _md, _lib = G.scene_for(d["meta"])
    r = glport.GLRenderer(scene, d["meta"]["W"], d["meta"]["H"])
    for k in range(len(d["frame"])):
        r.set_light(d["light_eye"][k], d["light_ambient"][k], d["light_diffuse"][k])
        got = r.render(d["pos"][k], float(d["angle"][k]), cam_height=float(d["cam_height"][k]), cam_angle_deg=float(d["cam_angle"][k]),
                       cam_fov_y_deg=float(d["cam_fov_y"][k]), camera_noise=d["camera_noise"][k], domain_rand=bool(d["meta"]["dr"]),
                       horizon=d["horizon"][k], ground=d["ground"][k], obj_states=G.obj_states(d, k))
        assert np.array_equal(got, d["frame"][k]), (case, k, G.stats(got, d["frame"][k]))


@pytest.mark.parametrize("case", ["small_loop_dr_t256_160", "episode2_t256_160", "town_dr_t128_320", "view_bbox_t256_320"])
def test_committed_goldens_are_what_the_recipe_produces(case):
    """Build container only (needs /root/reference): oracle/make_gl_golden.py re-run NOW -- the unmodified reference on llvmpipe -- gives the
    committed frames and states byte for byte.  (The goldens are data produced by a committed recipe, not hand-edited; and llvmpipe with
    LP_NUM_THREADS=1 is deterministic.)"""
    from oracle.gl import refgl
    if not refgl.available():
        pytest.skip("reference tree or swrast driver not present")
    import warnings
    from oracle import make_gl_golden as MG
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fresh = MG.build(case)
    d = G.load(case)
    for key in fresh:
        if key != "meta":
            assert np.array_equal(np.asarray(fresh[key]), d[key]), (case, key)
