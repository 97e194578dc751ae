"""Generate tests/golden/ from the REFERENCE'S OWN CODE (run in the build container only).

    python -m oracle.make_golden

Imports /root/reference/src/gym_duckietown through oracle/refstub.py (third-party modules
stubbed) and records what the reference computes, so that the fixtures can
travel to the GPU box where /root/reference does not exist:

  ref_probes_<map>.npz   random poses -> get_grid_coords, _drivable_pos, _collision,
                         _valid_pose (sf 1.0 / 1.3), proximity_penalty2, get_lane_pos2,
                         compute_reward, _compute_done_reward  (unmodified reference code)
  ref_probes_junctions.npz  the same on oracle/fixtures.py:junction_map (every tile kind in every orientation)
  ref_maps.npz           _interpret_map tables: curves, collidable corners/norms/centres/radii
  ref_resets.json        Simulator.reset() results per (map, domain_rand, seed): pose + DR values
  ref_duckie_walk.npz    DuckieObj.step trajectory (centre, active flag, y_rot) over 700 steps
  ref_kat.json           SURVEY Appendix A known-answer vectors, re-derived from the reference
  ref_distortion.npz     Distortion._invert_map/_fill_holes (reference code) applied to the
                         restated OpenCV rectify maps: rounded source pixel per output pixel
Inputs (maps, stand-in mesh extents) come from gym-duckietown_amd/dtsim/assets.py.
"""
from __future__ import annotations

import copy
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gym-duckietown_amd"))

from dtsim import assets  # noqa: E402
from oracle import refstub  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
EXT = assets.mesh_extents(("duckie",))
MAPS = ["small_loop", "small_loop_only_duckies", "loop_only_duckies"]


# (tag, obj stem, change_materials) -- the variants simulator.py:958-974 / 2091-2099 build
OBJ_CASES = [
    ("cone", "cone", None),
    ("sign_stop", "sign_generic", {"April_Tag": {"map_Kd": "sign_stop.png"}}),
    ("tree", "tree", None),
    ("duckiebot_blue", "duckiebot", {"gkmodel0_chassis_geom0_mat_001-material": {"Kd": np.array([0.0, 0.0, 1.0])},
                                    "gkmodel0_chassis_geom0_mat_001-material.001": {"Kd": np.array([0.0, 0.0, 1.0])}}),
]


def ref_sim(map_name, domain_rand=False, seed=None, md=None):
    md = md if md is not None else assets.get_map(map_name)
    sim, ns = refstub.make_simulator(copy.deepcopy(md), domain_rand=domain_rand, mesh_extents=EXT)
    sim.randomize_maps_on_reset = False
    sim.randomizer = ns.randomizer.Randomizer()
    sim.color_sky = list(ns.simulator.BLUE_SKY)
    sim.color_ground = (0.15, 0.15, 0.15)
    sim.camera_rand = False
    sim.num_tris_distractors = 12
    sim.user_tile_start = None
    sim.accept_start_angle_deg = 60
    sim.dynamics_rand = False
    sim.style = "photos"
    sim.graphics = False
    sim.distortion = False
    sim.undistort = False
    sim.camera_width, sim.camera_height = 640, 480
    sim.multi_fbo = sim.final_fbo = sim.img_array = None
    sim.np_random = np.random.default_rng(seed)
    return sim, ns


def probes(map_name, n=1500, seed=7, md=None):
    r, ns = ref_sim(map_name, md=md)
    rng = np.random.default_rng(seed)
    W, H, TS = r.grid_width, r.grid_height, r.road_tile_size
    poses = np.stack([rng.uniform(-0.2, W * TS + 0.2, n), rng.uniform(-0.2, H * TS + 0.2, n), rng.uniform(-4, 7, n)], 1)
    if len(r.collidable_centers):
        k = n // 3
        c = r.collidable_centers[rng.integers(0, len(r.collidable_centers), k)]
        poses[:k, 0] = c[:, 0] + rng.uniform(-0.3, 0.3, k)
        poses[:k, 1] = c[:, 2] + rng.uniform(-0.3, 0.3, k)
    out = dict(poses=poses, tile=np.zeros((n, 2), np.int64), drivable=np.zeros(n, bool), collision=np.zeros(n, bool),
               valid=np.zeros(n, bool), valid13=np.zeros(n, bool), prox=np.zeros(n), in_lane=np.zeros(n, bool),
               lane=np.zeros((n, 4)), reward=np.zeros(n), done=np.zeros(n, bool), done_reward=np.zeros(n))
    for q, (x, z, a) in enumerate(poses):
        pos = np.array([x, 0, z])
        out["tile"][q] = r.get_grid_coords(pos)
        out["drivable"][q] = r._drivable_pos(pos)
        out["collision"][q] = r._collision(ns.simulator.get_agent_corners(pos, a))
        out["valid"][q] = r._valid_pose(pos, a)
        out["valid13"][q] = r._valid_pose(pos, a, 1.3)
        out["prox"][q] = r.proximity_penalty2(pos, a)
        try:
            out["lane"][q] = [float(v) for v in r.get_lane_pos2(pos, a)]
            out["in_lane"][q] = True
        except ns.simulator.NotInLane:
            pass
        out["reward"][q] = r.compute_reward(pos, a, r.robot_speed)
        r.cur_pos, r.cur_angle, r.step_count = pos, a, 5
        d = r._compute_done_reward()
        out["done"][q], out["done_reward"][q] = d.done, d.reward
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    assert refstub.available(), "run in the build container (needs /root/reference)"
    maps_out = {}
    for m in MAPS:
        np.savez_compressed(os.path.join(OUT, f"ref_probes_{m}.npz"), **probes(m))
        r, _ = ref_sim(m)
        curves = np.concatenate([t["curves"] for t in r.drivable_tiles], axis=0)
        maps_out[f"{m}_curves"] = curves
        if len(r.collidable_centers):
            maps_out[f"{m}_corners"] = r.collidable_corners
            maps_out[f"{m}_norms"] = r.collidable_norms
            maps_out[f"{m}_centers"] = r.collidable_centers
            maps_out[f"{m}_radii"] = r.collidable_safety_radii
    np.savez_compressed(os.path.join(OUT, "ref_maps.npz"), **maps_out)
    from oracle.fixtures import junction_map           # every tile kind x orientation (3-way / 4-way curve tables)
    np.savez_compressed(os.path.join(OUT, "ref_probes_junctions.npz"), **probes("junctions", n=3000, seed=9, md=junction_map()))

    resets = []
    for m in MAPS:
        for dr in (False, True):
            for seed in range(6):
                r, _ = ref_sim(m, dr, seed)
                rec = dict(map=m, domain_rand=dr, seed=seed, resets=[])
                for _ in range(2):
                    r.reset()
                    rec["resets"].append(dict(
                        pos=[float(v) for v in r.cur_pos], angle=float(r.cur_angle),
                        horizon=[float(v) for v in r.horizon_color], ground=[float(v) for v in r.ground_color],
                        wheel_dist=float(r.wheel_dist), cam_height=float(np.asarray(r.cam_height).reshape(-1)[0]),
                        cam_angle=float(np.asarray(r.cam_angle[0]).reshape(-1)[0]),
                        cam_fov_y=float(np.asarray(r.cam_fov_y).reshape(-1)[0]),
                        trim=float(r.randomization_settings["trim"][0]),
                        camera_noise=[float(v) for v in r.randomization_settings["camera_noise"]],
                        light_pos=[float(v) for v in r.randomization_settings["light_pos"]]))
                resets.append(rec)
    json.dump(resets, open(os.path.join(OUT, "ref_resets.json"), "w"))

    md = assets.get_map("loop_pedestrians")
    r, _ = ref_sim("loop_pedestrians", md=md)
    for ob in r.objects:
        ob.wiggle = np.pi / 15
    T = 700
    cen = np.zeros((T, len(r.objects), 2)); act = np.zeros((T, len(r.objects)), bool); yrot = np.zeros((T, len(r.objects)))
    cor = np.zeros((T, len(r.objects), 4, 2))
    for t in range(T):
        for k, ob in enumerate(r.objects):
            ob.step(1 / 30)
            c = np.asarray(ob.center, dtype=float)
            cen[t, k] = c[[0, 2]]; act[t, k] = ob.pedestrian_active; yrot[t, k] = ob.y_rot; cor[t, k] = ob.obj_corners
    np.savez_compressed(os.path.join(OUT, "ref_duckie_walk.npz"), center=cen, active=act, y_rot=yrot, corners=cor)

    # DuckiebotObj followers (objects.py:180-336) on the loop_dyn_duckiebots fixture
    md = assets.get_map("loop_dyn_duckiebots")
    r, _ = ref_sim("loop_dyn_duckiebots", md=md)
    bots = [ob for ob in r.objects if ob.kind == "duckiebot"]
    T = 600
    bpos = np.zeros((T, len(bots), 2)); bang = np.zeros((T, len(bots))); bcor = np.zeros((T, len(bots), 4, 2))
    for t in range(T):
        for k, ob in enumerate(bots):
            ob.step_duckiebot(1 / 30, r.closest_curve_point, [])
            bpos[t, k] = np.asarray(ob.pos, dtype=float)[[0, 2]]; bang[t, k] = ob.angle; bcor[t, k] = ob.obj_corners
    np.savez_compressed(os.path.join(OUT, "ref_duckiebot_drive.npz"), pos=bpos, angle=bang, corners=bcor)

    # SURVEY Appendix A, re-derived
    r, ns = ref_sim("small_loop_only_duckies")
    TS = 0.585
    kat = {"lane": []}
    for a, b, ang in [(1.30, 1.60, -1.2), (2.50, 1.28, 0.05), (3.65, 1.35, -0.9), (1.28, 2.50, -1.5708),
                      (3.72, 2.50, 1.5708), (1.40, 3.70, -2.6), (2.50, 3.72, 3.1), (3.60, 3.60, 2.2)]:
        lp = r.get_lane_pos2(np.array([a * TS, 0, b * TS]), ang)
        kat["lane"].append(dict(a=a, b=b, angle=ang, dist=float(lp.dist), dot_dir=float(lp.dot_dir), angle_deg=float(lp.angle_deg)))
    kat["corners"] = ns.simulator.get_agent_corners(np.array([1.0, 0, 1.0]), 0.3).tolist()
    kat["actual_center"] = ns.simulator._actual_center(np.array([1.0, 0, 1.0]), 0.3).tolist()
    kat["agent_safety_rad"] = ns.simulator.AGENT_SAFETY_RAD
    json.dump(kat, open(os.path.join(OUT, "ref_kat.json"), "w"), indent=1)

    # distortion: reference _invert_map/_fill_holes on the restated rectify maps
    from oracle import distortion as odist
    Dm = ns.distortion.Distortion.__new__(ns.distortion.Distortion)
    dout = {}
    for (w, h) in ((640, 480), (160, 120), (84, 84)):
        mapx, mapy = odist.rectify_maps(w, h)
        rx, ry = Dm._invert_map(mapx.copy(), mapy.copy())
        dout[f"sx_{w}x{h}"] = np.rint(rx.astype(np.float64)).astype(np.int16)
        dout[f"sy_{w}x{h}"] = np.rint(ry.astype(np.float64)).astype(np.int16)
    np.savez_compressed(os.path.join(OUT, "ref_distortion.npz"), **dout)
    # TrafficLightObj.step (objects.py:455-463, the reference's own code on a bare instance): pattern per step
    import types
    tl = {}
    for tag, dt in (("30hz", 1 / 30), ("20hz", 1 / 20), ("frame_skip_dt", 1 / 30 / 3)):
        o = types.SimpleNamespace(time=0, freq=5, pattern=0, texs=[0, 1], mesh=types.SimpleNamespace(textures=[0]))
        pat = np.zeros(1300, np.uint8)
        for t in range(1300):
            ns.objects.TrafficLightObj.step(o, dt)
            pat[t] = o.pattern
        tl[tag] = pat
    np.savez_compressed(os.path.join(OUT, "ref_trafficlight.npz"), **tl)

    # CheckerboardObj.step (objects.py:531-587) on a bare instance: centre over 2.5 cycles of the script
    o = types.SimpleNamespace(time=0, steps=-20, center=np.array([1.5, 0.0, 2.25]), reset_start=np.array([1.5, 0.0, 2.25]), pos=None)
    cb = np.zeros((600, 3))
    for t in range(600):
        ns.objects.CheckerboardObj.step(o, 1 / 30)
        cb[t] = o.center
    np.savez_compressed(os.path.join(OUT, "ref_checkerboard.npz"), center=cb)

    # ObjMesh parser (objmesh.py:55-358, the reference's own code) on the procedural asset tree of
    # tests/golden/make_assets.py: triangle soup in draw order, extents, per-chunk textures
    lib = assets.AssetLibrary(os.path.join(OUT, "assets"))
    om = {}
    for tag, stem, change in OBJ_CASES:
        r = refstub.ref_objmesh(lib.resolve(stem + ".obj"), stem, lib.resolve, change)
        for k in ("verts", "uvs", "normals", "colors", "chunk_sizes", "min_coords", "max_coords"):
            om[f"{tag}_{k}"] = r[k]
        om[f"{tag}_textures"] = np.array(["" if t is None else os.path.basename(t) for t in r["textures"]])
    np.savez_compressed(os.path.join(OUT, "ref_objmesh.npz"), **om)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
