"""CPU model of the raster's record gathers: how many distinct cache lines does one 64-lane load instruction of
k_raster_v3 touch, per record layout?  (The main loop is bound by the texture unit working through the lines of a
wavefront's 16-byte loads: profiles/r03_variants_ab.txt.)  Poses: the reference-order spawns of small_loop.

    python tools/sim_lines.py [n_envs]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gym-duckietown_amd")); sys.path.insert(0, ROOT)
import numpy as np
from dtsim import assets, distortion as dist_mod
from oracle import sim as osim

W, H, S = 640, 480, 256
n_env = int(sys.argv[1]) if len(sys.argv) > 1 else 48
rmx, rmy = dist_mod.distortion_maps(W, H)
sx, sy = np.rint(rmx).astype(np.int64), np.rint(rmy).astype(np.int64)
ok = (sx >= 0) & (sx < W) & (sy >= 0) & (sy < H)
nx = (2.0 * (sx + 0.5)) / W - 1.0
ny = 1.0 - (2.0 * (sy + 0.5)) / H
th = np.deg2rad(19.15); sth, cth = np.sin(th), np.cos(th)
ty = np.tan(0.5 * np.deg2rad(75.0)); tx = ty * W / H
Cy = 0.108
xe, ye = nx * tx, ny * ty
yla = ye * cth - sth
fwd = ye * sth + cth
hit = ok & (yla < 0)
t = np.where(hit, Cy / np.where(hit, -yla, 1.0), 0.0)
lr, lf = t * xe, t * fwd                                   # yaw-local hit (right, forward), metres

md = assets.get_map("small_loop")
ts = md["tile_size"]
gh, gw = len(md["tiles"]), len(md["tiles"][0])
present = np.array([[c is not None and c != "" and not str(c).startswith("empty") for c in row] for row in md["tiles"]])
ext = assets.mesh_extents(("duckie",))
poses = []
for e in range(n_env):
    o = osim.OracleSim(md, ext, domain_rand=False, seed=1000 + e)
    poses.append((o.cur_pos.copy(), float(o.cur_angle)))

def layouts(cx, cz, tile):
    """record index inside the tile's block under each layout; lines of 64 B (4 records) and 128 B (8 records)"""
    out = {}
    lin_x = cz * S + cx
    lin_z = cx * S + cz
    t22 = ((cz >> 1) * (S // 2) + (cx >> 1)) * 4 + (cz & 1) * 2 + (cx & 1)          # 2 x 2 records per 64 B
    t42 = ((cz >> 1) * (S // 4) + (cx >> 2)) * 8 + (cz & 1) * 4 + (cx & 3)          # 4 x 2 records per 128 B
    t24 = ((cz >> 2) * (S // 2) + (cx >> 1)) * 8 + (cz & 3) * 2 + (cx & 1)          # 2 x 4 records per 128 B
    for name, idx in (("x-major", lin_x), ("z-major", lin_z), ("2x2", t22), ("4x2", t42), ("2x4", t24)):
        out[name] = tile.astype(np.int64) * (S * S) + idx
    return out

tot = {}
n_instr = 0
for pos, ang in poses:
    ca, sa = np.cos(ang), np.sin(ang)
    Cx, Cz = pos[0] + 0.066 * ca, pos[2] - 0.066 * sa
    wx = Cx + lr * sa + lf * ca
    wz = Cz + lr * ca - lf * sa
    X = wx / ts * S + 0.5
    Z = wz / ts * S + 0.5
    xi, zi = np.floor(X).astype(np.int64), np.floor(Z).astype(np.int64)
    ti, tj = xi >> 8, zi >> 8
    on = hit & (ti >= 0) & (ti < gw) & (tj >= 0) & (tj < gh)
    tile = np.where(on, tj * gw + ti, -1)
    cx, cz = xi & 255, zi & 255
    recs = layouts(cx, cz, tile)
    # choice of x-major / z-major per env: by which axis the screen-x direction is closer to
    swap = abs(ca) > abs(sa)          # right vector = (sa, ca): mostly along z when |ca| > |sa|
    recs["xz-choice"] = recs["z-major"] if swap else recs["x-major"]
    recs["4x2/2x4-choice"] = recs["2x4"] if swap else recs["4x2"]
    # instruction = 64 adjacent pixels of a row (k_raster_v3 map 0); off-grid lanes all hit one record
    for name, r in recs.items():
        r = np.where(on, r, -1).reshape(H, W // 64, 64)
        active = hit.reshape(H, W // 64, 64).any(axis=2)
        for lb, per in (("64B", 4), ("128B", 8)):
            ln = np.where(r >= 0, r // per, -1)
            ln_sorted = np.sort(ln, axis=2)
            distinct = 1 + (np.diff(ln_sorted, axis=2) != 0).sum(axis=2)
            tot[(name, lb)] = tot.get((name, lb), 0) + int(distinct[active].sum())
    n_instr += int(hit.reshape(H, W // 64, 64).any(axis=2).sum())
print(f"{n_env} envs, {n_instr} load instructions (64 adjacent pixels of a row each)")
for (name, lb), v in sorted(tot.items(), key=lambda kv: (kv[0][1], kv[1])):
    print(f"  {lb:>4} lines  {name:<16} {v / n_instr:6.2f} distinct lines per instruction")
