"""CPU model of k_raster_v3's one-ray / exact-path classification (shared camera, C3: small_loop, 640 x 480, fisheye): which
pixels the raster QUEUES for the exact path under its shipped test and under candidate tests, against the pixels that truly
need it (their four MSAA samples do not all see one primitive).  Prices a change of the test before it is built.

Shipped test (render_v3.inc `fastm`, render.hip k_pix_setup / pix_inv): record meta k = cells to the nearest tile boundary
(min over both axes, per CELL) > floor(reach + 0.5), reach = a CIRCLE (radius = the larger sample displacement, + 5 %) in cells.

Candidates:
  axis      per-axis: kx > floor(ex + .5) and kz > floor(ez + .5), (ex, ez) = the footprint's yaw-local half-extents (lateral,
            forward) taken to tile axes with the env's |sin|, |cos|
  axis_blk  the same with the half-extents maximised over the 32 x 2 pixel slot (a scalar per (slot, env))
  fake      shipped test, but boundaries between tiles that share one quad block (same texture, same angle) do not count
  fake+axis both
  exact     sub-cell circular test (what phase 1 of resolve_region decides)

An analysis aid (uses the test oracle's map / camera model), not part of the product path.   python tools/edge_test_model.py [n_poses]
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gym-duckietown_amd"))
sys.path.insert(0, ROOT)

from dtsim import assets                      # noqa: E402
from dtsim import distortion as pdist         # noqa: E402
from oracle import raster, sim as osim        # noqa: E402

W, H, S = 640, 480, 256
OX = [-0.125, 0.375, -0.375, 0.125]
OY = [-0.375, -0.125, 0.125, 0.375]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    mapname = sys.argv[2] if len(sys.argv) > 2 else "small_loop"
    ext = assets.mesh_extents(("duckie",))
    o = osim.OracleSim(assets.get_map(mapname), ext, domain_rand=False, seed=1000)
    m = o.map
    gw, gh, ts = m.grid_width, m.grid_height, m.tile_size
    qpm = S / ts
    # block id per tile (texture kind, angle) -- what dtsim_set_maps shares a quad block over
    blk = -np.ones((gh + 2, gw + 2), int)
    ids = {}
    for t in m.grid:
        if t is None:
            continue
        i, j = t["coords"]
        blk[j + 1, i + 1] = ids.setdefault((t["kind"], t["angle"]), len(ids))
    rmx, rmy = pdist.distortion_maps(W, H)
    sx, sy = np.rint(rmx.astype(np.float64)), np.rint(rmy.astype(np.float64))
    valid = (sx >= 0) & (sx < W) & (sy >= 0) & (sy < H)
    nx = 2 * (sx + 0.5) / W - 1
    ny = 1 - 2 * (sy + 0.5) / H
    cam0 = raster.Camera(np.zeros(3), 0.0, width=W, height=H)
    tx, ty, sth, cth, Cy = cam0.tx, cam0.ty, cam0.sth, cam0.cth, cam0.C[1]

    def ray(nx_, ny_):
        xe, ye = nx_ * tx, ny_ * ty
        return xe, ye * cth - sth, ye * sth + cth                     # xe, yla, fwd

    # ---- env-invariant per-pixel quantities (pix_inv)
    xe, yla, fwd = ray(nx, ny)
    down = yla < 0
    inv = np.where(down, 1.0 / np.where(down, -yla, 1.0), 0.0)
    t = Cy * inv
    lr, lf = t * xe, t * fwd
    ex = 0.375 * 2 / W * 1.01 * tx
    ey = 0.375 * 2 / H * 1.01 * ty
    dy = ey * abs(cth)
    rho = dy * inv
    kappa = rho / np.maximum(1 - rho, 0.25)
    ax_, ry_, fy_ = t * ex, np.abs(lr) * kappa, np.abs(lf) * kappa + t * ey * abs(sth)
    r1x, r1f, r2x, r2f = ax_ + ry_ / 3, fy_ / 3, ax_ / 3 + ry_, fy_
    mrg = 1.05 * np.sqrt(np.maximum(r1x ** 2 + r1f ** 2, r2x ** 2 + r2f ** 2))
    hr, hf = 1.05 * np.maximum(r1x, r2x), 1.05 * np.maximum(r1f, r2f)
    tg = (Cy - raster.GROUND_Y) * inv
    sky = ~down & (yla - dy >= 0)
    always = (~down & ~sky) | (down & ((rho > 0.25) | (tg * (1 + 2 * rho) > raster.FAR * 0.98) | (t * (1 - 2 * rho) < raster.NEAR * 1.02)))
    cand = valid & ~sky
    plain = cand & ~always & down & (t >= raster.NEAR) & (t <= raster.FAR) & (tg >= raster.NEAR) & (tg <= raster.FAR)
    # the four sample hits (SampTab)
    slr, slf, sdown = [], [], []
    for s in range(4):
        xs, ys, fs = ray(nx + OX[s] * 2 / W, ny - OY[s] * 2 / H)
        d = ys < 0
        iv = np.where(d, 1.0 / np.where(d, -ys, 1.0), 0.0)
        slr.append(Cy * iv * xs); slf.append(Cy * iv * fs); sdown.append(d & (Cy * iv >= raster.NEAR) & (Cy * iv <= raster.FAR))
    # slot-level (32 x 2) maxima of the half-extents
    def slot_max(a):
        b = a.reshape(H // 2, 2, W // 32, 32).max(axis=(1, 3))
        return np.repeat(np.repeat(b, 2, axis=0), 32, axis=1)
    hr_b, hf_b = slot_max(np.where(plain, hr, 0)), slot_max(np.where(plain, hf, 0))

    rng = np.random.default_rng(5)
    names = ["shipped", "axis", "axis_blk", "fake", "fake+axis", "fake+axis_blk", "exact", "true"]
    tot = {k: 0 for k in names}
    n_px = 0
    cat = {"always": 0, "seam_cell(k=0)": 0, "k>=1": 0, "untextured/offgrid": 0}
    for k in range(n):
        o.reset()
        for _ in range(int(rng.integers(0, 40))):
            a = rng.uniform(-1, 1, 2); a[0] = abs(a[0]) * 0.6 + 0.1
            _, done, _ = o.step_vel_steer(a)
            if done:
                o.reset()
        cam = raster.Camera(o.cur_pos, o.cur_angle, width=W, height=H)
        sa, ca, Cx, Cz = cam.sa, cam.ca, cam.C[0], cam.C[2]

        def to_q(lr_, lf_):
            wx, wz = Cx + lr_ * sa + lf_ * ca, Cz + lr_ * ca - lf_ * sa
            return wx * qpm + 0.5, wz * qpm + 0.5, wx, wz           # quad coordinates (grid origin at 0.5), world
        X, Z, wx, wz = to_q(lr, lf)
        xi, zi = np.floor(X).astype(int), np.floor(Z).astype(int)
        ti, tj = xi >> 8, zi >> 8
        cx, cz = xi & 255, zi & 255
        ing = (ti >= 0) & (tj >= 0) & (ti < gw) & (tj < gh)
        b_c = np.where(ing, blk[np.clip(tj, -1, gh) + 1, np.clip(ti, -1, gw) + 1], -1)
        textured = b_c >= 0
        kx, kz = np.minimum(cx, S - cx), np.minimum(cz, S - cz)
        kk = np.where(textured, np.minimum(kx, kz), 0)
        Mi = np.floor(mrg * qpm + 0.5)
        fast = plain & (kk > Mi)
        # off-grid cells whose every sample stays off the grid: ground quad on the one-ray path
        clear = (wx < -mrg) | (wx > gw * ts + mrg) | (wz < -mrg) | (wz > gh * ts + mrg)
        gfast = plain & ~ing & clear
        q_shipped = cand & ~fast & ~gfast
        # per-axis
        exc, ezc = (abs(sa) * hr + abs(ca) * hf) * qpm, (abs(ca) * hr + abs(sa) * hf) * qpm
        fast_ax = plain & textured & (kx > np.floor(exc + 0.5)) & (kz > np.floor(ezc + 0.5))
        exb, ezb = (abs(sa) * hr_b + abs(ca) * hf_b) * qpm, (abs(ca) * hr_b + abs(sa) * hf_b) * qpm
        fast_axb = plain & textured & (kx > np.floor(exb + 0.5)) & (kz > np.floor(ezb + 0.5))
        # fake seams: a boundary towards a neighbour of the same block does not count (distance = a tile further)
        tjc, tic = np.clip(tj, -1, gh) + 1, np.clip(ti, -1, gw) + 1
        def nb(dj, di):
            return np.where(ing, blk[np.clip(tjc + dj, 0, gh + 1), np.clip(tic + di, 0, gw + 1)], -2)
        same_l, same_r, same_u, same_d = nb(0, -1) == b_c, nb(0, 1) == b_c, nb(-1, 0) == b_c, nb(1, 0) == b_c
        kxf = np.minimum(np.where(same_l, cx + S, cx), np.where(same_r, 2 * S - cx, S - cx))
        kzf = np.minimum(np.where(same_u, cz + S, cz), np.where(same_d, 2 * S - cz, S - cz))
        fast_fk = plain & textured & (np.minimum(kxf, kzf) > Mi)
        fast_fk_ax = plain & textured & (kxf > np.floor(exc + 0.5)) & (kzf > np.floor(ezc + 0.5))
        fast_fk_axb = plain & textured & (kxf > np.floor(exb + 0.5)) & (kzf > np.floor(ezb + 0.5))
        # exact circular (phase 1): owner tile by (X - .5), distance to its boundary in cells
        Xo, Zo = X - 0.5, Z - 0.5
        ux, uz = Xo - np.floor(Xo / S) * S, Zo - np.floor(Zo / S) * S
        d = np.minimum(np.minimum(ux, S - ux), np.minimum(uz, S - uz))
        oti, otj = np.floor(Xo / S).astype(int), np.floor(Zo / S).astype(int)
        oing = (oti >= 0) & (otj >= 0) & (oti < gw) & (otj < gh)
        fast_ex = plain & oing & (blk[np.clip(otj, -1, gh) + 1, np.clip(oti, -1, gw) + 1] >= 0) & (d > mrg * qpm)
        # truth: the four samples' primitives
        keys = []
        for s in range(4):
            Xs, Zs, wxs, wzs = to_q(slr[s], slf[s])
            si, sj = np.floor((Xs - 0.5) / S).astype(int), np.floor((Zs - 0.5) / S).astype(int)
            sin_ = (si >= 0) & (sj >= 0) & (si < gw) & (sj < gh)
            st = sdown[s] & sin_ & (blk[np.clip(sj, -1, gh) + 1, np.clip(si, -1, gw) + 1] >= 0)
            kg = (Cy - raster.GROUND_Y) / Cy
            gx, gz = Cx + kg * (wxs - Cx), Cz + kg * (wzs - Cz)
            sg = sdown[s] & ~st & (np.abs(gx) <= raster.GROUND_HALF) & (np.abs(gz) <= raster.GROUND_HALF)
            keys.append(np.where(st, 16 + sj * 64 + si, np.where(sg, 2, 1)))
        true_edge = cand & ((keys[0] != keys[1]) | (keys[0] != keys[2]) | (keys[0] != keys[3]))
        n_px += W * H
        tot["shipped"] += q_shipped.sum()
        tot["axis"] += (cand & ~fast_ax & ~gfast).sum()
        tot["axis_blk"] += (cand & ~fast_axb & ~gfast).sum()
        tot["fake"] += (cand & ~fast_fk & ~gfast).sum()
        tot["fake+axis"] += (cand & ~fast_fk_ax & ~gfast).sum()
        tot["fake+axis_blk"] += (cand & ~fast_fk_axb & ~gfast).sum()
        tot["exact"] += (cand & ~fast_ex & ~gfast).sum()
        tot["true"] += true_edge.sum()
        cat["always"] += (q_shipped & ~plain).sum()
        cat["seam_cell(k=0)"] += (q_shipped & plain & textured & (kk == 0)).sum()
        cat["k>=1"] += (q_shipped & plain & textured & (kk >= 1)).sum()
        cat["untextured/offgrid"] += (q_shipped & plain & ~textured).sum()
    print(f"{mapname}: {n} poses, {W}x{H} fisheye; queued pixels as a share of the frame")
    for k_ in names:
        print(f"  {k_:14s} {100 * tot[k_] / n_px:6.3f} %")
    print("  shipped queue by cause: " + ", ".join(f"{k_} {100 * v / n_px:.3f} %" for k_, v in cat.items()))


if __name__ == "__main__":
    main()
