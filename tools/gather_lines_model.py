"""CPU model of what k_raster_v3's record gathers cost the texture unit: distinct 128-byte cache lines per load instruction
(64 lanes = one 32 x 2 pixel slot of the 128 x 2 wavefront block) under different layouts of the S x S quad-record blocks,
over a sample of C3 poses (small_loop, 640 x 480, fisheye).  tools/ubench/fmt_load.hip: a 16-byte gather costs ~ 18 cycles when
its lanes share 8 lines and ~ 115 when every lane has its own -- the line count IS the L1 cost.

Layouts (record = 16 B, line = 8 records):
  row    8 x 1 records per line (x fastest; as shipped)
  col    1 x 8 (z fastest)
  best   per env the better of row / col for ITS heading (two copies of the pool, chosen per env)
  t42    4 x 2 tiles,   t24   2 x 4 tiles

An analysis aid (uses the test oracle's camera model), not part of the product path.     python tools/gather_lines_model.py [n_poses]
"""
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


def line_ids(X, Z):
    xi, zi = np.floor(X).astype(np.int64), np.floor(Z).astype(np.int64)
    tile = (zi >> 8) * 64 + (xi >> 8)
    cx, cz = xi & 255, zi & 255
    return {
        "row": (tile * S + cz) * 32 + (cx >> 3),
        "col": (tile * S + cx) * 32 + (cz >> 3),
        "t42": (tile * 128 + (cz >> 1)) * 64 + (cx >> 2),
        "t24": (tile * 64 + (cz >> 2)) * 128 + (cx >> 1),
        "zmerge": (tile * S + (cz & ~1)) * 32 + (cx >> 3),        # ablation -DDT_ABL_ZMERGE (wrong frames): two texture rows share their records
    }


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    ext = assets.mesh_extents(("duckie",))
    o = osim.OracleSim(assets.get_map("small_loop"), ext, domain_rand=False, seed=1000)
    rmx, rmy = pdist.distortion_maps(W, H)
    sx, sy = np.rint(rmx.astype(np.float64)), np.rint(rmy.astype(np.float64))
    valid = (sx >= 0) & (sx < W) & (sy >= 0) & (sy < H)
    nx = 2 * (sx + 0.5) / W - 1
    ny = 1 - 2 * (sy + 0.5) / H
    rng = np.random.default_rng(5)
    ts = o.map.tile_size
    tot = {k: 0.0 for k in ("row", "col", "best", "t42", "t24", "zmerge", "t42nf")}
    nf_tot = [0, 0]
    nf_blk = [0, 0, 0]
    n_instr = 0
    shapes = {}
    hist = {k: np.zeros(65) for k in ("row", "best", "t42")}
    for k in range(n):
        o.reset()
        for _ in range(int(rng.integers(0, 40))):
            a = rng.uniform(-1, 1, 2); a[0] = abs(a[0]) * 0.6 + 0.1
            _, done, _ = o.step_vel_steer(a)
            if done:
                o.reset()
        cam = raster.Camera(o.cur_pos, o.cur_angle, width=W, height=H)
        xe, ye, yla, fwd = raster._rays(cam, nx, ny)
        tt, wx, wz = raster._plane_hit(cam, xe, fwd, yla, cam.C[1])
        ok = valid & (yla < 0) & (tt >= raster.NEAR) & (tt <= raster.FAR)
        X = np.where(ok, wx / ts * S + 0.5 + 4 * S, 0.0)
        Z = np.where(ok, wz / ts * S + 0.5 + 4 * S, 0.0)
        X = np.clip(X, 0, 40 * S); Z = np.clip(Z, 0, 40 * S)
        ids = line_ids(X, Z)
        # pixels whose MSAA reach (cells) exceeds what any record can grant (128 = half a tile): always resolved by the exact path, for every env --
        # their gather in the env loop is wasted; "t42nf": those lanes read one shared record instead
        reach = np.zeros_like(X)
        for ox_, oy_ in ((0.375, 0.125), (-0.125, 0.375), (-0.375, -0.125), (0.125, -0.375)):
            nxs, nys = nx + 2 * ox_ / W, ny + 2 * oy_ / H
            xe2, ye2, yla2, fwd2 = raster._rays(cam, nxs, nys)
            tt2, wx2, wz2 = raster._plane_hit(cam, xe2, fwd2, yla2, cam.C[1])
            with np.errstate(invalid="ignore"):
                reach = np.maximum(reach, np.where(yla2 < 0, np.hypot(wx2 - wx, wz2 - wz) / ts * S, 1e9))
        never = ok & (1.05 * reach + 0.5 >= 128)
        ids["t42nf"] = np.where(never, -1, ids["t42"])
        nf_tot[0] += int(never.sum()); nf_tot[1] += int(ok.sum())
        # 128 x 2 wavefront blocks whose tile-plane pixels are ALL such pixels (their env loop could skip loads and filter altogether)
        okb2 = ok.reshape(H // 2, 2, W // 128, 128).transpose(0, 2, 1, 3).reshape(-1, 256)
        nvb2 = never.reshape(H // 2, 2, W // 128, 128).transpose(0, 2, 1, 3).reshape(-1, 256)
        full = okb2.any(1) & (nvb2 | ~okb2).all(1)
        lines_blk = ids["t42"].reshape(H // 2, 2, W // 32, 32).transpose(0, 2, 1, 3).reshape(-1, 64)
        nf_blk[0] += int(full.sum()); nf_blk[1] += int(okb2.any(1).sum()); nf_blk[2] += int(nvb2[full].sum())
        per = {}
        for sw, sh in ((64, 1), (16, 4), (8, 8)):            # other slot shapes (not built): lines per gather under the shipped and the tiled layout
            for name in ("row", "t42", "t24"):
                b = ids[name].reshape(H // sh, sh, W // sw, sw).transpose(0, 2, 1, 3).reshape(-1, 64)
                okb = ok.reshape(H // sh, sh, W // sw, sw).transpose(0, 2, 1, 3).reshape(-1, 64)
                bs = np.sort(b[okb.all(1)], axis=1)
                key = f"{name}@{sw}x{sh}"
                shapes.setdefault(key, [0.0, 0])
                shapes[key][0] += float((1 + (np.diff(bs, axis=1) != 0).sum(1)).sum()); shapes[key][1] += len(bs)
        for name, idv in ids.items():
            # slots: 32 x 2 pixel sub-blocks of the 128 x 2 wavefront block
            b = idv.reshape(H // 2, 2, W // 32, 32).transpose(0, 2, 1, 3).reshape(-1, 64)
            okb = ok.reshape(H // 2, 2, W // 32, 32).transpose(0, 2, 1, 3).reshape(-1, 64)
            use = okb.all(1)                                 # slots made of tile-plane candidates only
            bs = np.sort(b[use], axis=1)
            per[name] = 1 + (np.diff(bs, axis=1) != 0).sum(1)
        m = len(per["row"])
        n_instr += m
        best = per["row"] if per["row"].sum() <= per["col"].sum() else per["col"]
        per["best"] = best
        for name in tot:
            tot[name] += per[name].sum()
        for name in hist:
            hist[name] += np.bincount(per[name], minlength=65)[:65]
    print(f"{n} poses, {n_instr} full 32 x 2 slots on the tile plane ({n_instr / n / (H // 2 * (W // 32)) * 100:.1f} % of the slots)")
    for name, v in tot.items():
        print(f"  {name:5s} {v / n_instr:6.2f} distinct lines per load instruction")
    print(f"  pixels on the tile plane that no record can make a one-ray pixel: {nf_tot[0] / max(nf_tot[1], 1) * 100:.2f} %")
    print(f"  128 x 2 blocks made of such pixels only: {nf_blk[0] / max(nf_blk[1], 1) * 100:.2f} % of the blocks with tile-plane pixels, holding {nf_blk[2] / max(nf_tot[0], 1) * 100:.0f} % of those pixels")
    for key, (tot_l, cnt) in shapes.items():
        print(f"  {key:12s} {tot_l / max(cnt, 1):6.2f} distinct lines per load instruction")
    for name, h in hist.items():
        c = np.cumsum(h) / h.sum()
        print(f"  {name:5s} share of instructions with <= 8 / 16 / 32 lines: {c[8]:.2f} / {c[16]:.2f} / {c[32]:.2f};  with 64: {h[64] / h.sum():.3f}")


if __name__ == "__main__":
    main()
