"""CPU analysis for the raster's work decomposition: how the 64x4-pixel wavefront blocks of a C3 frame split into
all-sky / ground-only / tile-only / mixed, over a sample of poses of the bench workload (small_loop, 640x480, fisheye).
Uses the test oracle's camera model (oracle/raster.py) -- an analysis aid, not part of the product path.

    python tools/wave_classes.py [n_poses]
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

W, H, BW, BH = 640, 480, 64, 4


def pixel_classes(cam, scene, sx, sy, valid):
    """class of every output pixel's centre ray: 0 border, 1 sky, 2 ground quad, 3 tile"""
    nx = 2 * (sx + 0.5) / W - 1
    ny = 1 - 2 * (sy + 0.5) / H
    xe, ye, yla, fwd = raster._rays(cam, nx, ny)
    down = yla < 0
    tt, wx, wz = raster._plane_hit(cam, xe, fwd, yla, cam.C[1])
    ts = scene.m.tile_size
    with np.errstate(invalid="ignore"):
        fi, fj = np.floor(wx / ts), np.floor(wz / ts)
    tok = down & (tt >= raster.NEAR) & (tt <= raster.FAR) & (fi >= 0) & (fj >= 0) & (fi < scene.m.grid_width) & (fj < scene.m.grid_height)
    ii, jj = np.where(tok, fi, 0).astype(int), np.where(tok, fj, 0).astype(int)
    tok &= scene.present[jj, ii]
    tg, gx, gz = raster._plane_hit(cam, xe, fwd, yla, cam.C[1] - raster.GROUND_Y)
    gok = down & (tg >= raster.NEAR) & (tg <= raster.FAR) & (np.abs(gx) <= raster.GROUND_HALF) & (np.abs(gz) <= raster.GROUND_HALF)
    cls = np.where(tok, 3, np.where(gok, 2, 1))
    key = np.where(tok, 16 + jj * 64 + ii, cls)            # primitive id: the tile, or sky / ground
    return np.where(valid, cls, 0), np.where(valid, key, 0)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    ext = assets.mesh_extents(("duckie",))
    o = osim.OracleSim(assets.get_map("small_loop"), ext, domain_rand=False, seed=1000)
    kinds = {t["kind"] for t in o.map.grid if t is not None}
    scene = raster.Scene(o.map, {k: assets.get_texture(k) for k in kinds}, {"duckie": assets.get_mesh("duckie"), "*": assets.get_mesh("*")})
    rmx, rmy = pdist.distortion_maps(W, H)
    sx, sy = np.rint(rmx.astype(np.float64)), np.rint(rmy.astype(np.float64))
    valid = (sx >= 0) & (sx < W) & (sy >= 0) & (sy < H)
    rng = np.random.default_rng(5)
    names = ["border/sky only", "ground only (+sky)", "tile only (+sky)", "tile + ground"]
    tot = np.zeros(4)
    px = np.zeros(4)
    edge_px = 0.0
    edge_blk_any = tile_blks = 0

    def blk_is_tile(cls):
        b = cls.reshape(H // BH, BH, W // BW, BW).transpose(0, 2, 1, 3).reshape(-1, BH * BW)
        return (b == 3).any(1)
    for k in range(n):
        o.reset()
        for _ in range(int(rng.integers(0, 40))):
            a = rng.uniform(-1, 1, 2); a[0] = abs(a[0]) * 0.6 + 0.1
            _, done, _ = o.step_vel_steer(a)
            if done:
                o.reset()
        cam = raster.Camera(o.cur_pos, o.cur_angle, width=W, height=H)
        cls, key0 = pixel_classes(cam, scene, sx, sy, valid)
        # a pixel needs the exact 4-sample path when its MSAA samples do not all see the same primitive
        edge = np.zeros((H, W), bool)
        for (ox, oy) in raster.SAMPLE_OFFSETS:
            _, ks = pixel_classes(cam, scene, sx + ox, sy + oy, valid)
            edge |= ks != key0
        eblk = edge.reshape(H // BH, BH, W // BW, BW).transpose(0, 2, 1, 3).reshape(-1, BH * BW)
        edge_px += edge.mean(); edge_blk_any += (eblk.any(1) & (blk_is_tile(cls))).sum(); tile_blks += blk_is_tile(cls).sum()
        blk = cls.reshape(H // BH, BH, W // BW, BW).transpose(0, 2, 1, 3).reshape(-1, BH * BW)
        has_t, has_g = (blk == 3).any(1), (blk == 2).any(1)
        kind = np.where(has_t & has_g, 3, np.where(has_t, 2, np.where(has_g, 1, 0)))
        tot += np.bincount(kind, minlength=4)
        px += np.bincount(np.clip(cls, 0, 3).ravel(), minlength=4)
    tot /= tot.sum(); px /= px.sum()
    print(f"{n} poses, {W}x{H} fisheye frames, {BW}x{BH}-pixel wavefront blocks:")
    for nm, f in zip(names, tot):
        print(f"  {nm:22s} {100 * f:5.1f} % of the blocks")
    print("  pixels: border %.1f %%, sky %.1f %%, ground %.1f %%, tile %.1f %%" % tuple(100 * px))
    print(f"  pixels whose 4 MSAA samples see different primitives: {100 * edge_px / n:.2f} %; "
          f"{100 * edge_blk_any / max(tile_blks, 1):.1f} % of the blocks with tile pixels contain at least one")


if __name__ == "__main__":
    main()
