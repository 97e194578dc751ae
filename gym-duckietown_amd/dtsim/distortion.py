"""One-time host build of the fisheye remap tables (Distortion, distortion.py:11-56,85-265).

The reference builds `rmapx/rmapy` lazily on the first distort() call with three OpenCV
calls (absent here -- restated from OpenCV's documented plumb-bob model, PARITY UNPINNED)
plus its own _invert_map/_fill_holes, then runs cv2.remap(INTER_NEAREST) on every frame.
Here the tables are built once (vectorised numpy) and handed to the library with
dtsim_set_distortion_lut; the per-frame remap disappears into the raster's ray set-up.

_fill_holes visits holes in the iteration order of a Python set (see fill_holes): kept.
"""
from __future__ import annotations

import functools
import itertools

import numpy as np

# distortion.py:13-32
CAL_W, CAL_H = 640, 480
CAMERA_MATRIX = np.array([[305.5718893575089, 0, 303.0797142544728],
                          [0, 308.8338858195428, 231.8845403702499],
                          [0, 0, 1]], dtype=np.float64)
DIST_COEFS = np.array([-0.2, 0.0305, 0.0005859930422629722, -0.0006697840226199427, 0], dtype=np.float64)


def undistort_points_normalized(pts, K, D, iters=5):
    """cv::undistortPoints with R = P = identity: pixel -> normalised coords, 5 fixed-point
    iterations (OpenCV calib3d/imgproc undistort.cpp, cvUndistortPointsInternal)."""
    k1, k2, p1, p2, k3 = D[:5]
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    x0 = (pts[:, 0].astype(np.float64) - cx) * (1.0 / fx)
    y0 = (pts[:, 1].astype(np.float64) - cy) * (1.0 / fy)
    x, y = x0.copy(), y0.copy()
    for _ in range(iters):
        r2 = x * x + y * y
        icdist = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2)
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x = (x0 - dx) * icdist
        y = (y0 - dy) * icdist
    return np.stack([x, y], axis=1).astype(np.float32)


def optimal_new_camera_matrix(K=CAMERA_MATRIX, D=DIST_COEFS, size=(CAL_W, CAL_H), alpha=0.0):
    """cv::getOptimalNewCameraMatrix(K, D, size, alpha, newImgSize=size,
    centerPrincipalPoint=false) via icvGetRectangles on a 9x9 grid (distortion.py:51-56)."""
    W, H = size
    N = 9
    g = np.array([[np.float32(x) * np.float32(W) / np.float32(N - 1), np.float32(y) * np.float32(H) / np.float32(N - 1)]
                  for y in range(N) for x in range(N)], dtype=np.float32)
    p = undistort_points_normalized(g, K, D).reshape(N, N, 2)
    iX0, iX1 = p[:, 0, 0].max(), p[:, N - 1, 0].min()
    iY0, iY1 = p[0, :, 1].max(), p[N - 1, :, 1].min()
    oX0, oX1, oY0, oY1 = p[..., 0].min(), p[..., 0].max(), p[..., 1].min(), p[..., 1].max()
    inner = (np.float32(iX0), np.float32(iY0), np.float32(iX1 - iX0), np.float32(iY1 - iY0))
    outer = (np.float32(oX0), np.float32(oY0), np.float32(oX1 - oX0), np.float32(oY1 - oY0))
    fx0, fy0 = (W - 1) / float(inner[2]), (H - 1) / float(inner[3])
    cx0, cy0 = -fx0 * float(inner[0]), -fy0 * float(inner[1])
    fx1, fy1 = (W - 1) / float(outer[2]), (H - 1) / float(outer[3])
    cx1, cy1 = -fx1 * float(outer[0]), -fy1 * float(outer[1])
    M = np.array(K, dtype=np.float64)
    M[0, 0] = fx0 * (1 - alpha) + fx1 * alpha
    M[1, 1] = fy0 * (1 - alpha) + fy1 * alpha
    M[0, 2] = cx0 * (1 - alpha) + cx1 * alpha
    M[1, 2] = cy0 * (1 - alpha) + cy1 * alpha
    return M


def rectify_maps(K, D, newK, size):
    """cv::initUndistortRectifyMap(K, D, R=I, newK, size, CV_32FC1) (distortion.py:100-107):
    for every *rectified* pixel, the distorted-image coordinate it samples."""
    W, H = size
    k1, k2, p1, p2, k3 = D[:5]
    fx, fy, u0, v0 = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    ir = np.linalg.inv(newK)
    rows = np.arange(H, dtype=np.float64)
    # the C loop accumulates _x += ir[0] per column: np.add.accumulate is the same sequence
    steps = np.full(W - 1, 1.0)

    def acc(start, inc):
        return np.add.accumulate(np.concatenate([start[:, None], inc * np.ones((H, 1)) * steps[None, :]], axis=1), axis=1)

    _x = acc(rows * ir[0, 1] + ir[0, 2], ir[0, 0])
    _y = acc(rows * ir[1, 1] + ir[1, 2], ir[1, 0])
    _w = acc(rows * ir[2, 1] + ir[2, 2], ir[2, 0])
    w = 1.0 / _w
    x, y = _x * w, _y * w
    x2, y2 = x * x, y * y
    r2, _2xy = x2 + y2, 2 * x * y
    kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
    xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2)
    yd = y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy
    return (fx * xd + u0).astype(np.float32), (fy * yd + v0).astype(np.float32)


def invert_map(mapx, mapy):
    """Distortion._invert_map (distortion.py:138-216): weighted 9-neighbour scatter with
    numpy's non-accumulating fancy `+=` (duplicate targets: last writer wins), divide."""
    H, W = mapx.shape
    ax = np.zeros((H, W), "float32")
    ay = np.zeros((H, W), "float32")
    aw = np.zeros((H, W), "float32")
    xd = np.clip(mapx.astype("int32"), 2, W - 2)
    yd = np.clip(mapy.astype("int32"), 2, H - 2)
    ys, xs = np.meshgrid(np.arange(H, dtype="int32"), np.arange(W, dtype="int32"), indexing="ij")
    for di, dj, wt in ((-1, -1, 7), (-1, 0, 10), (-1, 1, 7), (0, -1, 10), (0, 0, 20), (0, 1, 10),
                       (1, -1, 7), (1, 0, 10), (1, 1, 7)):
        ty, tx = yd + di, xd + dj
        aw[ty, tx] += wt
        ax[ty, tx] += wt * xs
        ay[ty, tx] += wt * ys
    rx = np.full((H, W), np.nan, dtype=mapx.dtype)
    ry = np.full((H, W), np.nan, dtype=mapx.dtype)
    nz = aw > 0
    rx[nz] = ax[nz] / aw[nz]
    ry[nz] = ay[nz] / aw[nz]
    return rx, ry


def fill_holes(rx, ry):
    """Distortion._fill_holes (distortion.py:218-265): NaN entries take the first non-NaN
    neighbour in the reference's offset list (its (i-R-1) off-by-one is kept: 11 offsets in
    [-3..1]^2 with norm <= 2, stable-sorted by norm), pass after pass until no progress.

    Values filled earlier in a pass feed later holes of the same pass, so the result
    depends on the visiting order.  The reference visits `list(holes)` of a Python *set* of
    (i, j) tuples inserted in row-major order; to land on the same table we keep the holes
    in the same container with the same insert/remove sequence (CPython >= 3.8 hashes int
    tuples deterministically, so the order is reproducible)."""
    H, W = rx.shape
    Rr = 2
    F = 2 * Rr + 1
    deltas = [(i - Rr - 1, j - Rr - 1) for i, j in itertools.product(range(F), range(F))]
    deltas = [d for d in deltas if np.hypot(d[0], d[1]) <= Rr]
    deltas.sort(key=lambda d: np.hypot(d[0], d[1]))
    holes = set()
    for i, j in np.argwhere(np.isnan(rx)).tolist():      # row-major, like product(range(H), range(W))
        holes.add((i, j))
    while holes:
        filled = 0
        for (i, j) in list(holes):
            for di, dj in deltas:
                u, v = i + di, j + dj
                if 0 <= u < H and 0 <= v < W and not np.isnan(rx[u, v]):
                    rx[i, j], ry[i, j] = rx[u, v], ry[u, v]
                    filled += 1
                    holes.remove((i, j))
                    break
        if filled == 0:
            break
    return rx, ry


@functools.lru_cache(maxsize=8)
def distortion_maps(width: int, height: int):
    """(rmapx, rmapy) float32 [height,width]: Distortion().distort's cached tables for an
    observation of this size.  new_K is always computed for 640x480 (distortion.py:13-14,51)."""
    newK = optimal_new_camera_matrix()
    mapx, mapy = rectify_maps(CAMERA_MATRIX, DIST_COEFS, newK, (width, height))
    rx, ry = invert_map(mapx, mapy)
    rx, ry = fill_holes(rx, ry)
    return np.ascontiguousarray(rx, np.float32), np.ascontiguousarray(ry, np.float32)


# UndistortWrapper (src/gym_duckietown/wrappers.py:145-227): the wrapper turns the simulator's own fisheye off
# (`env.undistort = True`) and remaps the rectilinear render with initUndistortRectifyMap(K, D, I, P) / INTER_NEAREST.
UNDISTORT_P = np.array([[220.2460277141687, 0, 301.8668918355899], [0, 238.6758484095299, 227.0880056118307], [0, 0, 1.0]])


def undistort_wrapper_maps(width: int, height: int):
    """(mapx, mapy) float32 [height,width] of UndistortWrapper._undistort (wrappers.py:209-227).  Installed as the
    raster's per-pixel source map (`BatchedSimulator(distortion=True, undistort=True)`) they give the wrapped
    observation in the one render pass -- the same folding as the fisheye remap."""
    return rectify_maps(CAMERA_MATRIX, DIST_COEFS, UNDISTORT_P, (width, height))
