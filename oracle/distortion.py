"""Oracle restatement of Distortion (src/gym_duckietown/distortion.py) and of the three
OpenCV calls it makes.  TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

OpenCV is not installed and not vendored under /root/reference (opencv-python, unpinned):
PARITY UNPINNED for getOptimalNewCameraMatrix / initUndistortRectifyMap / remap.  They are
restated here from OpenCV's published plumb-bob model, in the scalar loop order of the C
sources (calib3d/calibration.cpp cvGetOptimalNewCameraMatrix + icvGetRectangles,
imgproc/undistort.dispatch.cpp initUndistortRectifyMap + cvUndistortPointsInternal).
_invert_map / _fill_holes follow distortion.py:138-265 statement by statement and are pinned
against the reference's own code in tests/test_oracle_vs_reference.py.
"""
from __future__ import annotations

import itertools

import numpy as np

W0, H0 = 640, 480
K = np.reshape([305.5718893575089, 0, 303.0797142544728, 0, 308.8338858195428, 231.8845403702499, 0, 0, 1], (3, 3))
D = [-0.2, 0.0305, 0.0005859930422629722, -0.0006697840226199427, 0]


def _undistort_point(u, v):
    """cvUndistortPointsInternal, one point, R = P = I, 5 iterations."""
    k1, k2, p1, p2, k3 = D
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    ifx, ify = 1.0 / fx, 1.0 / fy
    x = x0 = (u - cx) * ifx
    y = y0 = (v - cy) * ify
    for _ in range(5):
        r2 = x * x + y * y
        icdist = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2)
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x = (x0 - dx) * icdist
        y = (y0 - dy) * icdist
    return np.float32(x), np.float32(y)


def new_camera_matrix():
    """cv2.getOptimalNewCameraMatrix(K, D, (640,480), alpha=0) (distortion.py:51-56)."""
    N = 9
    f32 = np.float32
    pts = [[None] * N for _ in range(N)]
    for y in range(N):
        for x in range(N):
            pts[y][x] = _undistort_point(float(f32(x) * f32(W0) / f32(N - 1)), float(f32(y) * f32(H0) / f32(N - 1)))
    iX0, iX1, iY0, iY1 = f32(-3.4e38), f32(3.4e38), f32(-3.4e38), f32(3.4e38)
    for y in range(N):
        for x in range(N):
            px, py = pts[y][x]
            if x == 0:
                iX0 = max(iX0, px)
            if x == N - 1:
                iX1 = min(iX1, px)
            if y == 0:
                iY0 = max(iY0, py)
            if y == N - 1:
                iY1 = min(iY1, py)
    inner = (iX0, iY0, f32(iX1 - iX0), f32(iY1 - iY0))
    fx0 = (W0 - 1) / float(inner[2])
    fy0 = (H0 - 1) / float(inner[3])
    M = np.array(K, dtype=np.float64)
    M[0, 0], M[1, 1] = fx0, fy0          # alpha = 0: the inner rectangle only
    M[0, 2], M[1, 2] = -fx0 * float(inner[0]), -fy0 * float(inner[1])
    return M


def rectify_maps(w, h):
    """cv2.initUndistortRectifyMap(K, D, I, new_K, (w,h), CV_32FC1) (distortion.py:100-107)."""
    k1, k2, p1, p2, k3 = D
    fx, fy, u0, v0 = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    ir = np.linalg.inv(new_camera_matrix()).reshape(-1)
    mapx = np.zeros((h, w), np.float32)
    mapy = np.zeros((h, w), np.float32)
    for i in range(h):
        _x = i * ir[1] + ir[2]
        _y = i * ir[4] + ir[5]
        _w = i * ir[7] + ir[8]
        for j in range(w):
            ww = 1.0 / _w
            x, y = _x * ww, _y * ww
            x2, y2 = x * x, y * y
            r2, _2xy = x2 + y2, 2 * x * y
            kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
            xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2)
            yd = y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy
            mapx[i, j] = fx * xd + u0
            mapy[i, j] = fy * yd + v0
            _x += ir[0]
            _y += ir[3]
            _w += ir[6]
    return mapx, mapy


def invert_map(mapx, mapy):
    """distortion.py:138-216"""
    H, W = mapx.shape[0:2]
    rmapx = np.empty_like(mapx); rmapx.fill(np.nan)
    rmapy = np.empty_like(mapx); rmapy.fill(np.nan)
    around_rmapx = np.zeros((H, W), "float32")
    around_rmapy = np.zeros((H, W), "float32")
    around = np.zeros((H, W), "float32")
    deltas = [(-1, -1, 7), (-1, 0, 10), (-1, +1, 7), (0, -1, 10), (0, 0, 20), (0, +1, 10), (+1, -1, 7), (+1, 0, 10), (+1, +1, 7)]
    mapx_disc = np.clip(mapx.astype("int32"), 2, W - 2)
    mapy_disc = np.clip(mapy.astype("int32"), 2, H - 2)
    xs = np.zeros((H, W), "int32"); ys = np.zeros((H, W), "int32")
    for j in range(W):
        xs[:, j] = j
    for i in range(H):
        ys[i, :] = i
    for di, dj, w in deltas:
        a, b = mapy_disc + di, mapx_disc + dj
        around[a, b] += w                 # fancy-index +=: duplicates do NOT accumulate
        around_rmapx[a, b] += w * xs
        around_rmapy[a, b] += w * ys
    nonzero = around > 0
    rmapx[nonzero] = around_rmapx[nonzero] / around[nonzero]
    rmapy[nonzero] = around_rmapy[nonzero] / around[nonzero]
    fill_holes(rmapx, rmapy)
    return rmapx, rmapy


def fill_holes(rmapx, rmapy):
    """distortion.py:218-265, incl. the set-iteration order (CPython >= 3.8 deterministic)."""
    H, W = rmapx.shape[0:2]
    R = 2
    F = R * 2 + 1
    deltas0 = [(i - R - 1, j - R - 1) for i, j in itertools.product(range(F), range(F))]
    deltas0 = [x for x in deltas0 if np.hypot(x[0], x[1]) <= R]
    deltas0.sort(key=lambda x: np.hypot(x[0], x[1]))
    holes = set()
    for i, j in itertools.product(range(H), range(W)):
        if np.isnan(rmapx[i, j]):
            holes.add((i, j))
    while holes:
        nholes_filled = 0
        for i, j in list(holes):
            for di, dj in deltas0:
                u, v = i + di, j + dj
                if (0 <= u < H) and (0 <= v < W):
                    if not np.isnan(rmapx[u, v]):
                        rmapx[i, j] = rmapx[u, v]
                        rmapy[i, j] = rmapy[u, v]
                        nholes_filled += 1
                        holes.remove((i, j))
                        break
        if nholes_filled == 0:
            break


def distortion_maps(w, h):
    """rmapx, rmapy of Distortion.distort for a (h, w) observation."""
    return invert_map(*rectify_maps(w, h))
