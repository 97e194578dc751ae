"""Observation resampling tables: Pillow's antialiased bilinear `Image.resize` restated exactly.

What learners of the reference consume is a down-scaled observation: `learning/utils/wrappers.py:38-54`
(`ResizeWrapper`, scipy `imresize` = PIL `Image.resize(..., BILINEAR)`), then `/255` and HWC->CHW
(`NormalizeWrapper :57`, `ImgWrapper :72`).  `dtsim_observe` does that on the device from the rendered
frame batch; this module builds the coefficient tables it uses and is the CPU statement of the same
arithmetic.  Pillow is importable here, so the restatement is *pinned bit-exact* against
`PIL.Image.resize` in tests/test_observe_host.py.

Algorithm (Pillow `src/libImaging/Resample.c`, 8 bits per channel): separable; per output coordinate
the triangle filter is stretched by the scale factor (support = max(scale, 1)), its taps are normalised
in double precision and rounded to 22-bit fixed point; each pass accumulates in int32 starting from
1 << 21 and stores `clip8(acc >> 22)` -- the horizontal pass writes a uint8 intermediate image, which the
vertical pass then reads (so the two roundings are part of the result).
"""
from __future__ import annotations

import math
from functools import lru_cache

import numpy as np

PRECISION_BITS = 32 - 8 - 2


@lru_cache(maxsize=None)
def coeffs(in_size: int, out_size: int):
    """(bounds int32 [out,2] = (first tap, tap count), kk int32 [out, ksize]) for one axis."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale                      # bilinear: support 1
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.array([max(1.0 - abs((x + xmin - center + 0.5) * ss), 0.0) for x in range(xmax)], dtype=np.float64)
        ww = 0.0
        for v in w:                                  # same summation order as the C loop
            ww += v
        if ww != 0.0:
            w = w / ww
        kk[xx, :xmax] = [int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS)) for v in w]
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    """One separable pass along `axis` (0 = rows / vertical, 1 = columns / horizontal); uint8 in, uint8 out."""
    in_size = img.shape[axis]
    bounds, kk = coeffs(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for o in range(out_size):
        x0, n = bounds[o]
        acc = np.tensordot(kk[o, :n].astype(np.int64), src[x0:x0 + n], axes=(0, 0)) + (1 << (PRECISION_BITS - 1))
        out[o] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bilinear(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """`np.asarray(Image.fromarray(img).resize((out_w, out_h), Image.BILINEAR))` for uint8 [H,W,C]."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W = img.shape[:2]
    tmp = _pass(img, out_w, 1) if out_w != W else img       # horizontal first (Resample.c)
    return _pass(tmp, out_h, 0) if out_h != H else tmp


def observation(frames: np.ndarray, out_h: int, out_w: int, chw: bool = False, normalize: bool = False) -> np.ndarray:
    """Host statement of `dtsim_observe`: frames uint8 [N,H,W,3] -> [N,h,w,3] (or [N,3,h,w]) uint8, or
    float32 in [0,1] (`/ 255`, NormalizeWrapper)."""
    out = np.stack([resize_bilinear(f, out_h, out_w) for f in frames])
    if chw:
        out = np.ascontiguousarray(out.transpose(0, 3, 1, 2))
    if normalize:
        out = out.astype(np.float32) / np.float32(255.0)
    return out


# ---- OpenCV INTER_CUBIC (the reference's own ResizeWrapper, wrappers.py:129-138) -----------------------------------
# cv2.resize(img, (w, h), interpolation=cv2.INTER_CUBIC) for 8-bit images, as published in OpenCV's imgproc/resize.cpp
# (resizeGeneric_ / HResizeCubic / VResizeCubic, fixed-point path): per output coordinate d
#     fx = (float)((d + 0.5) * scale - 0.5);  sx = floor(fx);  fx -= sx          (scale = in / out in double, fx in float)
#     interpolateCubic(fx), A = -0.75, in float; taps = saturate_cast<short>(c * 2048) (round half to even)
# four taps at sx - 1 .. sx + 2 with replicated borders; the horizontal pass keeps int32 rows (no rounding), the
# vertical pass sums them with its own taps and the pixel is saturate_cast<uchar>((v + (1 << 21)) >> 22).
# cv2 is not installed in this image: parity UNPINNED against the library itself (its SIMD vertical pass works in float
# and may differ from the integer path by one LSB at ties); dtsim_observe_cubic is bit-identical to THIS statement.
CV_COEF_BITS = 11


@lru_cache(maxsize=None)
def cubic_coeffs(in_size: int, out_size: int):
    """(first int32 [out] = index of the first of the 4 taps (may be < 0: borders replicate), taps int32 [out, 4])."""
    A = np.float32(-0.75)
    one, two, three = np.float32(1), np.float32(2), np.float32(3)
    scale = 1.0 / (out_size / in_size)                  # cv::resize: inv_scale_x = dsize.width / ssize.width (double), scale_x = 1. / inv_scale_x
    first = np.zeros(out_size, np.int32)
    taps = np.zeros((out_size, 4), np.int32)
    for d in range(out_size):
        fx = np.float32((d + 0.5) * scale - 0.5)
        sx = int(math.floor(float(fx)))
        x = np.float32(fx - np.float32(sx))
        c = np.empty(4, np.float32)
        x1 = np.float32(x + one)
        c[0] = ((A * x1 - np.float32(5) * A) * x1 + np.float32(8) * A) * x1 - np.float32(4) * A
        c[1] = ((A + two) * x - (A + three)) * x * x + one
        xm = np.float32(one - x)
        c[2] = ((A + two) * xm - (A + three)) * xm * xm + one
        c[3] = one - c[0] - c[1] - c[2]
        first[d] = sx - 1
        taps[d] = np.clip(np.rint((c * np.float32(1 << CV_COEF_BITS)).astype(np.float32)), -32768, 32767).astype(np.int32)
    first.setflags(write=False); taps.setflags(write=False)
    return first, taps


def resize_cubic(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_CUBIC) for uint8 [H, W] or [H, W, C] (restated; see above)."""
    img = np.ascontiguousarray(img)
    H, W = img.shape[:2]
    xi, xt = cubic_coeffs(W, out_w)
    yi, yt = cubic_coeffs(H, out_h)
    xi = xi.astype(np.int64); yi = yi.astype(np.int64)
    src = img.astype(np.int64)
    hor = np.zeros((H, out_w) + img.shape[2:], np.int64)                  # int32 rows in OpenCV, no rounding in between
    for k in range(4):
        hor += src[:, np.clip(xi + k, 0, W - 1)] * xt[:, k].astype(np.int64).reshape((1, out_w) + (1,) * (img.ndim - 2))
    out = np.zeros((out_h, out_w) + img.shape[2:], np.int64)
    for k in range(4):
        out += hor[np.clip(yi + k, 0, H - 1)] * yt[:, k].astype(np.int64).reshape((out_h, 1) + (1,) * (img.ndim - 2))
    assert np.abs(out).max(initial=0) < 2 ** 31                          # the device (and OpenCV) accumulate in int32
    return np.clip((out + (1 << (2 * CV_COEF_BITS - 1))) >> (2 * CV_COEF_BITS), 0, 255).astype(np.uint8)
