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
