"""What does the reference's renderer (Mesa llvmpipe) do where GL leaves the arithmetic to the implementation?

    python -m oracle.gl.measure_filter > profiles/r06_gl_filter_precision.txt

TEST INFRASTRUCTURE ONLY.  Three measurements on the headless context of oracle/gl/gl_headless.c, each compared with
integer models until one is bit-identical; oracle/raster.py restates the winners:

  1. GL_LINEAR on an RGBA8 texture, magnified: weight resolution and rounding of a 1-D lerp between two texels;
  2. the 2-D filter on random 2 x 2 texel blocks: order of the lerps and precision of the intermediate;
  3. the 4 x MSAA sample positions of a GL_RGBA32F multisample FBO (glGetMultisamplefv and 0.1-pixel probe squares),
     and the rounding of the resolve blit + glReadPixels.
"""
from __future__ import annotations

import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-duckietown_amd")]
from oracle.gl import glshim  # noqa: E402

gl = glshim.install()["pyglet.gl"]


def float_fbo(w, h, samples=0):
    f, t = ctypes.c_uint(0), ctypes.c_uint(0)
    gl.glGenFramebuffers(1, ctypes.byref(f)); gl.glBindFramebuffer(gl.GL_FRAMEBUFFER, f)
    gl.glGenTextures(1, ctypes.byref(t))
    if samples:
        gl.glBindTexture(gl.GL_TEXTURE_2D_MULTISAMPLE, t)
        gl.glTexImage2DMultisample(gl.GL_TEXTURE_2D_MULTISAMPLE, samples, gl.GL_RGBA32F, w, h, True)
        gl.glFramebufferTexture2D(gl.GL_FRAMEBUFFER, gl.GL_COLOR_ATTACHMENT0, gl.GL_TEXTURE_2D_MULTISAMPLE, t, 0)
    else:
        gl.glBindTexture(gl.GL_TEXTURE_2D, t)
        gl.glTexImage2D(gl.GL_TEXTURE_2D, 0, gl.GL_RGBA32F, w, h, 0, gl.GL_RGBA, gl.GL_FLOAT, None)
        gl.glFramebufferTexture2D(gl.GL_FRAMEBUFFER, gl.GL_COLOR_ATTACHMENT0, gl.GL_TEXTURE_2D, t, 0)
    assert gl.glCheckFramebufferStatus(gl.GL_FRAMEBUFFER) == gl.GL_FRAMEBUFFER_COMPLETE
    return f


def draw_texel_cell(block, W, H, color=(1, 1, 1)):
    """A 4 x 4 RGBA8 GL_LINEAR / GL_REPEAT texture whose texels (0..1, 0..1) are `block` [2,2,3]; a quad of W x H pixels spans
    texel centre (0, 0) -> (1, 1): pixel (i, j) has filter weights ((i + 0.5) / W, (j + 0.5) / H).  Returns [H,W,3] * 255."""
    tx = np.zeros((4, 4, 4), np.uint8); tx[..., 3] = 255; tx[:2, :2, :3] = block
    tex = ctypes.c_uint(0)
    gl.glGenTextures(1, ctypes.byref(tex)); gl.glBindTexture(gl.GL_TEXTURE_2D, tex)
    gl.glTexParameteri(gl.GL_TEXTURE_2D, gl.GL_TEXTURE_MIN_FILTER, gl.GL_LINEAR); gl.glTexParameteri(gl.GL_TEXTURE_2D, gl.GL_TEXTURE_MAG_FILTER, gl.GL_LINEAR)
    gl.glTexImage2D(gl.GL_TEXTURE_2D, 0, gl.GL_RGBA, 4, 4, 0, gl.GL_RGBA, gl.GL_UNSIGNED_BYTE, tx.tobytes())
    gl.glViewport(0, 0, W, H); gl.glDisable(gl.GL_LIGHTING); gl.glDisable(gl.GL_DEPTH_TEST); gl.glEnable(gl.GL_TEXTURE_2D)
    gl.glMatrixMode(gl.GL_PROJECTION); gl.glLoadIdentity(); gl.glOrtho(0, 1, 0, 1, -1, 1); gl.glMatrixMode(gl.GL_MODELVIEW); gl.glLoadIdentity()
    gl.glClearColor(0, 0, 0, 1); gl.glClear(gl.GL_COLOR_BUFFER_BIT); gl.glColor4f(*color, 1)
    u0, u1 = 0.5 / 4, 1.5 / 4
    gl.glBegin(gl.GL_QUADS)
    for x, y, u, v in ((0, 0, u0, u0), (1, 0, u1, u0), (1, 1, u1, u1), (0, 1, u0, u1)):
        gl.glTexCoord2f(u, v); gl.glVertex3f(x, y, 0)
    gl.glEnd()
    out = np.zeros((H, W, 4), np.float32)
    gl.glReadPixels(0, 0, W, H, gl.GL_RGBA, gl.GL_FLOAT, out.ctypes.data)
    gl.glDeleteTextures(1, ctypes.byref(tex))
    return out[..., :3] * 255


def lerp8(w, p, q):
    return p + ((w * (q - p) + 128) >> 8)


def main():
    print("renderer:", glshim.renderer())
    # ---- 1. one-dimensional lerp
    W = 4096
    float_fbo(W, 2)
    w_true = (np.arange(W) + 0.5) / W
    print("\n1. GL_LINEAR between two RGBA8 texels a -> b over", W, "pixels (8 pixels per 1/256 of weight)")
    print("   models: weight wq = floor(256 w) | round(256 w);  value = a + ((wq (b - a)) >> 8) [floor] | a + ((wq (b - a) + 128) >> 8) [round] | float lerp, rounded [exact]")
    for a, b in ((0, 16), (16, 0), (100, 101), (37, 203), (203, 37), (0, 255), (255, 0)):
        blk = np.zeros((2, 2, 3), np.uint8); blk[:, 0] = a; blk[:, 1] = b
        r = draw_texel_cell(blk, W, 2)[1, :, 0]
        ri = np.round(r).astype(np.int64)
        ch = (np.nonzero(np.diff(ri))[0] + 1)[:4] / W * 256
        res = {}
        for wname, wq in (("floor", np.floor(w_true * 256).astype(np.int64)), ("round", np.floor(w_true * 256 + 0.5).astype(np.int64))):
            res[f"w {wname} / floor"] = int((a + ((wq * (b - a)) >> 8) != ri).sum())
            res[f"w {wname} / round"] = int((lerp8(wq, a, b) != ri).sum())
        res["exact"] = int((np.round(a + w_true * (b - a)).astype(np.int64) != ri).sum())
        print(f"   {a:3d} -> {b:3d}: {len(np.unique(ri)):3d} levels, all integers to {np.abs(r - ri).max():.1e}; first steps at 256 w = {np.round(ch, 2)}; mismatching pixels: {res}")
    # ---- 2. two-dimensional
    W = H = 512
    float_fbo(W, H)
    wq = np.floor((np.arange(W) + 0.5) / W * 256 + 0.5).astype(np.int64)
    wx, wy = wq[None, :, None], wq[:, None, None]
    rng = np.random.default_rng(0)
    print("\n2. random 2 x 2 texel blocks, 512 x 512 pixels per cell (weights round(256 w)); mismatching values of 786 432:")
    for trial in range(4):
        blk = rng.integers(0, 256, (2, 2, 3)).astype(np.uint8)
        ri = np.round(draw_texel_cell(blk, W, H)).astype(np.int64)
        a, b, c, d = [blk[i, j].astype(np.int64)[None, None, :] for i, j in ((0, 0), (0, 1), (1, 0), (1, 1))]
        s_then_t = lerp8(wy, lerp8(wx, a, b), lerp8(wx, c, d))
        t_then_s = lerp8(wx, lerp8(wy, a, c), lerp8(wy, b, d))
        one_round = (a * (256 - wx) * (256 - wy) + b * wx * (256 - wy) + c * (256 - wx) * wy + d * wx * wy + 32768) >> 16
        fx, fy = ((np.arange(W) + 0.5) / W)[None, :, None], ((np.arange(H) + 0.5) / H)[:, None, None]
        exact = np.round((a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy).astype(np.int64)
        print(f"   block {trial}: lerp along s, then t, 8-bit intermediate: {int((s_then_t != ri).sum())};  t then s: {int((t_then_s != ri).sum())};"
              f"  16-bit weights, one rounding: {int((one_round != ri).sum())};  float bilinear, rounded: {int((exact != ri).sum())}")
    r = draw_texel_cell(np.full((2, 2, 3), 200, np.uint8), 8, 8, (0.5, 0.3, 0.7))[4, 4]
    print(f"   GL_MODULATE of the 8-bit sample 200 with the colour (0.5, 0.3, 0.7): {r} (float: the sampler's result is 8-bit, the texture environment is not)")
    # ---- 3. sample positions, resolve
    W = H = 8
    fm = float_fbo(W, H, 4)
    buf = (ctypes.c_float * 2)()
    pos = []
    for i in range(4):
        gl.glGetMultisamplefv(gl.GL_SAMPLE_POSITION, i, buf); pos.append((buf[0], buf[1]))
    print("\n3. GL_SAMPLE_POSITION (window coordinates, +y up):", pos)
    f2, t2 = ctypes.c_uint(0), ctypes.c_uint(0)
    gl.glGenFramebuffers(1, ctypes.byref(f2)); gl.glBindFramebuffer(gl.GL_FRAMEBUFFER, f2)
    gl.glGenTextures(1, ctypes.byref(t2)); gl.glBindTexture(gl.GL_TEXTURE_2D, t2)
    gl.glTexImage2D(gl.GL_TEXTURE_2D, 0, gl.GL_RGBA, W, H, 0, gl.GL_RGBA, gl.GL_FLOAT, None)     # the reference's resolve target (graphics.py:232)
    gl.glFramebufferTexture2D(gl.GL_FRAMEBUFFER, gl.GL_COLOR_ATTACHMENT0, gl.GL_TEXTURE_2D, t2, 0)

    def probe(cx, cy, level=1.0, hs=0.05):
        gl.glBindFramebuffer(gl.GL_FRAMEBUFFER, fm)
        gl.glEnable(gl.GL_MULTISAMPLE); gl.glDisable(gl.GL_LIGHTING); gl.glDisable(gl.GL_DEPTH_TEST); gl.glDisable(gl.GL_TEXTURE_2D)
        gl.glViewport(0, 0, W, H)
        gl.glMatrixMode(gl.GL_PROJECTION); gl.glLoadIdentity(); gl.glOrtho(0, W, 0, H, -1, 1); gl.glMatrixMode(gl.GL_MODELVIEW); gl.glLoadIdentity()
        gl.glClearColor(0, 0, 0, 1); gl.glClear(gl.GL_COLOR_BUFFER_BIT); gl.glColor4f(level, level, level, 1)
        gl.glBegin(gl.GL_QUADS)
        for dx, dy in ((-hs, -hs), (hs, -hs), (hs, hs), (-hs, hs)):
            gl.glVertex3f(3 + cx + dx, 2 + cy + dy, 0)
        gl.glEnd()
        gl.glBindFramebuffer(gl.GL_READ_FRAMEBUFFER, fm); gl.glBindFramebuffer(gl.GL_DRAW_FRAMEBUFFER, f2)
        gl.glBlitFramebuffer(0, 0, W, H, 0, 0, W, H, gl.GL_COLOR_BUFFER_BIT, gl.GL_LINEAR)
        gl.glBindFramebuffer(gl.GL_FRAMEBUFFER, f2)
        out = np.zeros((H, W, 3), np.uint8)
        gl.glReadPixels(0, 0, W, H, gl.GL_RGB, gl.GL_UNSIGNED_BYTE, out.ctypes.data)
        return int(out[2, 3, 0])

    for cx, cy in pos:
        print(f"   0.1-px square at ({cx}, {cy}) of pixel (3, 2): resolved {probe(cx, cy)} / 255;  mirrored in y ({cx}, {1 - cy}): {probe(cx, 1 - cy)}")
    lv = [(v, probe(pos[0][0], pos[0][1], v)) for v in (0.1, 0.3, 0.5, 0.7, 0.9, 0.99)]
    print("   one covered sample of four at float level v -> byte:", lv, "; round(255 v / 4):", [int(np.floor(255 * v / 4 + 0.5)) for v, _ in lv])


if __name__ == "__main__":
    main()
