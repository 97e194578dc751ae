"""The reference's CPU render path -- Simulator._render_img's OpenGL call stream -- restated so that it can TRAVEL.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): used by bench.py's `cpu_baseline` leg and by tests; never by the product.

/root/reference cannot leave the build container (and is Python, so no build of it can), but `north_star` asks for the
reference's CPU Pyglet path timed beside the GPU on the bench box's own cores.  What that path costs is its OpenGL work on the
software renderer: this module issues the SAME GL calls in the SAME order as simulator.py:1707-1951 (+ graphics.py:172-251 for
the 4 x MSAA RGBA32F frame buffer and its resolve, objects.py:123-148 / objmesh.py:360-375 for the meshes, simulator.py:386-526
for the tile / ground vertex lists) against the same driver -- Mesa llvmpipe through oracle/gl/gl_headless.c -- from the state
an oracle/sim.py env holds.  It is pinned where the reference can run: tests/test_gl_golden.py::test_gl_port_reproduces_the_
reference_s_frames renders every GL golden's recorded state through it and requires the frames BYTE-IDENTICAL to what the
unmodified reference produced.  So `cpu_baseline.kind` stays "port", but of the reference's path on the reference's renderer,
not of a numpy software rasteriser.
"""
from __future__ import annotations

import ctypes
import itertools
import math

import numpy as np

from oracle.gl import glshim

NEAR, FAR = 0.04, 100.0
CAMERA_FORWARD_DIST = 0.066


def available() -> bool:
    return glshim.available()


class GLRenderer:
    """One headless context + the reference's frame buffers, vertex lists and textures for one oracle raster.Scene."""

    def __init__(self, scene, width: int, height: int):
        mods = glshim.install()
        self.gl = gl = mods["pyglet.gl"]
        self._graphics = mods["pyglet.graphics"]
        glshim.Window()                                  # a fresh context, as the reference's shadow window is
        self.scene, self.W, self.H = scene, int(width), int(height)
        self.multi_fbo, self.final_fbo = self._frame_buffers(self.W, self.H, 4)
        self.img = np.zeros((self.H, self.W, 3), np.uint8)
        ts = scene.m.tile_size
        # ---- simulator.py:386-526: the tile as 7 x 7 quads with uv = (pu, 1 - pv), normals +y, white c4B colours; the ground quad without normals
        ns = 8
        v, t, n, c = [], [], [], []

        def point(u_, v_):
            pu, pv = u_ / (ns - 1), v_ / (ns - 1)
            return (-ts / 2 + pu * ts, 0.0, -ts / 2 + pv * ts), (pu, 1 - pv)

        for i, j in itertools.product(range(ns - 1), range(ns - 1)):
            for (a, b) in ((i, j), (i + 1, j), (i + 1, j + 1), (i, j + 1)):
                p, uv = point(a, b)
                v.extend(p); t.extend(uv); n.extend([0.0, 1.0, 0.0]); c.extend((255, 255, 255, 255))
        self.road_vlist = self._graphics.vertex_list(len(v) // 3, ("v3f", v), ("t2f", t), ("n3f", n), ("c4B", c))
        self.ground_vlist = self._graphics.vertex_list(4, ("v3f", [-1, -0.8, 1, -1, -0.8, -1, 1, -0.8, -1, 1, -0.8, 1]))
        # ---- textures (graphics.py:69-169: pyglet's texture, then glTexImage2D(GL_RGBA, ...) of the image rows bottom-up)
        self.tile_tex = {k: self._texture(img) for k, img in scene.textures.items()}
        self.mesh_chunks = {}
        for key, m in scene.meshes.items():
            self.mesh_chunks[key] = self._mesh(m)
        cards = getattr(scene, "light_cards", None)
        self.card_tex = [self._texture(im) for im in cards] if cards else None
        gl.glEnable(gl.GL_DEPTH_TEST)

    # graphics.py:172-251
    def _frame_buffers(self, w, h, samples):
        gl = self.gl
        multi, tex, rb = ctypes.c_uint(0), ctypes.c_uint(0), ctypes.c_uint(0)
        gl.glGenFramebuffers(1, ctypes.byref(multi)); gl.glBindFramebuffer(gl.GL_FRAMEBUFFER, multi)
        gl.glGenTextures(1, ctypes.byref(tex)); gl.glBindTexture(gl.GL_TEXTURE_2D_MULTISAMPLE, tex)
        gl.glTexImage2DMultisample(gl.GL_TEXTURE_2D_MULTISAMPLE, samples, gl.GL_RGBA32F, w, h, True)
        gl.glFramebufferTexture2D(gl.GL_FRAMEBUFFER, gl.GL_COLOR_ATTACHMENT0, gl.GL_TEXTURE_2D_MULTISAMPLE, tex, 0)
        gl.glGenRenderbuffers(1, ctypes.byref(rb)); gl.glBindRenderbuffer(gl.GL_RENDERBUFFER, rb)
        gl.glRenderbufferStorageMultisample(gl.GL_RENDERBUFFER, samples, gl.GL_DEPTH_COMPONENT, w, h)
        gl.glFramebufferRenderbuffer(gl.GL_FRAMEBUFFER, gl.GL_DEPTH_ATTACHMENT, gl.GL_RENDERBUFFER, rb)
        assert gl.glCheckFramebufferStatus(gl.GL_FRAMEBUFFER) == gl.GL_FRAMEBUFFER_COMPLETE
        final, ftex = ctypes.c_uint(0), ctypes.c_uint(0)
        gl.glGenFramebuffers(1, ctypes.byref(final)); gl.glBindFramebuffer(gl.GL_FRAMEBUFFER, final)
        gl.glGenTextures(1, ctypes.byref(ftex)); gl.glBindTexture(gl.GL_TEXTURE_2D, ftex)
        gl.glTexImage2D(gl.GL_TEXTURE_2D, 0, gl.GL_RGBA, w, h, 0, gl.GL_RGBA, gl.GL_FLOAT, None)
        gl.glFramebufferTexture2D(gl.GL_FRAMEBUFFER, gl.GL_COLOR_ATTACHMENT0, gl.GL_TEXTURE_2D, ftex, 0)
        assert gl.glCheckFramebufferStatus(gl.GL_FRAMEBUFFER) == gl.GL_FRAMEBUFFER_COMPLETE
        gl.glBindFramebuffer(gl.GL_FRAMEBUFFER, 0)
        return multi, final

    def _texture(self, rgba_bottom_up):
        gl = self.gl
        img = np.ascontiguousarray(rgba_bottom_up, dtype=np.uint8)
        h, w = img.shape[:2]
        tid = ctypes.c_uint(0)
        gl.glGenTextures(1, ctypes.byref(tid)); gl.glBindTexture(gl.GL_TEXTURE_2D, tid.value)
        gl.glTexParameteri(gl.GL_TEXTURE_2D, gl.GL_TEXTURE_MIN_FILTER, gl.GL_LINEAR)
        gl.glTexParameteri(gl.GL_TEXTURE_2D, gl.GL_TEXTURE_MAG_FILTER, gl.GL_LINEAR)
        gl.glTexImage2D(gl.GL_TEXTURE_2D, 0, gl.GL_RGBA, w, h, 0, gl.GL_RGBA, gl.GL_UNSIGNED_BYTE, img.tobytes())
        return tid.value

    def _mesh(self, m):
        """objmesh.py:241-293: one vertex list (v3f, t2f, n3f, c3f) + one texture (or none) per material chunk, in draw order."""
        chunks, start = [], 0
        T = m.verts.shape[0]
        sizes = list(getattr(m, "chunk_sizes", [T])) or [T]
        texs = [self._texture(t) for t in (getattr(m, "textures", None) or [])]
        tri_tex = np.asarray(getattr(m, "tri_tex", np.full(T, -1)))
        uvs = np.asarray(getattr(m, "uvs", np.zeros((T, 3, 2), np.float32)))
        for sz in sizes:
            sl = slice(start, start + sz)
            vl = self._graphics.vertex_list(3 * sz, ("v3f", m.verts[sl].reshape(-1)), ("t2f", uvs[sl].reshape(-1)), ("n3f", m.normals[sl].reshape(-1)),
                                            ("c3f", m.colors[sl].reshape(-1)))
            ti = int(tri_tex[start]) if sz else -1
            chunks.append((vl, texs[ti] if 0 <= ti < len(texs) else None))
            start += sz
        return chunks

    def set_light(self, light_eye, ambient, diffuse):
        """reset() (simulator.py:565-584) with the position given as GL holds it (eye space): identity model-view, then glLightfv."""
        gl = self.gl
        F4 = ctypes.c_float * 4
        gl.glMatrixMode(gl.GL_MODELVIEW); gl.glLoadIdentity()
        gl.glLightfv(gl.GL_LIGHT0, gl.GL_POSITION, F4(*[float(x) for x in light_eye]))
        gl.glLightfv(gl.GL_LIGHT0, gl.GL_AMBIENT, F4(*[float(x) for x in ambient][:3], 1.0))
        gl.glLightfv(gl.GL_LIGHT0, gl.GL_DIFFUSE, F4(*[float(x) for x in diffuse][:3], 1.0))
        gl.glLightfv(gl.GL_LIGHT0, gl.GL_SPECULAR, F4(0.0, 0.0, 0.0, 1.0))
        gl.glEnable(gl.GL_LIGHTING); gl.glEnable(gl.GL_COLOR_MATERIAL)

    def render(self, pos, angle, *, cam_height=0.108, cam_angle_deg=19.15, cam_fov_y_deg=75.0, camera_noise=(0, 0, 0), domain_rand=False,
               horizon=(0.45, 0.82, 1.0), ground=(0.15, 0.15, 0.15), obj_states=None):
        """simulator.py:1707-1951, top_down = False, segment = False, no bbox / curve / LED overlays.  Returns [H,W,3] uint8, row 0 = top."""
        gl, W, H, m = self.gl, self.W, self.H, self.scene.m
        gl.glEnable(gl.GL_LIGHT0); gl.glEnable(gl.GL_LIGHTING); gl.glEnable(gl.GL_COLOR_MATERIAL)
        gl.glEnable(gl.GL_POLYGON_SMOOTH)
        gl.glLightModelfv(gl.GL_LIGHT_MODEL_AMBIENT, (ctypes.c_float * 4)(0.3, 0.3, 0.3, 1.0))
        gl.glEnable(gl.GL_MULTISAMPLE)
        gl.glBindFramebuffer(gl.GL_FRAMEBUFFER, self.multi_fbo)
        gl.glViewport(0, 0, W, H)
        gl.glClearColor(float(horizon[0]), float(horizon[1]), float(horizon[2]), 1.0)
        gl.glClearDepth(1.0)
        gl.glClear(gl.GL_COLOR_BUFFER_BIT | gl.GL_DEPTH_BUFFER_BIT)
        gl.glMatrixMode(gl.GL_PROJECTION); gl.glLoadIdentity()
        gl.gluPerspective(cam_fov_y_deg, W / float(H), NEAR, FAR)
        p = np.asarray(pos, dtype=np.float64)
        if domain_rand:
            p = p + np.asarray(camera_noise, dtype=np.float64)
        x, y, z = p
        dx, dy, dz = math.cos(angle), 0.0, -math.sin(angle)
        gl.glMatrixMode(gl.GL_MODELVIEW); gl.glLoadIdentity()
        y += cam_height
        gl.glRotatef(cam_angle_deg, 1, 0, 0); gl.glRotatef(0, 0, 1, 0); gl.glRotatef(0, 0, 0, 1)
        gl.glTranslatef(0, 0, CAMERA_FORWARD_DIST)
        gl.gluLookAt(x, y, z, x + dx, y + dy, z + dz, 0.0, 1.0, 0.0)
        # ground quad
        gl.glDisable(gl.GL_TEXTURE_2D)
        gl.glColor3f(float(ground[0]), float(ground[1]), float(ground[2]))
        gl.glPushMatrix(); gl.glScalef(50, 0.01, 50); self.ground_vlist.draw(gl.GL_QUADS); gl.glPopMatrix()
        # (the distractor triangles lie below the ground quad: never visible, and the oracle env does not keep them)
        gl.glEnable(gl.GL_TEXTURE_2D)
        gl.glTexParameteri(gl.GL_TEXTURE_2D, gl.GL_TEXTURE_MIN_FILTER, gl.GL_LINEAR)
        gl.glTexParameteri(gl.GL_TEXTURE_2D, gl.GL_TEXTURE_MAG_FILTER, gl.GL_LINEAR)
        ts = m.tile_size
        for i, j in itertools.product(range(m.grid_width), range(m.grid_height)):
            tile = m.grid[j * m.grid_width + i]
            if tile is None:
                continue
            gl.glColor4f(1.0, 1.0, 1.0, 1.0)             # (the tile colour: overridden by the vertex list's c4B array)
            gl.glPushMatrix()
            gl.glTranslatef((i + 0.5) * ts, 0, (j + 0.5) * ts)
            gl.glRotatef(tile["angle"] * 90 + 180, 0, 1, 0)
            gl.glBindTexture(gl.GL_TEXTURE_2D, self.tile_tex[tile["kind"]])
            self.road_vlist.draw(gl.GL_QUADS)
            gl.glPopMatrix()
        # objects (objects.py:123-148, objmesh.py:360-375)
        for k, o in enumerate(m.objects):
            st = obj_states[k] if obj_states is not None else None
            if st is not None and not st.get("visible", True):
                continue
            chunks = self.mesh_chunks.get(o.kind) or self.mesh_chunks.get("*")
            if chunks is None:
                continue
            op = st["pos"] if st is not None else o.pos
            gl.glPushMatrix()
            gl.glTranslatef(float(op[0]), float(op[1]), float(op[2]))
            gl.glScalef(float(o.scale), float(o.scale), float(o.scale))
            gl.glRotatef(0, 1, 0, 0); gl.glRotatef(float(st["y_rot"] if st is not None else o.y_rot), 0, 1, 0); gl.glRotatef(0, 0, 0, 1)
            gl.glColor4f(1.0, 1.0, 1.0, 1.0)
            for ci, (vl, tex) in enumerate(chunks):
                if ci == 0 and getattr(o, "light_freq", 0) > 0 and self.card_tex is not None:      # TrafficLightObj: mesh.textures[0] = the card of the pattern
                    tex = self.card_tex[int(st["light_pattern"]) if (st is not None and "light_pattern" in st) else int(o.light_pattern)]
                if tex:
                    gl.glEnable(gl.GL_TEXTURE_2D); gl.glBindTexture(gl.GL_TEXTURE_2D, tex)
                else:
                    gl.glDisable(gl.GL_TEXTURE_2D)
                vl.draw(gl.GL_TRIANGLES)
            gl.glDisable(gl.GL_TEXTURE_2D)
            gl.glPopMatrix()
        gl.glBindFramebuffer(gl.GL_READ_FRAMEBUFFER, self.multi_fbo)
        gl.glBindFramebuffer(gl.GL_DRAW_FRAMEBUFFER, self.final_fbo)
        gl.glBlitFramebuffer(0, 0, W, H, 0, 0, W, H, gl.GL_COLOR_BUFFER_BIT, gl.GL_LINEAR)
        gl.glBindFramebuffer(gl.GL_FRAMEBUFFER, self.final_fbo)
        gl.glReadPixels(0, 0, W, H, gl.GL_RGB, gl.GL_UNSIGNED_BYTE, self.img.ctypes.data)
        gl.glBindFramebuffer(gl.GL_FRAMEBUFFER, 0)
        return np.ascontiguousarray(np.flip(self.img, axis=0))
