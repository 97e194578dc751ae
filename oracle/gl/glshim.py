"""A minimal `pyglet` over a headless Mesa context -- enough for the REFERENCE's render path to run unmodified.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): nothing under gym-duckietown_amd/ imports this.

pyglet is absent from the image; the GL it would drive is not (oracle/gl/gl_headless.c: Mesa 23.2.1 llvmpipe through the
DRI swrast driver, no X).  This module provides the slice of pyglet's API the reference's simulator.py / graphics.py /
objmesh.py / objects.py / check_hw.py touch, each piece doing what pyglet 1.4/1.5 (requirements.pin.txt:1) does with GL:

  pyglet.gl           ctypes prototypes for the ~70 GL entry points involved, every GL_* enum of <GL/gl.h> / <GL/glext.h>,
                      GLfloat / GLuint / GLubyte, and GLU's gluPerspective / gluLookAt / gluNewQuadric / gluSphere restated from
                      the SGI libGLU sources Mesa ships (libutil/project.c, libutil/quad.c): float matrices through glMultMatrixf /
                      glMultMatrixd, the sphere as triangle fans at the poles + quad strips with glNormal3f per vertex.
  pyglet.graphics     vertex_list(n, ("v3f", data), ("t2f", ...), ("n3f", ...), ("c3f" | "c4B", ...)).draw(mode): client arrays +
                      glDrawArrays inside glPushClientAttrib / glPopClientAttrib.  Arrays the list does not carry are NOT enabled,
                      so the draw uses GL's *current* normal / colour for them (the ground quad has no normals, simulator.py:526).
  pyglet.image        load(path) (PIL decode, rows bottom-up as pyglet stores them), ImageData(w, h, fmt, data, pitch),
                      .get_texture() (glGenTextures + GL_LINEAR min / mag, as Texture.create does + upload), .get_image_data().get_data(fmt, pitch).
  pyglet.window       Window(...) = the one headless context; switch_to() = make current.
  pyglet.text.Label   inert.

`install(width, height)` builds the context (once per process) and returns the dict of modules to put in sys.modules.
LP_NUM_THREADS=1 is exported before the driver loads so that llvmpipe rasterises on one thread (deterministic, and "one core"
for timing).
"""
from __future__ import annotations

import ctypes
import math
import os
import re
import subprocess
import types
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_ubyte, c_uint, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libglheadless.so")
DRI_DRIVER = os.path.join(os.environ.get("GLH_DRI_DIR", "/usr/lib/x86_64-linux-gnu/dri"), "swrast_dri.so")

_lib = None
_gl = None


def available() -> bool:
    """Can a headless llvmpipe context be made here?  (driver + headers + gcc or a prebuilt helper)"""
    return os.path.isfile(DRI_DRIVER) and (os.path.isfile(LIB) or os.path.isfile("/usr/include/GL/internal/dri_interface.h"))


def build() -> str:
    """Compile the helper with the committed recipe (oracle/gl/Makefile) if it is missing or stale."""
    src = os.path.join(HERE, "gl_headless.c")
    if not os.path.isfile(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.run(["make", "-C", HERE, "-s", "libglheadless.so"], check=True)
    return LIB


def _context(width: int, height: int):
    global _lib
    if _lib is None:
        os.environ.setdefault("LP_NUM_THREADS", "1")
        lib = ctypes.CDLL(build())
        lib.glh_error.restype = c_char_p
        lib.glh_get_proc.restype = c_void_p
        lib.glh_get_proc.argtypes = [c_char_p]
        rc = lib.glh_init(int(width), int(height))
        if rc != 0:
            raise RuntimeError(f"headless GL context failed at stage {rc}: {lib.glh_error().decode()}")
        _lib = lib
    return _lib


# ---------------------------------------------------------------------------------------------- GL prototypes ----
E, I, U, F, D, P, B = c_uint, c_int, c_uint, c_float, c_double, c_void_p, c_ubyte
_PROTOS = {
    # state
    "glEnable": (None, [E]), "glDisable": (None, [E]), "glGetError": (E, []), "glGetString": (c_void_p, [E]),
    "glGetFloatv": (None, [E, POINTER(F)]), "glGetDoublev": (None, [E, POINTER(D)]), "glGetIntegerv": (None, [E, POINTER(I)]),
    "glGetBooleanv": (None, [E, POINTER(B)]), "glIsEnabled": (B, [E]),
    "glPushAttrib": (None, [U]), "glPopAttrib": (None, []), "glPushClientAttrib": (None, [U]), "glPopClientAttrib": (None, []),
    "glFlush": (None, []), "glFinish": (None, []), "glHint": (None, [E, E]), "glPixelStorei": (None, [E, I]),
    "glShadeModel": (None, [E]), "glDepthFunc": (None, [E]), "glDepthMask": (None, [B]), "glCullFace": (None, [E]), "glFrontFace": (None, [E]),
    "glLineWidth": (None, [F]), "glPointSize": (None, [F]), "glBlendFunc": (None, [E, E]), "glPolygonMode": (None, [E, E]),
    # matrices
    "glMatrixMode": (None, [E]), "glLoadIdentity": (None, []), "glPushMatrix": (None, []), "glPopMatrix": (None, []),
    "glRotatef": (None, [F, F, F, F]), "glTranslatef": (None, [F, F, F]), "glTranslated": (None, [D, D, D]), "glScalef": (None, [F, F, F]),
    "glMultMatrixf": (None, [POINTER(F)]), "glMultMatrixd": (None, [POINTER(D)]), "glLoadMatrixf": (None, [POINTER(F)]),
    "glLoadMatrixd": (None, [POINTER(D)]), "glOrtho": (None, [D] * 6), "glFrustum": (None, [D] * 6), "glViewport": (None, [I, I, I, I]),
    # lighting / material
    "glLightfv": (None, [E, E, POINTER(F)]), "glLightf": (None, [E, E, F]), "glLightModelfv": (None, [E, POINTER(F)]),
    "glGetLightfv": (None, [E, E, POINTER(F)]), "glMaterialfv": (None, [E, E, POINTER(F)]), "glColorMaterial": (None, [E, E]),
    # immediate mode
    "glBegin": (None, [E]), "glEnd": (None, []), "glVertex3f": (None, [F, F, F]), "glVertex2f": (None, [F, F]), "glNormal3f": (None, [F, F, F]),
    "glColor3f": (None, [F, F, F]), "glColor4f": (None, [F, F, F, F]), "glTexCoord2f": (None, [F, F]),
    # client arrays
    "glEnableClientState": (None, [E]), "glDisableClientState": (None, [E]), "glVertexPointer": (None, [I, E, I, P]),
    "glNormalPointer": (None, [E, I, P]), "glColorPointer": (None, [I, E, I, P]), "glTexCoordPointer": (None, [I, E, I, P]),
    "glDrawArrays": (None, [E, I, I]),
    # clears / reads
    "glClearColor": (None, [F, F, F, F]), "glClearDepth": (None, [D]), "glClear": (None, [U]),
    "glReadPixels": (None, [I, I, I, I, E, E, P]), "glReadBuffer": (None, [E]), "glDrawBuffer": (None, [E]),
    # textures
    "glGenTextures": (None, [I, POINTER(U)]), "glDeleteTextures": (None, [I, POINTER(U)]), "glBindTexture": (None, [E, U]),
    "glTexParameteri": (None, [E, E, I]), "glTexParameterf": (None, [E, E, F]), "glTexEnvi": (None, [E, E, I]), "glTexEnvf": (None, [E, E, F]),
    "glTexImage2D": (None, [E, I, I, I, I, I, E, E, P]), "glTexSubImage2D": (None, [E, I, I, I, I, I, E, E, P]),
    "glGetTexImage": (None, [E, I, E, E, P]), "glGetTexLevelParameteriv": (None, [E, I, E, POINTER(I)]), "glGetTexParameteriv": (None, [E, E, POINTER(I)]),
    "glTexImage2DMultisample": (None, [E, I, E, I, I, B]), "glGetMultisamplefv": (None, [E, U, POINTER(F)]),
    # framebuffer objects
    "glGenFramebuffers": (None, [I, POINTER(U)]), "glBindFramebuffer": (None, [E, U]), "glFramebufferTexture2D": (None, [E, E, E, U, I]),
    "glGenRenderbuffers": (None, [I, POINTER(U)]), "glBindRenderbuffer": (None, [E, U]), "glRenderbufferStorage": (None, [E, E, I, I]),
    "glRenderbufferStorageMultisample": (None, [E, I, E, I, I]), "glFramebufferRenderbuffer": (None, [E, E, E, U]),
    "glCheckFramebufferStatus": (E, [E]), "glBlitFramebuffer": (None, [I] * 8 + [U, E]),
    "glDeleteFramebuffers": (None, [I, POINTER(U)]), "glDeleteRenderbuffers": (None, [I, POINTER(U)]),
}


def _enums() -> dict:
    out = {}
    pat = re.compile(r"^#define\s+(GL_[A-Za-z0-9_]+)\s+(0x[0-9A-Fa-f]+|\d+)u?\s*$")
    for hdr in ("/usr/include/GL/gl.h", "/usr/include/GL/glext.h"):
        with open(hdr) as f:
            for line in f:
                m = pat.match(line)
                if m and m.group(1) not in out:
                    out[m.group(1)] = int(m.group(2), 0)
    return out


def _as_double(v):
    """What ctypes' c_double does with an argument: float(v) -- which also takes the 1-element arrays the reference's
    randomiser hands out (cam_fov_y, cam_height under domain randomisation)."""
    return c_double(v).value


class _Quadric:
    """GLUquadric defaults (quad.c gluNewQuadric): GLU_SMOOTH normals, GLU_OUTSIDE, GLU_FILL, no texture coordinates."""


def _make_gl(lib) -> types.ModuleType:
    gl = types.ModuleType("pyglet.gl")
    gl.__dict__.update(_enums())
    gl.GLfloat, gl.GLdouble, gl.GLuint, gl.GLint, gl.GLubyte, gl.GLenum, gl.GLsizei = c_float, c_double, c_uint, c_int, c_ubyte, c_uint, c_int
    missing = []
    for name, (res, args) in _PROTOS.items():
        addr = lib.glh_get_proc(name.encode())
        if not addr:
            missing.append(name)
            continue
        fn = ctypes.CFUNCTYPE(res, *args)(addr)
        fn.__name__ = name
        setattr(gl, name, fn)
    if missing:
        raise RuntimeError(f"GL entry points not exported by the driver: {missing}")

    # ---- GLU (libGLU is absent; SGI libutil restated) ----
    def gluPerspective(fovy, aspect, zNear, zFar):
        """project.c gluPerspective: double matrix through glMultMatrixd."""
        fovy, aspect, zNear, zFar = _as_double(fovy), _as_double(aspect), _as_double(zNear), _as_double(zFar)
        radians = fovy / 2 * math.pi / 180
        deltaZ = zFar - zNear
        sine = math.sin(radians)
        if deltaZ == 0 or sine == 0 or aspect == 0:
            return
        cotangent = math.cos(radians) / sine
        m = [0.0] * 16                                   # column-major
        m[0] = cotangent / aspect
        m[5] = cotangent
        m[10] = -(zFar + zNear) / deltaZ
        m[11] = -1.0
        m[14] = -2 * zNear * zFar / deltaZ
        m[15] = 0.0
        gl.glMultMatrixd((c_double * 16)(*m))

    def gluLookAt(ex, ey, ez, cx, cy, cz, ux, uy, uz):
        """project.c gluLookAt: forward / side / up in FLOAT, glMultMatrixf, then glTranslated(-eye)."""
        import numpy as np
        f32 = np.float32
        ex, ey, ez, cx, cy, cz, ux, uy, uz = (_as_double(v) for v in (ex, ey, ez, cx, cy, cz, ux, uy, uz))
        fwd = np.array([cx - ex, cy - ey, cz - ez], dtype=f32)     # (double differences stored into GLfloat)
        up = np.array([ux, uy, uz], dtype=f32)

        def normalize(v):
            r = f32(math.sqrt(float(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])))   # sqrt() of a float expression, stored to float
            return v if r == 0 else (v / r).astype(f32)

        def cross(a, b):
            return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], dtype=f32)

        fwd = normalize(fwd)
        side = normalize(cross(fwd, up))
        up = cross(side, fwd)
        m = [float(side[0]), float(up[0]), float(-fwd[0]), 0.0,
             float(side[1]), float(up[1]), float(-fwd[1]), 0.0,
             float(side[2]), float(up[2]), float(-fwd[2]), 0.0,
             0.0, 0.0, 0.0, 1.0]
        gl.glMultMatrixf((c_float * 16)(*m))
        gl.glTranslated(-float(ex), -float(ey), -float(ez))

    def gluNewQuadric():
        return _Quadric()

    def gluSphere(qobj, radius, slices, stacks):
        """quad.c gluSphere for the default quadric (GLU_FILL, GLU_SMOOTH, GLU_OUTSIDE, no texture): sin / cos tables in float,
        stack 0 as a GL_TRIANGLE_FAN around (0, 0, r), the last stack as a fan around (0, 0, -r), quad strips between."""
        import numpy as np
        f32 = np.float32
        radius = float(radius)
        sinC = [f32(math.sin(2 * math.pi * i / slices)) for i in range(slices)] + [None]
        cosC = [f32(math.cos(2 * math.pi * i / slices)) for i in range(slices)] + [None]
        sinC[slices], cosC[slices] = sinC[0], cosC[0]
        sinC2 = [f32(math.sin(math.pi * j / stacks)) for j in range(stacks + 1)]     # GLU_OUTSIDE: normals = +direction
        cosC2 = [f32(math.cos(math.pi * j / stacks)) for j in range(stacks + 1)]
        sinC1 = [f32(radius * math.sin(math.pi * j / stacks)) for j in range(stacks + 1)]
        cosC1 = [f32(radius * math.cos(math.pi * j / stacks)) for j in range(stacks + 1)]
        sinC1[0] = sinC1[stacks] = f32(0)                # "Make sure it comes to a point"
        cosC1[0], cosC1[stacks] = f32(radius), f32(-radius)
        # top cap (stack 0 -> 1): fan, slices run backwards so that the winding faces outward
        sT1, cT1 = sinC1[1], cosC1[1]
        sT2, cT2 = sinC2[1], cosC2[1]
        gl.glBegin(gl.GL_TRIANGLE_FAN)
        gl.glNormal3f(float(sinC[0] * sinC2[0]), float(cosC[0] * sinC2[0]), float(cosC2[0]))
        gl.glVertex3f(0.0, 0.0, radius)
        for i in range(slices, -1, -1):
            gl.glNormal3f(float(sinC[i] * sT2), float(cosC[i] * sT2), float(cT2))
            gl.glVertex3f(float(sT1 * sinC[i]), float(sT1 * cosC[i]), float(cT1))
        gl.glEnd()
        # bottom cap (stack stacks-1 -> stacks)
        sT1, cT1 = sinC1[stacks - 1], cosC1[stacks - 1]
        sT2, cT2 = sinC2[stacks - 1], cosC2[stacks - 1]
        gl.glBegin(gl.GL_TRIANGLE_FAN)
        gl.glNormal3f(float(sinC[stacks] * sinC2[stacks]), float(cosC[stacks] * sinC2[stacks]), float(cosC2[stacks]))
        gl.glVertex3f(0.0, 0.0, -radius)
        for i in range(0, slices + 1):
            gl.glNormal3f(float(sinC[i] * sT2), float(cosC[i] * sT2), float(cT2))
            gl.glVertex3f(float(sT1 * sinC[i]), float(sT1 * cosC[i]), float(cT1))
        gl.glEnd()
        # the stacks between
        for j in range(1, stacks - 1):
            sA1, cA1, sB1, cB1 = sinC1[j], cosC1[j], sinC1[j + 1], cosC1[j + 1]
            sA2, cA2, sB2, cB2 = sinC2[j], cosC2[j], sinC2[j + 1], cosC2[j + 1]
            gl.glBegin(gl.GL_QUAD_STRIP)
            for i in range(slices + 1):
                gl.glNormal3f(float(sinC[i] * sB2), float(cosC[i] * sB2), float(cB2))
                gl.glVertex3f(float(sB1 * sinC[i]), float(sB1 * cosC[i]), float(cB1))
                gl.glNormal3f(float(sinC[i] * sA2), float(cosC[i] * sA2), float(cA2))
                gl.glVertex3f(float(sA1 * sinC[i]), float(sA1 * cosC[i]), float(cA1))
            gl.glEnd()

    gl.gluPerspective, gl.gluLookAt, gl.gluNewQuadric, gl.gluSphere = gluPerspective, gluLookAt, gluNewQuadric, gluSphere

    class Config:
        def __init__(self, **kw):
            self.__dict__.update(kw)

    gl.Config = Config
    return gl


# ---------------------------------------------------------------------------------------------- pyglet.graphics ----
_ATTR = re.compile(r"^([vtnc])(\d)([fBdis])(?:/\w+)?$")
_CT = {"f": (c_float, "GL_FLOAT"), "B": (c_ubyte, "GL_UNSIGNED_BYTE"), "d": (c_double, "GL_DOUBLE"), "i": (c_int, "GL_INT")}


class VertexList:
    """pyglet.graphics.vertex_list(count, *(format, data)): interleaving is pyglet's business, not GL's -- one tight client array
    per attribute, enabled for the draw only (vertexdomain.VertexList.draw: glPushClientAttrib(GL_CLIENT_VERTEX_ARRAY_BIT),
    enable + set_pointer per attribute, glDrawArrays(mode, start, count), glPopClientAttrib)."""

    def __init__(self, gl, count, *attrs):
        self.gl, self.count, self.attrs = gl, int(count), []
        for fmt, data in attrs:
            m = _ATTR.match(fmt)
            if not m:
                raise ValueError(f"vertex format {fmt!r} not handled by the shim")
            kind, n, t = m.group(1), int(m.group(2)), m.group(3)
            ct, glt = _CT[t]
            vals = list(data)
            if len(vals) != n * self.count:
                raise ValueError(f"{fmt}: {len(vals)} values for {self.count} vertices")
            conv = int if t in "Bi" else float
            arr = (ct * len(vals))(*[conv(v) for v in vals])
            self.attrs.append((kind, n, getattr(gl, glt), arr))

    def draw(self, mode):
        gl = self.gl
        gl.glPushClientAttrib(gl.GL_CLIENT_VERTEX_ARRAY_BIT)
        for kind, n, glt, arr in self.attrs:
            if kind == "v":
                gl.glEnableClientState(gl.GL_VERTEX_ARRAY); gl.glVertexPointer(n, glt, 0, arr)
            elif kind == "t":
                gl.glEnableClientState(gl.GL_TEXTURE_COORD_ARRAY); gl.glTexCoordPointer(n, glt, 0, arr)
            elif kind == "n":
                gl.glEnableClientState(gl.GL_NORMAL_ARRAY); gl.glNormalPointer(glt, 0, arr)
            elif kind == "c":
                gl.glEnableClientState(gl.GL_COLOR_ARRAY); gl.glColorPointer(n, glt, 0, arr)
        gl.glDrawArrays(mode, 0, self.count)
        gl.glPopClientAttrib()

    def delete(self):
        pass


# ---------------------------------------------------------------------------------------------- pyglet.image ----
_CH = {"L": 1, "LA": 2, "RGB": 3, "RGBA": 4, "BGR": 3, "BGRA": 4}


class Texture:
    def __init__(self, target, tid, width, height):
        self.target, self.id, self.width, self.height = target, tid, width, height


class ImageData:
    """pyglet.image.ImageData: `pitch` > 0 = rows bottom-up (pyglet's native order), < 0 = top-down."""

    def __init__(self, width, height, fmt=None, data=None, pitch=None, *, format=None):   # noqa: A002 (pyglet's keyword)
        self.width, self.height = int(width), int(height)
        self.format = fmt if fmt is not None else format
        self.pitch = pitch if pitch is not None else self.width * len(self.format)
        self._data = bytes(data) if not isinstance(data, bytes) else data

    def _rgba_bottom_up(self):
        import numpy as np
        nch = len(self.format)
        a = np.frombuffer(self._data, dtype=np.uint8)
        rows = a[: self.height * abs(self.pitch)].reshape(self.height, abs(self.pitch))[:, : self.width * nch].reshape(self.height, self.width, nch)
        if self.pitch < 0:
            rows = rows[::-1]
        out = np.empty((self.height, self.width, 4), np.uint8)
        out[..., 3] = 255
        for k, ch in enumerate(self.format):
            if ch == "L":
                out[..., 0] = out[..., 1] = out[..., 2] = rows[..., k]
            else:
                out[..., "RGBA".index(ch)] = rows[..., k]
        return out

    def get_image_data(self):
        return self

    def get_data(self, fmt, pitch):
        import numpy as np
        rgba = self._rgba_bottom_up()
        sel = np.stack([rgba[..., "RGBA".index(ch)] for ch in fmt], axis=-1)
        if pitch < 0:
            sel = sel[::-1]
        if abs(pitch) != self.width * len(fmt):
            raise ValueError("padded pitches are not handled by the shim")
        return np.ascontiguousarray(sel).tobytes()

    def get_texture(self, rectangle=False, force_rectangle=False):
        gl = _gl
        tid = c_uint(0)
        gl.glGenTextures(1, ctypes.byref(tid))
        gl.glBindTexture(gl.GL_TEXTURE_2D, tid.value)
        gl.glTexParameteri(gl.GL_TEXTURE_2D, gl.GL_TEXTURE_MIN_FILTER, gl.GL_LINEAR)    # Texture.default_min_filter / default_mag_filter
        gl.glTexParameteri(gl.GL_TEXTURE_2D, gl.GL_TEXTURE_MAG_FILTER, gl.GL_LINEAR)
        gl.glTexImage2D(gl.GL_TEXTURE_2D, 0, gl.GL_RGBA, self.width, self.height, 0, gl.GL_RGBA, gl.GL_UNSIGNED_BYTE,
                        self.get_data("RGBA", self.width * 4))
        return Texture(gl.GL_TEXTURE_2D, tid.value, self.width, self.height)

    def blit(self, *a, **k):
        pass


def image_load(path, file=None, decoder=None):
    """pyglet.image.load with the PIL decoder (codecs/pil.py): flip to bottom-up rows, palettes / 1-bit to RGB(A)."""
    from PIL import Image
    with Image.open(path) as im:
        im.load()
        if im.mode in ("1", "P"):
            im = im.convert()
        if im.mode not in ("L", "LA", "RGB", "RGBA"):
            im = im.convert("RGBA")
        fmt = im.mode
        im = im.transpose(Image.FLIP_TOP_BOTTOM)
        return ImageData(im.width, im.height, fmt, im.tobytes(), im.width * len(fmt))


class Window:
    """Every pyglet Window owns a NEW context that shares objects with pyglet's shadow context: default GL state (identity
    matrices, GL_LIGHT0 at its defaults, current normal (0, 0, 1)), textures of earlier windows still valid.  One window is
    current at a time here (the reference makes one per Simulator and a second only for human viewing)."""

    def __init__(self, width=1, height=1, visible=False, **kw):
        self.width, self.height = width, height
        if _lib.glh_new_context() != 0:
            raise RuntimeError(_lib.glh_error().decode())

    def switch_to(self):
        if _lib.glh_make_current() != 0:
            raise RuntimeError(_lib.glh_error().decode())

    def clear(self):
        pass

    def dispatch_events(self):
        pass

    def close(self):
        pass

    def flip(self):
        pass


class Label:
    def __init__(self, *a, **k):
        self.text = ""

    def draw(self):
        pass


def install(width: int = 64, height: int = 64) -> dict:
    """Create the context (once) and return {module name: module} for sys.modules."""
    global _gl
    lib = _context(width, height)
    if _gl is None:
        _gl = _make_gl(lib)
        _gl._shadow_window = types.SimpleNamespace(switch_to=lambda: None)
    gl = _gl
    pyglet = types.ModuleType("pyglet")
    pyglet.__path__ = []
    pyglet.version = "1.5.0 (oracle/gl/glshim.py over " + renderer() + ")"
    pyglet.options = {"debug_gl": True, "headless": True}
    graphics = types.ModuleType("pyglet.graphics")
    graphics.vertex_list = lambda count, *attrs: VertexList(gl, count, *attrs)
    image = types.ModuleType("pyglet.image")
    image.load, image.ImageData, image.Texture = image_load, ImageData, Texture
    window = types.ModuleType("pyglet.window")
    window.Window = Window
    text = types.ModuleType("pyglet.text")
    text.Label = Label
    pyglet.gl, pyglet.graphics, pyglet.image, pyglet.window, pyglet.text = gl, graphics, image, window, text
    return {"pyglet": pyglet, "pyglet.gl": gl, "pyglet.graphics": graphics, "pyglet.image": image, "pyglet.window": window,
            "pyglet.text": text}


def gl_module():
    install()
    return _gl


def renderer() -> str:
    lib = _context(64, 64)
    gs = ctypes.CFUNCTYPE(c_char_p, c_uint)(lib.glh_get_proc(b"glGetString"))
    return f"{gs(0x1F01).decode()} / {gs(0x1F02).decode()}"
