"""Oracle restatement of Simulator._render_img + Distortion.distort.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Follows
src/gym_duckietown/simulator.py:1707-1951 (scene order, camera, lighting state),
:386-526 (tile / ground vertex lists), objects.py:123-148, objmesh.py:181-293,
graphics.py:172-251 (4x MSAA float FBO + resolve) and distortion.py:85-125, in float64, evaluating all four MSAA samples
of every pixel (no fast path).

PINNED against the reference's own frames on real OpenGL (round 6): tests/test_gl_golden.py holds the GL-faithful mode
(lighting="gouraud") to tests/golden/ref_gl_*.npz -- 116 frames the UNMODIFIED reference Simulator rendered on Mesa 23.2.1
llvmpipe, the renderer of its own CI (oracle/gl/, oracle/make_gl_golden.py) -- at >= 99.6 % of the pixels bit-identical,
<= 0.12 % beyond +-1/255.  What GL does where the reference's calls leave room (all measured there):
  * 4 samples at GL_SAMPLE_POSITION, rows flipped; coverage/depth per sample, shading once per primitive at the pixel
    centre (attributes extrapolated) -- GL MSAA semantics.
  * lighting: C = clamp01(m * (0.3 + A + D max(0, N.L))) per vertex; normals go through the inverse transpose of the
    model-view and are NOT renormalised (GL_NORMALIZE is off): object normals are divided by the object's scale, the ground
    quad -- which has no normals -- is lit with GL's current normal (0, 0, 1) through glScalef(50, 0.01, 50).
    The light position is fixed in eye space (whatever model-view was current when reset() called glLightfv).
  * tiles: lighting="gouraud" evaluates the lit colour at the 8x8 vertex grid of the tile (simulator.py:386-433) and
    interpolates bilinearly inside each quad; the "pixel*" modes evaluate it at the fragment -- what the HIP raster does.
  * GL_LINEAR: llvmpipe's 8-bit integer filter (_gl_linear); the "pixel" / "pixel-dr" modes restate the product's
    byte-weight filter instead (_dtsim8_shade); see MODES.
  * objects: per-vertex lighting, perspective-correct barycentric interpolation; triangles with a vertex closer than the
    near plane are dropped (no near clipping).
  * readback: round(255 * clamp01(mean of the 4 samples)).
"""
from __future__ import annotations

import math

import numpy as np

NEAR, FAR = 0.04, 100.0
GROUND_Y, GROUND_HALF = -0.008, 50.0
CAMERA_FORWARD_DIST = 0.066
# 4x MSAA sample positions, (dx, dy) in pixels from the pixel centre, +y DOWN the returned image.  GL_SAMPLE_POSITION on Mesa
# llvmpipe: (0.375, 0.125) (0.875, 0.375) (0.125, 0.625) (0.625, 0.875) from the pixel's lower-left corner in window
# coordinates (+y up; confirmed by rasterising 0.1-px squares, oracle/gl/measure_filter.py); _render_img flips the rows.
SAMPLE_OFFSETS = [(-0.125, 0.375), (0.375, 0.125), (-0.375, -0.125), (0.125, -0.375)]


class Camera:
    """Camera / light state of one env (simulator.py:1758-1803, 565-584)."""

    def __init__(self, pos, angle, *, cam_height=0.108, cam_angle_deg=19.15, cam_fov_y_deg=75.0,
                 camera_noise=(0, 0, 0), domain_rand=False, horizon_color=(0.45, 0.82, 1.0),
                 ground_color=(0.15, 0.15, 0.15), light_pos=(0.0, 3.0, 0.0, 1.0),
                 light_ambient=(0.25, 0.25, 0.25), light_diffuse=(0.35, 0.35, 0.35), width=640, height=480):
        pos = np.asarray(pos, dtype=np.float64)
        if domain_rand:
            pos = pos + np.asarray(camera_noise, dtype=np.float64)
        self.sa, self.ca = math.sin(angle), math.cos(angle)
        self.C = np.array([pos[0] + CAMERA_FORWARD_DIST * self.ca, pos[1] + cam_height,
                           pos[2] - CAMERA_FORWARD_DIST * self.sa])
        th = math.radians(float(cam_angle_deg))
        self.sth, self.cth = math.sin(th), math.cos(th)
        self.ty = math.tan(math.radians(float(cam_fov_y_deg)) / 2)
        self.tx = self.ty * (width / float(height))
        self.W, self.H = width, height
        self.horizon = np.asarray(horizon_color, dtype=np.float64)[:3] * 255.0
        self.ground = np.asarray(ground_color, dtype=np.float64)[:3] * 255.0
        self.base = 0.3 + np.asarray(light_ambient, dtype=np.float64)[:3]
        self.dif = np.asarray(light_diffuse, dtype=np.float64)[:3]
        L = np.zeros(4)
        lp = list(light_pos)
        L[:len(lp)] = lp
        if L[3] == 0:
            L[:3] = L[:3] / np.linalg.norm(L[:3])
        self.L = L

    # world -> eye (rotation part acts on offsets from the camera centre)
    def to_eye(self, P):
        rel = np.asarray(P, dtype=np.float64) - self.C
        xla = rel[..., 0] * self.sa + rel[..., 2] * self.ca
        yla = rel[..., 1]
        zla = -(rel[..., 0] * self.ca - rel[..., 2] * self.sa)
        return np.stack([xla, yla * self.cth - zla * self.sth, yla * self.sth + zla * self.cth], axis=-1)

    def normal_to_eye(self, n):
        n = np.asarray(n, dtype=np.float64)
        xla = n[..., 0] * self.sa + n[..., 2] * self.ca
        yla = n[..., 1]
        zla = -(n[..., 0] * self.ca - n[..., 2] * self.sa)
        return np.stack([xla, yla * self.cth - zla * self.sth, yla * self.sth + zla * self.cth], axis=-1)

    def ndl(self, p_eye, n_eye):
        """max(0, N.L) for eye-space points / unit normals."""
        if self.L[3] == 0:
            d = n_eye @ self.L[:3]
        else:
            lv = self.L[:3] - p_eye
            d = np.sum(n_eye * lv, axis=-1) / np.linalg.norm(lv, axis=-1)
        return np.maximum(d, 0.0)

    def lit(self, ndl):
        """clamp01(base + dif * ndl) per channel -> [...,3]."""
        return np.minimum(self.base + self.dif * np.asarray(ndl)[..., None], 1.0)


def _rays(cam, nx, ny):
    xe, ye = nx * cam.tx, ny * cam.ty
    return xe, ye, ye * cam.cth - cam.sth, ye * cam.sth + cam.cth


def _plane_hit(cam, xe, fwd, yla, h):
    with np.errstate(divide="ignore", invalid="ignore"):
        t = h / (-yla)
    rr, ff = t * xe, t * fwd
    return t, cam.C[0] + rr * cam.sa + ff * cam.ca, cam.C[2] + rr * cam.ca - ff * cam.sa


def _tile_uv(angle, fx, fz):
    u = np.where(angle == 0, 1 - fx, np.where(angle == 1, fz, np.where(angle == 2, fx, 1 - fz)))
    v = np.where(angle == 0, fz, np.where(angle == 1, fx, np.where(angle == 2, 1 - fz, 1 - fx)))
    return u, v


def _gl_linear(tex, u, v):
    """GL_LINEAR / GL_REPEAT fetch as the reference's renderer computes it; tex [h,w,4] uint8 with row 0 = v=0; returns the
    sampler's 8-bit result as float [...,3].

    GL leaves the precision of the filter to the implementation.  This is the arithmetic of Mesa's llvmpipe for 8-bit unorm
    textures, MEASURED on Mesa 23.2.1 (oracle/gl/measure_filter.py -> profiles/r06_gl_filter_precision.txt: bit-identical on
    786 432 random cases; it is gallivm's AoS sampling path, lp_bld_sample_aos.c): the texel coordinate times 256, rounded,
    minus 128 (half a texel) -- the integer part addresses the texels, the low 8 bits are the weight;
    lerp(w, p, q) = p + ((w (q - p) + 128) >> 8), first along s for both rows, THEN along t on the 8-bit results; the sampler
    hands an 8-bit colour to the (float) texture environment."""
    h, w = tex.shape[:2]
    f32 = np.float32
    xs = np.floor((np.asarray(u, dtype=f32) * f32(w)) * f32(256) + f32(0.5)).astype(np.int64) - 128
    ys = np.floor((np.asarray(v, dtype=f32) * f32(h)) * f32(256) + f32(0.5)).astype(np.int64) - 128
    ax, ay = (xs & 255)[..., None], (ys & 255)[..., None]
    x0, y0 = (xs >> 8) % w, (ys >> 8) % h
    x1, y1 = (x0 + 1) % w, (y0 + 1) % h
    t = tex[..., :3].astype(np.int64)
    top = t[y0, x0] + ((ax * (t[y0, x1] - t[y0, x0]) + 128) >> 8)
    bot = t[y1, x0] + ((ax * (t[y1, x1] - t[y1, x0]) + 128) >> 8)
    return (top + ((ay * (bot - top) + 128) >> 8)).astype(np.float64)


def _exact_linear(tex, u, v):
    """Float64 bilinear (no implementation's filter: the oracle before the GL goldens existed)."""
    h, w = tex.shape[:2]
    x, y = u * w - 0.5, v * h - 0.5
    x0f, y0f = np.floor(x), np.floor(y)
    ax, ay = (x - x0f)[..., None], (y - y0f)[..., None]
    x0, y0 = x0f.astype(np.int64) % w, y0f.astype(np.int64) % h
    x1, y1 = (x0 + 1) % w, (y0 + 1) % h
    t = tex[..., :3].astype(np.float64)
    top = t[y0, x0] + ax * (t[y0, x1] - t[y0, x0])
    bot = t[y1, x0] + ax * (t[y1, x1] - t[y1, x0])
    return top + ay * (bot - top)


def _dtsim8_shade(tex, u, v, I, folded):
    """Tile colour (0..255 float, unrounded) of the product's quad-record pipeline (csrc/render.hip quad_weights8 / quad_filter,
    DESIGN.md section 5), at the precision GL's own filter has:
      * the texel coordinate is rounded to 256ths of a texel first (round-to-nearest-even: the product's one float add X + 32768),
        as llvmpipe rounds its own (_gl_linear); the integer part addresses the texels -- a fraction that rounds up to 1 carries;
      * ONE multiply-accumulate per texel and channel: the four bilinear weights, times 256, rounded to bytes (float32 arithmetic
        in the product's order of operations, round-to-nearest-even as v_cvt_pk_u8_f32 does); sum(texel * weight) / 256.
    `folded` (shared camera, k_raster_v3 / k_raster_q: the lit factor I [...,3] is the same for the three channels): it is folded
    into the weights; otherwise (k_raster_v3dr: per-env, per-channel light) the weights are unlit and the sum is multiplied by I.
    Not llvmpipe's arithmetic (which rounds to 8 bits between its two lerps): it differs from _gl_linear(...) * I by +-1/255 on
    about a quarter of the textured pixels, never systematically (tests/test_gl_golden.py measures it on the GL frames)."""
    h, w = tex.shape[:2]
    f32, f64 = np.float32, np.float64
    xq = np.rint((np.asarray(u, dtype=f64) * w - 0.5) * 256.0)
    yq = np.rint((np.asarray(v, dtype=f64) * h - 0.5) * 256.0)
    x0f, y0f = np.floor(xq / 256.0), np.floor(yq / 256.0)
    a8, b8 = (xq - 256.0 * x0f).astype(f32), (yq - 256.0 * y0f).astype(f32)      # the fractions x 256: 0..255
    x0, y0 = x0f.astype(np.int64) % w, y0f.astype(np.int64) % h
    x1, y1 = (x0 + 1) % w, (y0 + 1) % h
    I = np.asarray(I, dtype=f64)
    l = (I[..., 0].astype(f32) * f32(1.0 / 256.0)) if folded else np.full(a8.shape, 1.0 / 256.0, f32)
    fma = lambda a, b, c: (a.astype(f64) * f64(b) + c.astype(f64)).astype(f32)      # float32 fused multiply-add (the product is exact in float64)
    uu = a8 * l
    vv = fma(l, 256.0, -uu)
    w11, w01 = uu * b8, vv * b8
    w10, w00 = fma(uu, 256.0, -w11), fma(vv, 256.0, -w01)
    W = [np.clip(np.rint(wgt), 0, 255).astype(np.int64)[..., None] for wgt in (w00, w10, w01, w11)]
    t = tex[..., :3].astype(np.int64)
    S = t[y0, x0] * W[0] + t[y0, x1] * W[1] + t[y1, x0] * W[2] + t[y1, x1] * W[3]
    out = S.astype(f64) / 256.0
    return out if folded else out * I


def _tile_shade(tex, u, v, I, tile_filter):
    if tile_filter in ("dtsim8", "dtsim8-dr"):
        return _dtsim8_shade(tex, u, v, I, tile_filter == "dtsim8" and bool(np.all(I[..., 0] == I[..., 1]) and np.all(I[..., 0] == I[..., 2])))
    return (_gl_linear(tex, u, v) if tile_filter == "llvmpipe" else _exact_linear(tex, u, v)) * I


# `lighting` argument of render_obs / render_rectilinear -> (where the tile light is evaluated, the tile texture filter)
#   "gouraud"   GL itself: the light at the tile's 8 x 8 vertices, llvmpipe's filter -- what the GL goldens pin
#   "pixel"     the product's quad-record pipeline: per-fragment light, the byte-weight filter -- lit factor folded into the weights when it
#               is one number for the three channels (k_raster_v3 / k_raster_q), applied per channel otherwise (k_raster_v3dr)
#   "pixel-dr"  the same with the per-channel form forced (k_raster_v3dr through per_env_camera, whatever the light's colour)
#   "pixel-gl"  the product's generic raster and exact paths (k_raster / k_resolve): per-fragment light, llvmpipe's filter
MODES = {"gouraud": ("gouraud", "llvmpipe"), "pixel": ("pixel", "dtsim8"), "pixel-dr": ("pixel", "dtsim8-dr"), "pixel-gl": ("pixel", "llvmpipe"), "pixel-exact": ("pixel", "exact")}


class Scene:
    """Static inputs of the raster: map grid, textures per tile kind, object instances."""

    def __init__(self, omap, textures, meshes):
        self.m = omap
        self.textures = textures          # kind -> [h,w,4] uint8 (GL row order)
        self.meshes = meshes              # kind -> MeshData-like (verts, normals, colors)
        W, H = omap.grid_width, omap.grid_height
        self.present = np.zeros((H, W), bool)
        self.angle = np.zeros((H, W), np.int64)
        self.kinds = [[None] * W for _ in range(H)]
        for t in omap.grid:
            if t is None:
                continue
            i, j = t["coords"]
            self.present[j, i] = True
            self.angle[j, i] = t["angle"]
            self.kinds[j][i] = t["kind"]


def _tile_light(cam, scene, lighting, ti, tj, t_c, xe_c, ye_c, wx, wz):
    """Lit white vertex colour of tile pixels -> [...,3]."""
    n_eye = np.array([0.0, cam.cth, cam.sth])
    if lighting == "pixel" or cam.L[3] == 0:
        p_eye = np.stack([t_c * xe_c, t_c * ye_c, -t_c], axis=-1)
        return cam.lit(cam.ndl(p_eye, n_eye))
    # gouraud: 8x8 vertices per tile in the tile-local frame (simulator.py:388-401);
    # the vertex grid is symmetric under the 90-degree tile rotations, so it is
    # axis-aligned in world space with spacing ts/7.
    ts = scene.m.tile_size
    gx = (wx - ti * ts) / ts * 7.0
    gz = (wz - tj * ts) / ts * 7.0
    qx, qz = np.clip(np.floor(gx), 0, 6), np.clip(np.floor(gz), 0, 6)
    ax, az = (gx - qx)[..., None], (gz - qz)[..., None]

    def vert(dx, dz):
        P = np.stack([(ti + (qx + dx) / 7.0) * ts, np.zeros_like(wx), (tj + (qz + dz) / 7.0) * ts], axis=-1)
        return cam.lit(cam.ndl(cam.to_eye(P), n_eye))

    top = vert(0, 0) + ax * (vert(1, 0) - vert(0, 0))
    bot = vert(0, 1) + ax * (vert(1, 1) - vert(0, 1))
    return top + az * (bot - top)


def _shade_planes(cam, scene, lighting, cls, ti, tj, t_s, wx_s, wz_s, rc):
    """Colour (0..255) of the plane primitive seen by each sample, evaluated at the pixel
    centre.  cls: 0 sky 1 ground 2 tile."""
    xe_c, ye_c, yla_c, fwd_c = rc
    out = np.empty(cls.shape + (3,))
    out[:] = cam.horizon
    centre_down = yla_c < 0
    # ground
    g = cls == 1
    if g.any():
        t_c, wx_c, wz_c = _plane_hit(cam, xe_c, fwd_c, yla_c, cam.C[1] - GROUND_Y)
        wx = np.where(centre_down, wx_c, wx_s)[g]
        wz = np.where(centre_down, wz_c, wz_s)[g]
        corners = np.array([[-GROUND_HALF, GROUND_Y, -GROUND_HALF], [GROUND_HALF, GROUND_Y, -GROUND_HALF],
                            [-GROUND_HALF, GROUND_Y, GROUND_HALF], [GROUND_HALF, GROUND_Y, GROUND_HALF]])
        # the ground vertex list carries no normals (simulator.py:526): GL lights it with the CURRENT normal -- (0, 0, 1), the
        # initial value, nothing in the reference's default path calls glNormal -- through the inverse transpose of
        # glScalef(50, 0.01, 50) (:1810), un-normalised: length 1/50, so the diffuse term all but vanishes.
        nd = np.broadcast_to(cam.ndl(cam.to_eye(corners), cam.normal_to_eye(np.array([0.0, 0.0, 1.0 / GROUND_HALF]))), (4,))
        a = np.clip((wx + GROUND_HALF) / (2 * GROUND_HALF), 0, 1)
        b = np.clip((wz + GROUND_HALF) / (2 * GROUND_HALF), 0, 1)
        n0 = nd[0] + a * (nd[1] - nd[0])
        n1 = nd[2] + a * (nd[3] - nd[2])
        out[g] = cam.ground * cam.lit(n0 + b * (n1 - n0))
    # tiles
    tl = cls == 2
    if tl.any():
        t_c, wx_c, wz_c = _plane_hit(cam, xe_c, fwd_c, yla_c, cam.C[1])
        t = np.where(centre_down, t_c, t_s)[tl]
        wx = np.where(centre_down, wx_c, wx_s)[tl]
        wz = np.where(centre_down, wz_c, wz_s)[tl]
        i, j = ti[tl], tj[tl]
        ts = scene.m.tile_size
        light_mode, tile_filter = MODES[lighting]
        I = np.broadcast_to(_tile_light(cam, scene, light_mode, i, j, t, xe_c[tl], ye_c[tl], wx, wz), (i.size, 3))
        fx, fz = wx / ts - i, wz / ts - j
        u, v = _tile_uv(scene.angle[j, i], fx, fz)
        col = 255.0 * I
        flat_kind = np.array([hash(scene.kinds[jj][ii]) for ii, jj in zip(i.tolist(), j.tolist())]) if i.size else np.zeros(0)
        for kind in {scene.kinds[jj][ii] for ii, jj in zip(i.tolist(), j.tolist())}:
            sel = flat_kind == hash(kind)
            tex = scene.textures.get(kind)
            if tex is not None:
                col[sel] = _tile_shade(tex, u[sel], v[sel], I[sel], tile_filter)
        out[tl] = col
    return out


def _object_instances(scene, obj_states):
    """World-space lit-ready triangles of all visible objects: list of
    (verts [T,3,3], normals [T,3,3], colors [T,3,3], uvs [T,3,2], per-triangle texture image or None
    -- the chunk's map_Kd texture, objmesh.py:268-275)."""
    out = []
    for k, o in enumerate(scene.m.objects):
        st = obj_states[k] if obj_states is not None else None
        if st is not None and not st.get("visible", True):
            continue
        if st is None and not getattr(o, "visible", True):
            continue
        mesh = scene.meshes.get(o.kind) or scene.meshes.get("*")
        if mesh is None:
            continue
        pos = np.asarray(st["pos"] if st is not None else o.pos, dtype=np.float64)
        yrot = math.radians(st["y_rot"] if st is not None else o.y_rot)
        c, s = math.cos(yrot), math.sin(yrot)
        # glRotatef(y_rot, 0,1,0): x' = x c + z s ; z' = -x s + z c   (objects.py:140-146)
        V = mesh.verts.astype(np.float64) * o.scale
        Vw = np.stack([V[..., 0] * c + V[..., 2] * s, V[..., 1], -V[..., 0] * s + V[..., 2] * c], axis=-1) + pos
        Nn = mesh.normals.astype(np.float64)
        # GL transforms normals by the inverse transpose of the model-view and does NOT renormalise them (GL_NORMALIZE /
        # GL_RESCALE_NORMAL are never enabled in the reference): glScalef(scale) divides them by `scale`, and whatever length
        # the OBJ file gives them stays.  Measured against Mesa llvmpipe (tests/test_gl_golden.py).
        Nw = np.stack([Nn[..., 0] * c + Nn[..., 2] * s, Nn[..., 1], -Nn[..., 0] * s + Nn[..., 2] * c], axis=-1) / o.scale
        T = Vw.shape[0]
        uvs = np.asarray(getattr(mesh, "uvs", np.zeros((T, 3, 2))), dtype=np.float64)
        texs = getattr(mesh, "textures", None) or []
        tri_tex = np.asarray(getattr(mesh, "tri_tex", np.full(T, -1)))
        tri_img = [texs[t] if (t >= 0 and t < len(texs)) else None for t in tri_tex]
        cards = getattr(scene, "light_cards", None)
        if getattr(o, "light_freq", 0) > 0 and cards is not None:
            # TrafficLightObj: mesh.textures[0] (the first material chunk) is the card of the current pattern
            pat = int(st["light_pattern"]) if (st is not None and "light_pattern" in st) else int(o.light_pattern)
            n0 = int(getattr(mesh, "chunk_sizes", [0])[0])
            tri_img = [cards[pat] if t < n0 else im for t, im in enumerate(tri_img)]
        out.append((Vw, Nw, mesh.colors.astype(np.float64), uvs, tri_img))
    return out


def render_rectilinear(cam, scene, lighting="gouraud", obj_states=None, return_depth=False):
    """[H,W,3] float (0..255) of the un-distorted frame, row 0 = top.  return_depth: also the four per-sample depth buffers
    (eye depth of the nearest opaque fragment, inf = clear colour) -- what overlay_leds tests the LED spheres against."""
    W, H = cam.W, cam.H
    cols, rows = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    nxc = 2 * (cols + 0.5) / W - 1
    nyc = 1 - 2 * (rows + 0.5) / H
    rc = _rays(cam, nxc, nyc)
    gw, gh, ts = scene.m.grid_width, scene.m.grid_height, scene.m.tile_size

    # objects: eye-space vertices, per-vertex lit colours, screen coordinates
    tris = []
    for Vw, Nw, Cc, UV, TI in _object_instances(scene, obj_states):
        Pe = cam.to_eye(Vw)
        Ne = cam.normal_to_eye(Nw)
        lit = np.minimum(Cc * (cam.base + cam.dif * cam.ndl(Pe, Ne)[..., None]), 1.0) * 255.0
        w = -Pe[..., 2]
        ok = (w > NEAR).all(axis=1)
        with np.errstate(divide="ignore", invalid="ignore"):
            sx = (Pe[..., 0] / w / cam.tx + 1) * 0.5 * W       # pixel coords, x right
            sy = (1 - Pe[..., 1] / w / cam.ty) * 0.5 * H       # y down
        for k in np.flatnonzero(ok):
            tris.append((sx[k], sy[k], w[k], lit[k], UV[k], TI[k]))

    acc = np.zeros((H, W, 3))
    depths = []
    for (ox, oy) in SAMPLE_OFFSETS:
        nx, ny = nxc + 2 * ox / W, nyc - 2 * oy / H
        xe, ye, yla, fwd = _rays(cam, nx, ny)
        down = yla < 0
        cls = np.zeros((H, W), np.int64)
        ti = np.zeros((H, W), np.int64)
        tj = np.zeros((H, W), np.int64)
        depth = np.full((H, W), np.inf)
        t_s = np.zeros((H, W)); wx_s = np.zeros((H, W)); wz_s = np.zeros((H, W))
        # ground
        tg, wxg, wzg = _plane_hit(cam, xe, fwd, yla, cam.C[1] - GROUND_Y)
        gok = down & (tg >= NEAR) & (tg <= FAR) & (np.abs(wxg) <= GROUND_HALF) & (np.abs(wzg) <= GROUND_HALF)
        cls[gok] = 1
        depth[gok] = tg[gok]; t_s[gok] = tg[gok]; wx_s[gok] = wxg[gok]; wz_s[gok] = wzg[gok]
        # tiles (depth func LESS: y=0 is in front of y=-0.008 wherever both are hit)
        tt, wxt, wzt = _plane_hit(cam, xe, fwd, yla, cam.C[1])
        with np.errstate(invalid="ignore"):
            fi, fj = np.floor(wxt / ts), np.floor(wzt / ts)
        tok = down & (tt >= NEAR) & (tt <= FAR) & (fi >= 0) & (fj >= 0) & (fi < gw) & (fj < gh)
        ii = np.where(tok, fi, 0).astype(np.int64)
        jj = np.where(tok, fj, 0).astype(np.int64)
        tok &= scene.present[jj, ii]
        cls[tok] = 2
        ti[tok] = ii[tok]; tj[tok] = jj[tok]
        depth[tok] = tt[tok]; t_s[tok] = tt[tok]; wx_s[tok] = wxt[tok]; wz_s[tok] = wzt[tok]
        col = _shade_planes(cam, scene, lighting, cls, ti, tj, t_s, wx_s, wz_s, rc)
        # objects: z-buffered triangles, coverage at the sample, colour at the pixel centre
        for (sx, sy, w, lit, uv, timg) in tris:
            x0 = max(int(math.floor(sx.min() - 1)), 0); x1 = min(int(math.ceil(sx.max() + 1)), W - 1)
            y0 = max(int(math.floor(sy.min() - 1)), 0); y1 = min(int(math.ceil(sy.max() + 1)), H - 1)
            if x0 > x1 or y0 > y1:
                continue
            area = (sx[1] - sx[0]) * (sy[2] - sy[0]) - (sx[2] - sx[0]) * (sy[1] - sy[0])
            if area == 0:
                continue
            px = cols[y0:y1 + 1, x0:x1 + 1] + 0.5
            py = rows[y0:y1 + 1, x0:x1 + 1] + 0.5

            def bary(qx, qy):
                b0 = ((sx[1] - qx) * (sy[2] - qy) - (sx[2] - qx) * (sy[1] - qy)) / area
                b1 = ((sx[2] - qx) * (sy[0] - qy) - (sx[0] - qx) * (sy[2] - qy)) / area
                return b0, b1, 1 - b0 - b1

            b0, b1, b2 = bary(px + ox, py + oy)
            inside = (b0 >= 0) & (b1 >= 0) & (b2 >= 0)
            if not inside.any():
                continue
            iw = b0 / w[0] + b1 / w[1] + b2 / w[2]
            with np.errstate(divide="ignore", invalid="ignore"):
                d = 1.0 / iw
            sub = depth[y0:y1 + 1, x0:x1 + 1]
            win = inside & (d < sub) & (d >= NEAR) & (d <= FAR)
            if not win.any():
                continue
            c0, c1, c2 = bary(px, py)                      # attributes at the pixel centre
            iwc = c0 / w[0] + c1 / w[1] + c2 / w[2]
            with np.errstate(divide="ignore", invalid="ignore"):
                colc = (c0[..., None] * lit[0] / w[0] + c1[..., None] * lit[1] / w[1] + c2[..., None] * lit[2] / w[2]) / iwc[..., None]
            colc = np.clip(np.nan_to_num(colc, nan=0.0, posinf=255.0, neginf=0.0), 0.0, 255.0)
            if timg is not None:               # GL_MODULATE with the material's texture (GL_LINEAR / GL_REPEAT)
                with np.errstate(divide="ignore", invalid="ignore"):
                    uc = (c0 * uv[0, 0] / w[0] + c1 * uv[1, 0] / w[1] + c2 * uv[2, 0] / w[2]) / iwc
                    vc = (c0 * uv[0, 1] / w[0] + c1 * uv[1, 1] / w[1] + c2 * uv[2, 1] / w[2]) / iwc
                uc, vc = np.nan_to_num(uc, nan=0.0, posinf=0.0, neginf=0.0), np.nan_to_num(vc, nan=0.0, posinf=0.0, neginf=0.0)
                uc, vc = np.where(win, uc, 0.0), np.where(win, vc, 0.0)
                colc = colc * (_gl_linear(timg, uc, vc) / 255.0)
            sub[win] = d[win]
            csub = col[y0:y1 + 1, x0:x1 + 1]
            csub[win] = colc[win]
        acc += col
        depths.append(depth)
    return (acc / 4.0, depths) if return_depth else acc / 4.0


def to_u8(img):
    return np.floor(np.clip(img, 0.0, 255.0) + 0.5).astype(np.uint8)


def distort(img_u8, rmapx, rmapy):
    """cv2.remap(INTER_NEAREST, BORDER_CONSTANT 0) (distortion.py:118-124)."""
    H, W = img_u8.shape[:2]
    sx = np.rint(rmapx.astype(np.float64)).astype(np.int64)
    sy = np.rint(rmapy.astype(np.float64)).astype(np.int64)
    ok = (sx >= 0) & (sx < W) & (sy >= 0) & (sy < H)
    out = np.zeros_like(img_u8)
    out[ok] = img_u8[sy[ok], sx[ok]]
    return out


def overlay_lines(img_u8, cam, lines):
    """The GL_LINE overlays of the reference on the RESOLVED rectilinear image (uint8 [H,W,3]): draw_curve (simulator.py:1886-1904,
    graphics.py:336-349) and draw_bbox (simulator.py:1907-1918, objects.py:131-146).  `lines`: [n, 9] world-space segments
    (ax, ay, az, bx, by, bz) + glColor (r, g, b in 0..1).  One documented interpretation (the GL state at those draw calls is whatever
    the previous call left behind -- PARITY UNPINNED): a segment is clipped at the near plane, projected, and covers the MSAA samples
    within half a pixel of it (the 1-px line rectangle of multisample rasterisation); its colour is the glColor lit as a surface with
    normal +y at the segment's middle (GL_COLOR_MATERIAL); the first line in the list wins a sample; a pixel with n covered samples
    becomes ((4 - n) pixel + sum of the covering colours) / 4, rounded -- no depth test against meshes.  What the HIP post-pass
    k_overlay_lines does, in float64."""
    H, W = img_u8.shape[:2]
    out = img_u8.astype(np.float64)
    cov = np.zeros((H, W, 4), bool)
    acc = np.zeros((H, W, 3))
    cols, rows = np.meshgrid(np.arange(W, dtype=np.float64) + 0.5, np.arange(H, dtype=np.float64) + 0.5)
    n_eye = np.array([0.0, cam.cth, cam.sth])
    for L in np.asarray(lines, dtype=np.float64).reshape(-1, 9):
        pe = cam.to_eye(np.stack([L[0:3], L[3:6]]))
        w = -pe[:, 2]
        wn = NEAR * 1.0001
        if w[0] < wn and w[1] < wn:
            continue
        if w[0] < wn or w[1] < wn:
            vb = 0 if w[0] < wn else 1
            vf = 1 - vb
            t = (wn - w[vb]) / (w[vf] - w[vb])
            pe[vb] = pe[vb] + t * (pe[vf] - pe[vb])
            w = -pe[:, 2]
        ndl = 0.5 * (cam.ndl(pe[0], n_eye) + cam.ndl(pe[1], n_eye))
        col = 255.0 * np.minimum(L[6:9] * (cam.base + cam.dif * ndl), 1.0)
        ax, ay = (pe[0, 0] / w[0] / cam.tx + 1) * 0.5 * W, (1 - pe[0, 1] / w[0] / cam.ty) * 0.5 * H
        bx, by = (pe[1, 0] / w[1] / cam.tx + 1) * 0.5 * W, (1 - pe[1, 1] / w[1] / cam.ty) * 0.5 * H
        dx, dy = bx - ax, by - ay
        l2 = dx * dx + dy * dy
        if l2 <= 1e-12:
            continue
        x0, x1 = max(int(math.floor(min(ax, bx) - 1.5)), 0), min(int(math.ceil(max(ax, bx) + 1.5)), W - 1)
        y0, y1 = max(int(math.floor(min(ay, by) - 1.5)), 0), min(int(math.ceil(max(ay, by) + 1.5)), H - 1)
        if x0 > x1 or y0 > y1:
            continue
        px, py = cols[y0:y1 + 1, x0:x1 + 1], rows[y0:y1 + 1, x0:x1 + 1]
        for q, (ox, oy) in enumerate(SAMPLE_OFFSETS):
            qx, qy = px + ox - ax, py + oy - ay
            u = (qx * dx + qy * dy) / l2
            cr = qx * dy - qy * dx
            inside = (u >= 0) & (u <= 1) & (cr * cr / l2 <= 0.25) & ~cov[y0:y1 + 1, x0:x1 + 1, q]
            cov[y0:y1 + 1, x0:x1 + 1, q] |= inside
            acc[y0:y1 + 1, x0:x1 + 1] += inside[..., None] * col
    n = cov.sum(axis=-1)
    hit = n > 0
    out[hit] = 0.25 * ((4 - n[hit])[:, None] * out[hit] + acc[hit])
    return to_u8(out)


LED_POSITIONS = [(0.1, 0.05, -0.05), (0.1, 0.05, 0.05), (0.1, 0.05, 0.0), (-0.1, 0.05, -0.05), (-0.1, 0.05, 0.05)]   # glTranslatef(px, pz, py) of
# front_left, front_right, center, back_left, back_right in the dict's order (objects.py:74-80, 90-95)
LED_FOLLOWER = [(0.5, 0.5, 0.5), (0.5, 0.5, 0.5), (0.0, 0.0, 0.2), (0.5, 0.0, 0.0), (0.5, 0.0, 0.0)]                    # DuckiebotObj.leds_color (objects.py:218-224)
LED_STATIC = [(0.0, 0.0, 1.0)] * 5                                                                                     # any other duckiebot-kind object (objects.py:86-92)


def led_spheres(scene, obj_states=None):
    """The gluSpheres WorldObj.render_mesh draws for every visible object of kind "duckiebot" when enable_leds is on (objects.py:68-121), in
    draw order, as world-space rows (cx, cy, cz, radius, r, g, b, alpha): per LED a 1 cm sphere at alpha 1 and a halo of radius
    mean(colour) * 4 cm at alpha 0.2, inside the object's translate / scale / rotate."""
    rows = []
    for k, o in enumerate(scene.m.objects):
        if o.kind != "duckiebot":
            continue
        st = obj_states[k] if obj_states is not None else None
        if (st is not None and not st.get("visible", True)) or (st is None and not getattr(o, "visible", True)):
            continue
        pos = np.asarray(st["pos"] if st is not None else o.pos, dtype=np.float64)
        yrot = math.radians(st["y_rot"] if st is not None else o.y_rot)
        c, s_ = math.cos(yrot), math.sin(yrot)
        cols = LED_STATIC if o.static else LED_FOLLOWER
        for (lx, ly, lz), col in zip(LED_POSITIONS, cols):
            col = np.clip(np.asarray(col, dtype=np.float64), 0.0, 1.0)
            x, y, z = lx * o.scale, ly * o.scale, lz * o.scale
            cw = np.array([x * c + z * s_, y, -x * s_ + z * c]) + pos
            rows.append([*cw, 0.01 * o.scale, *col, 1.0])
            rows.append([*cw, float(np.mean(col)) * 0.04 * o.scale, *col, 0.2])
    return np.asarray(rows, dtype=np.float64).reshape(-1, 8)


def overlay_leds(img_u8, cam, depths, spheres):
    """The additively blended LED spheres (objects.py:68-121: glBlendFunc(GL_SRC_ALPHA, GL_ONE), depth test and depth writes on) as a pass
    over the RESOLVED rectilinear image and the per-sample depth buffers of the opaque scene.  One documented interpretation (PARITY
    UNPINNED: the real result depends on the order in which gluSphere emits its 10 x 10 strips): a sphere is the analytic sphere, only its
    FRONT surface counts (a back face drawn before its front face would add a second layer), it is lit per sample like a mesh vertex
    (GL_COLOR_MATERIAL: colour x (ambient + diffuse N.L), clamped) and adds alpha x colour to every sample where its front surface is
    nearer than whatever the sample's depth holds -- the opaque scene, then the spheres before it in the list, which write depth -- within
    [near, far].  Spheres are applied after ALL opaque objects (GL interleaves them with the objects: an object drawn after a duckiebot and
    behind one of its LEDs is hidden there).  A pixel becomes pixel + sum / 4, rounded.  What k_overlay_leds does, in float64."""
    H, W = img_u8.shape[:2]
    out = img_u8.astype(np.float64)
    cols, rows = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    nxc = 2 * (cols + 0.5) / W - 1
    nyc = 1 - 2 * (rows + 0.5) / H
    add = np.zeros((H, W, 3))
    dep = [d.copy() for d in depths]
    for sp in np.asarray(spheres, dtype=np.float64).reshape(-1, 8):
        r = sp[3]
        if not r > 0:
            continue
        pe = cam.to_eye(sp[0:3])
        w = -pe[2]
        if w + r <= NEAR:
            continue
        # conservative screen box of the sphere
        wn = max(w - r, NEAR)
        cx, cy = (pe[0] / w / cam.tx + 1) * 0.5 * W, (1 - pe[1] / w / cam.ty) * 0.5 * H
        rad = r / wn / min(cam.tx / W, cam.ty / H) * 0.5 * 1.5 + 2
        x0, x1 = max(int(math.floor(cx - rad)), 0), min(int(math.ceil(cx + rad)), W - 1)
        y0, y1 = max(int(math.floor(cy - rad)), 0), min(int(math.ceil(cy + rad)), H - 1)
        if x0 > x1 or y0 > y1:
            continue
        for q, (ox, oy) in enumerate(SAMPLE_OFFSETS):
            nx = nxc[y0:y1 + 1, x0:x1 + 1] + 2 * ox / W
            ny = nyc[y0:y1 + 1, x0:x1 + 1] - 2 * oy / H
            dx, dy = nx * cam.tx, ny * cam.ty                 # eye-space ray (dx, dy, -1) t: eye depth = t
            a = dx * dx + dy * dy + 1.0
            b = dx * pe[0] + dy * pe[1] - pe[2]               # d . P
            c = pe @ pe - r * r
            disc = b * b - a * c
            hit = disc > 0
            t = (b - np.sqrt(np.where(hit, disc, 0.0))) / a
            sub = dep[q][y0:y1 + 1, x0:x1 + 1]
            win = hit & (t >= NEAR) & (t <= FAR) & (t < sub)
            if not win.any():
                continue
            ph = np.stack([t * dx, t * dy, -t], axis=-1)
            n = (ph - pe) / r
            lit = np.minimum(sp[4:7] * (cam.base + cam.dif * cam.ndl(ph, n)[..., None]), 1.0) * 255.0
            add[y0:y1 + 1, x0:x1 + 1] += np.where(win[..., None], sp[7] * lit, 0.0)
            sub[win] = t[win]
    return to_u8(out + 0.25 * add)


def render_obs(cam, scene, lighting="gouraud", rmap=None, obj_states=None, lines=None, leds=None):
    """Simulator.render_obs (simulator.py:1953-1972): uint8 [H,W,3]; `lines`: draw_curve / draw_bbox overlays (overlay_lines); `leds`: the LED
    spheres of enable_leds (led_spheres -> overlay_leds)."""
    if leds is not None and len(leds):
        f, depths = render_rectilinear(cam, scene, lighting, obj_states, return_depth=True)
        img = overlay_leds(to_u8(f), cam, depths, leds)
    else:
        img = to_u8(render_rectilinear(cam, scene, lighting, obj_states))
    if lines is not None and len(lines):
        img = overlay_lines(img, cam, lines)
    if rmap is not None:
        img = distort(img, rmap[0], rmap[1])
    return img


def segment_view(cam, scene, seg_textures, mesh_colors):
    """(camera, scene) of `_render_img(..., segment=True)`: GL_LIGHTING / GL_LIGHT0 / GL_COLOR_MATERIAL disabled
    (simulator.py:1730-1733: the fragment is texture x vertex colour), colour buffer cleared to and ground quad drawn
    in (255, 0, 255) clamped = magenta (:1753, :1808), tile textures bound through Texture.bind(segment=True)
    (graphics.py:52-57; `seg_textures`: kind -> segmented image, prepared like load_texture(segment=True)), objects
    drawn through get_mesh(name, segment=True): every chunk textured with the flat gen_segmentation_color(name)
    (objmesh.py:255-292; `mesh_colors`: mesh key -> RGB 0..254), no traffic-light card switching."""
    import copy
    c = copy.copy(cam)
    c.base, c.dif = np.ones(3), np.zeros(3)
    c.horizon = np.array([255.0, 0.0, 255.0])
    c.ground = np.array([255.0, 0.0, 255.0])
    meshes = {}
    for key, m in scene.meshes.items():
        mm = copy.copy(m)
        flat = np.zeros((2, 2, 4), np.uint8)
        flat[..., :3] = np.asarray(mesh_colors[key], np.uint8)
        flat[..., 3] = 255
        mm.textures = [flat]
        mm.tri_tex = np.zeros(m.verts.shape[0], np.int32)
        meshes[key] = mm
    sc = Scene(scene.m, dict(seg_textures), meshes)
    sc.light_cards = None
    return c, sc
