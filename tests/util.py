"""Shared helpers for the parity tests (oracle is the checker, never the product)."""
import copy
import os

import numpy as np

from dtsim import _ffi, assets
from oracle import sim as osim

EXT = assets.mesh_extents(("duckie",))


def make_oracle(map_name, **kw):
    return osim.OracleSim(assets.get_map(map_name), EXT, **kw)


def init_state_from_oracle(o, map_id=0):
    """dtsim_init_state equal to what the oracle's reset() decided (parity mode)."""
    st = _ffi.InitState()
    st.pos[:] = [float(v) for v in o.cur_pos]
    st.angle = float(o.cur_angle)
    st.map_id = map_id
    st.dynamics_trim_on = 1 if o.dynamics_rand else 0
    st.dynamics_trim = float(0 + o.randomization_settings["trim"][0])
    st.wheel_dist = float(o.wheel_dist)
    st.cam_height = float(np.asarray(o.cam_height).reshape(-1)[0])
    st.cam_angle_deg = float(np.asarray(o.cam_angle[0]).reshape(-1)[0])
    st.cam_fov_y_deg = float(np.asarray(o.cam_fov_y).reshape(-1)[0])
    st.camera_noise[:] = [float(v) for v in o.randomization_settings["camera_noise"]]
    st.horizon_color[:] = [float(v) for v in o.horizon_color]
    st.ground_color[:] = [float(v) for v in o.ground_color]
    lp = list(o.light_pos) + [0.0] * (4 - len(o.light_pos))
    st.light_pos[:] = [float(v) for v in lp]
    st.light_ambient[:] = [float(v) for v in o.light_ambient[:3]]
    st.light_diffuse[:] = [float(v) for v in o.light_diffuse[:3]]
    return st


def random_poses(rng, mt_w, mt_h, ts, n, centers=None):
    pos = np.stack([rng.uniform(-0.2, mt_w * ts + 0.2, n), rng.uniform(-0.2, mt_h * ts + 0.2, n),
                    rng.uniform(-4, 7, n)], axis=1)
    if centers is not None and len(centers):
        k = n // 3
        c = centers[rng.integers(0, len(centers), k)]
        pos[:k, 0] = c[:, 0] + rng.uniform(-0.3, 0.3, k)
        pos[:k, 1] = c[:, 1] + rng.uniform(-0.3, 0.3, k)
    return pos


from oracle.fixtures import junction_map  # noqa: E402,F401  (every tile kind x orientation)


def oracle_mode(sim, segment=False):
    """`lighting` argument of oracle.raster.render_obs that restates what the product's pipeline for THIS simulator does (the dispatch of
    csrc/render.hip dt_launch_render): the quad-record kernels filter tile textures with byte weights -- the lit factor folded into them on
    the shared-camera path ("pixel": k_raster_v3 / k_raster_q), applied per channel afterwards on the per-env path ("pixel-dr":
    k_raster_v3dr, domain randomisation or per_env_camera) -- and the generic raster (segment view, widths that are not a multiple of 4,
    per-env cameras over tile textures that are not 256 x 256) with llvmpipe's own arithmetic ("pixel-gl")."""
    side = sim.textures[0].shape[0] if sim.textures else 256
    quad_ok = sim.camera_width % 4 == 0 and not segment
    if sim.domain_rand or sim.per_env_camera:
        v3dr = quad_ok and side == 256 and os.environ.get("DTSIM_RASTER_OLD", "0") != "1"   # (the A/B switch that keeps k_raster_q also keeps the generic per-env raster)
        return "pixel-dr" if v3dr else "pixel-gl"
    return "pixel" if quad_ok else "pixel-gl"
