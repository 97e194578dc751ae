"""ctypes binding of libdtsim.so (include/dtsim.h).

The HIP library is the product: there is no Python/CPU fallback.  `load()` raises
`DtsimLibraryError` when the shared object is missing, and `dtsim_create` fails with
DTSIM_E_NOGPU when no HIP device is visible.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libdtsim.so")

# ---- constants (mirror include/dtsim.h) -------------------------------------
ABI_VERSION = 11
OK, E_INVALID, E_HIP, E_NOGPU, E_STATE, E_LIMIT = 0, -1, -2, -3, -4, -5
MAX_MAPS, MAX_TILES, MAX_CURVES, MAX_STATIC, MAX_DYNAMIC, MAX_OBJECTS = 32, 1024, 1024, 56, 8, 64
MAX_DELAY, MAX_TEXTURES, MAX_MESHES = 16, 96, 64
F_RENDER, F_DISTORTION, F_DOMAIN_RAND, F_AUTO_RESET, F_ACTIONS_F64, F_PROFILE, F_LIGHT_CAPTURE = 1, 2, 4, 8, 16, 32, 64
ACTION_WHEELS, ACTION_VEL_STEER = 0, 1
DONE_IN_PROGRESS, DONE_INVALID_POSE, DONE_MAX_STEPS = 0, 1, 2
DONE_CODES = ["in-progress", "invalid-pose", "max-steps-reached"]  # simulator.py:1691,1699,1704
TILE_KINDS = {
    "empty": 0, "straight": 1, "curve_left": 2, "curve_right": 3, "3way_left": 4, "3way_right": 5,
    "4way": 6, "asphalt": 7, "grass": 8, "floor": 9,
}
TILE_OTHER = 10
(FIELD_POS, FIELD_ANGLE, FIELD_REWARD, FIELD_DONE, FIELD_DONE_CODE, FIELD_STEP_COUNT, FIELD_TILE,
 FIELD_LANE, FIELD_IN_LANE, FIELD_PROX, FIELD_SPEED, FIELD_TIMESTAMP, FIELD_WHEELS, FIELD_MAP_ID,
 FIELD_OBJ_CENTER, FIELD_OBJ_ACTIVE, FIELD_OBJ_YROT, FIELD_OBJ_PARAMS, FIELD_OBJ_VISIBLE,
 FIELD_EPISODE, FIELD_STATE_BLOB, FIELD_OBJ_LIGHT, FIELD_OBJ_Y, FIELD_OBJ_EXTRA, FIELD_CAMERA, FIELD_COLORS,
 FIELD_WHEEL_DIST, FIELD_RENDER_POS) = range(28)
KERNEL_STEP, KERNEL_RENDER, KERNEL_RESET, KERNEL_QUERY, KERNEL_OBSERVE = range(5)
OBS_HWC, OBS_CHW, OBS_F32 = 0, 1, 2
RENDER_SEGMENT, RENDER_GL_FILTER = 1, 2
STEP_ONE_UPDATE, STEP_POSE_ONLY = 1, 2        # dtsim_step_ex flags
MAP_RELOAD = 0x40000000

EXPORTS = [
    "dtsim_abi_version", "dtsim_last_error", "dtsim_device_count", "dtsim_create", "dtsim_destroy",
    "dtsim_set_assets", "dtsim_set_maps", "dtsim_set_distortion_lut", "dtsim_reset",
    "dtsim_set_spawn_pool", "dtsim_step", "dtsim_step_ex", "dtsim_render", "dtsim_render_ex", "dtsim_set_segment_assets", "dtsim_frames_devptr", "dtsim_frames_bytes",
    "dtsim_bind_frames", "dtsim_draw_lines", "dtsim_draw_leds", "dtsim_allgather_frames", "dtsim_observe", "dtsim_observe_cubic", "dtsim_set_reset_sampler", "dtsim_reset_done", "dtsim_query", "dtsim_read_agent", "dtsim_read", "dtsim_write", "dtsim_field_devptr",
    "dtsim_field_bytes", "dtsim_state_bytes", "dtsim_sync", "dtsim_stream", "dtsim_profile_read",
]


class DtsimLibraryError(RuntimeError):
    pass


class DtsimError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"dtsim error {code}: {msg}")
        self.code = code


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("flags", C.c_uint32), ("num_envs", C.c_int32), ("device", C.c_int32),
        ("cam_width", C.c_int32), ("cam_height", C.c_int32), ("frame_skip", C.c_int32),
        ("max_steps", C.c_int32), ("delay_steps", C.c_int32), ("action_mode", C.c_int32),
        ("delta_time", C.c_double), ("robot_speed", C.c_double),
        ("gain", C.c_double), ("trim", C.c_double), ("radius", C.c_double), ("k", C.c_double),
        ("limit", C.c_double), ("stream", C.c_void_p),
    ]


class Object(C.Structure):
    _fields_ = [
        ("mesh_id", C.c_int32), ("dynamic", C.c_int32), ("collidable", C.c_int32), ("optional", C.c_int32),
        ("pos", C.c_double * 3), ("angle", C.c_double), ("scale", C.c_double),
        ("corners", C.c_double * 8), ("norm", C.c_double * 4), ("safety_radius", C.c_double),
        ("spawn_clear", C.c_double),
        ("walk_distance", C.c_double), ("vel", C.c_double), ("wait_time", C.c_double), ("wiggle", C.c_double),
        ("light_freq", C.c_int32), ("light_pattern", C.c_int32), ("light_tex", C.c_int32 * 2),
        ("light_tris", C.c_int32), ("light_pad", C.c_int32),
    ]


class Map(C.Structure):
    _fields_ = [
        ("grid_w", C.c_int32), ("grid_h", C.c_int32), ("tile_size", C.c_double),
        ("tile_kind", C.POINTER(C.c_uint8)), ("tile_angle", C.POINTER(C.c_uint8)),
        ("tile_tex", C.POINTER(C.c_int16)), ("tile_curve_off", C.POINTER(C.c_int16)),
        ("tile_curve_cnt", C.POINTER(C.c_uint8)),
        ("n_curves", C.c_int32), ("curves", C.POINTER(C.c_double)), ("curve_heads", C.POINTER(C.c_double)),
        ("n_objects", C.c_int32), ("objects", C.POINTER(Object)),
    ]


class Texture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgba", C.POINTER(C.c_uint8))]


class Mesh(C.Structure):
    _fields_ = [("n_tris", C.c_int32), ("verts", C.POINTER(C.c_float)), ("normals", C.POINTER(C.c_float)),
                ("colors", C.POINTER(C.c_float)), ("uvs", C.POINTER(C.c_float)), ("tri_tex", C.POINTER(C.c_int32))]


class ResetSampler(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64), ("domain_rand", C.c_int32), ("dynamics_rand", C.c_int32), ("map_cycle", C.c_int32),
        ("max_attempts", C.c_int32), ("accept_start_angle_deg", C.c_double),
        ("color_sky", C.c_double * 3), ("color_ground", C.c_double * 3), ("start_tile", (C.c_int32 * 2) * MAX_MAPS),
        ("has_start_pose", C.c_int32 * MAX_MAPS), ("start_pose", (C.c_double * 3) * MAX_MAPS),
    ]


class InitState(C.Structure):
    _fields_ = [
        ("pos", C.c_double * 3), ("angle", C.c_double), ("map_id", C.c_int32), ("dynamics_trim_on", C.c_int32),
        ("dynamics_trim", C.c_double), ("wheel_dist", C.c_double), ("cam_height", C.c_double),
        ("cam_angle_deg", C.c_double), ("cam_fov_y_deg", C.c_double), ("camera_noise", C.c_double * 3),
        ("horizon_color", C.c_double * 3), ("ground_color", C.c_double * 3), ("light_pos", C.c_double * 4),
        ("light_ambient", C.c_double * 3), ("light_diffuse", C.c_double * 3),
    ]


class Probe(C.Structure):
    _fields_ = [
        ("tile_i", C.c_int32), ("tile_j", C.c_int32), ("curve_idx", C.c_int32),
        ("drivable", C.c_uint8), ("collision", C.c_uint8), ("valid", C.c_uint8), ("in_lane", C.c_uint8),
        ("inconvenient", C.c_uint8), ("pad", C.c_uint8 * 3),
        ("t", C.c_double), ("point", C.c_double * 2), ("tangent", C.c_double * 2),
        ("dist", C.c_double), ("dot_dir", C.c_double), ("angle_deg", C.c_double), ("angle_rad", C.c_double),
        ("prox", C.c_double), ("reward", C.c_double),
    ]


class AgentInfo(C.Structure):
    """dtsim_agent_info: what Simulator.step returns about one env besides the observation."""
    _fields_ = [
        ("pos", C.c_double * 3), ("angle", C.c_double), ("speed", C.c_double), ("timestamp", C.c_double),
        ("wheels", C.c_double * 2), ("lane", C.c_double * 4), ("prox", C.c_double), ("reward", C.c_double),
        ("tile", C.c_int32 * 2), ("step_count", C.c_int32),
        ("in_lane", C.c_uint8), ("done", C.c_uint8), ("done_code", C.c_uint8), ("pad", C.c_uint8),
    ]


_lib = None


def load(path: str | None = None):
    """Load libdtsim.so and declare every prototype of include/dtsim.h."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("DTSIM_LIB", LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so (SONAME
    # libamdhip64.so.7).  If it is loaded first, libdtsim's DT_NEEDED libamdhip64.so.7
    # resolves to that same copy and both share devices/streams/pointers; loaded the other
    # way round the process ends up with two runtimes and torch sees "No HIP GPUs".
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(p):
        raise DtsimLibraryError(
            f"{p} not found: build it with `python gym-duckietown_amd/build.py` "
            "(hipcc, gfx950).  dtsim has no CPU fallback.")
    lib = C.CDLL(p)
    vp, ci, sz = C.c_void_p, C.c_int, C.c_size_t
    protos = {
        "dtsim_abi_version": (ci, []),
        "dtsim_last_error": (C.c_char_p, []),
        "dtsim_device_count": (ci, []),
        "dtsim_create": (ci, [C.POINTER(Config), C.POINTER(vp)]),
        "dtsim_destroy": (None, [vp]),
        "dtsim_set_assets": (ci, [vp, C.POINTER(Texture), ci, C.POINTER(Mesh), ci]),
        "dtsim_set_maps": (ci, [vp, C.POINTER(Map), ci]),
        "dtsim_set_distortion_lut": (ci, [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
        "dtsim_reset": (ci, [vp, C.POINTER(C.c_uint8), C.POINTER(InitState)]),
        "dtsim_set_spawn_pool": (ci, [vp, C.POINTER(InitState), ci]),
        "dtsim_step": (ci, [vp, vp, ci, ci]),
        "dtsim_step_ex": (ci, [vp, vp, ci, ci, C.c_uint32]),
        "dtsim_render": (ci, [vp]),
        "dtsim_render_ex": (ci, [vp, C.c_uint32]),
        "dtsim_set_segment_assets": (ci, [vp, C.POINTER(Texture), ci, C.POINTER(C.c_uint8), ci]),
        "dtsim_frames_devptr": (vp, [vp]),
        "dtsim_frames_bytes": (sz, [vp]),
        "dtsim_bind_frames": (ci, [vp, vp]),
        "dtsim_allgather_frames": (ci, [vp, vp, vp, vp, C.c_size_t]),
        "dtsim_draw_lines": (ci, [vp, C.POINTER(C.c_float), C.POINTER(C.c_int32), ci]),
        "dtsim_draw_leds": (ci, [vp, C.POINTER(C.c_float), C.POINTER(C.c_int32), ci]),
        "dtsim_set_reset_sampler": (ci, [vp, C.POINTER(ResetSampler)]),
        "dtsim_reset_done": (ci, [vp]),
        "dtsim_observe": (ci, [vp, vp, ci, ci, ci, C.POINTER(C.c_int32), C.POINTER(C.c_int32), ci, C.POINTER(C.c_int32), C.POINTER(C.c_int32), ci]),
        "dtsim_observe_cubic": (ci, [vp, vp, ci, ci, ci, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
        "dtsim_query": (ci, [vp, ci, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_double, C.POINTER(Probe)]),
        "dtsim_read_agent": (ci, [vp, ci, C.POINTER(AgentInfo)]),
        "dtsim_read": (ci, [vp, ci, vp, sz]),
        "dtsim_write": (ci, [vp, ci, vp, sz]),
        "dtsim_field_devptr": (vp, [vp, ci]),
        "dtsim_field_bytes": (sz, [vp, ci]),
        "dtsim_state_bytes": (sz, [vp]),
        "dtsim_sync": (ci, [vp]),
        "dtsim_stream": (vp, [vp]),
        "dtsim_profile_read": (ci, [vp, ci, C.POINTER(ci), C.POINTER(C.c_double)]),
    }
    for name in EXPORTS:
        fn = getattr(lib, name)  # AttributeError => symbol missing
        fn.restype, fn.argtypes = protos[name]
    if lib.dtsim_abi_version() != ABI_VERSION:
        raise DtsimLibraryError(f"ABI version mismatch: lib {lib.dtsim_abi_version()} != {ABI_VERSION}")
    if path is None:
        _lib = lib
    return lib


PROBE_DTYPE = None


def probe_dtype():
    """numpy structured dtype mirroring dtsim_probe (ctypes layout, incl. padding)."""
    global PROBE_DTYPE
    if PROBE_DTYPE is None:
        import numpy as np
        names, formats, offsets = [], [], []
        cmap = {C.c_int32: "i4", C.c_uint8: "u1", C.c_double: "f8"}
        for name, ct in Probe._fields_:
            off = getattr(Probe, name).offset
            if hasattr(ct, "_length_"):
                fmt = (cmap[ct._type_], (ct._length_,))
            else:
                fmt = cmap[ct]
            names.append(name); formats.append(fmt); offsets.append(off)
        PROBE_DTYPE = np.dtype({"names": names, "formats": formats, "offsets": offsets,
                                "itemsize": C.sizeof(Probe)})
    return PROBE_DTYPE


def check(lib, rc):
    if rc != OK:
        raise DtsimError(rc, lib.dtsim_last_error().decode("utf-8", "replace"))
