"""BatchedSimulator: N Duckietown envs on one MI355X behind libdtsim.so.

Vector counterpart of the reference's Simulator / DuckietownEnv (simulator.py:188,
envs/duckietown_env.py:9): same constructor keywords, same reset/step semantics per
env, batched over env index.  All per-step work happens in the HIP library; this class
only owns host-side set-up (assets, map tables, reset RNG order) and marshals arrays.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional, Sequence, Union

import numpy as np

from . import _ffi, assets, maps, reset as R
from . import distortion as dist_mod


class DeviceArray:
    """Zero-copy view of library-owned device memory (`__cuda_array_interface__`, v2):
    `torch.as_tensor(arr, device="cuda")` wraps it without a copy."""

    def __init__(self, ptr: int, shape, typestr: str, owner):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}
        self._owner = owner


class BatchedSimulator:
    def __init__(self, map_name: Union[str, Sequence[str]] = "small_loop", num_envs: int = 1, *,
                 max_steps: int = 1500, domain_rand: bool = True, frame_rate: float = 30, frame_skip: int = 1,
                 camera_width: int = 640, camera_height: int = 480, robot_speed: float = 1.20,
                 accept_start_angle_deg: float = 60, user_tile_start=None, seed: Optional[int] = None,
                 distortion: bool = False, dynamics_rand: bool = False, camera_rand: bool = False,
                 num_tris_distractors: int = 12, color_ground=(0.15, 0.15, 0.15), color_sky=R.BLUE_SKY,
                 # DuckietownEnv (envs/duckietown_env.py:15)
                 action_mode: str = "wheels", gain=1.0, trim=0.0, radius=0.0318, k=27.0, limit=1.0,
                 # batched / device options
                 render: bool = True, auto_reset: bool = False, delay_steps: Optional[int] = None, device: int = 0,
                 stream: Optional[int] = None, profile: bool = False, actions_f64: bool = False,
                 map_cycle: bool = False, map_random: bool = False, transform_uses_width: bool = False,
                 map_data: Optional[dict] = None,
                 asset_root: Optional[str] = None, style: str = "photos", device_reset: bool = False,
                 undistort: bool = False, per_env_camera: bool = False, light_capture: bool = False,
                 do_reset: bool = True):
        self._lib = _ffi.load()
        self._h = C.c_void_p()
        self._device = int(device)
        self.num_envs = int(num_envs)
        self.max_steps, self.domain_rand = max_steps, bool(domain_rand)
        self.frame_rate, self.frame_skip = frame_rate, int(frame_skip)
        self.delta_time = 1.0 / frame_rate
        if delay_steps is None:
            # the DB18 model's actuation delay is 0.15 s of simulated time (simulator.py:745-755, get_DB18_nominal(delay=0.15)):
            # the smallest k with k * delta_time >= 0.15 -- 5 steps at the default 30 Hz
            delay_steps = int(math.ceil(0.15 * frame_rate - 1e-9))
        if not 0 <= int(delay_steps) <= _ffi.MAX_DELAY:
            raise ValueError(f"delay_steps {delay_steps} (0.15 s at frame_rate {frame_rate}) outside [0, {_ffi.MAX_DELAY}] (DTSIM_MAX_DELAY)")
        self.delay_steps = int(delay_steps)
        self.camera_width, self.camera_height = int(camera_width), int(camera_height)
        self.robot_speed = robot_speed
        self.accept_start_angle_deg = accept_start_angle_deg
        self.user_tile_start = user_tile_start
        self.distortion = bool(distortion)
        self.dynamics_rand, self.camera_rand = bool(dynamics_rand), bool(camera_rand)
        self.num_tris_distractors = num_tris_distractors
        self.color_ground, self.color_sky = color_ground, list(color_sky)
        self.action_mode = action_mode
        self.render_enabled = bool(render)
        self.auto_reset = bool(auto_reset)
        self.actions_f64 = bool(actions_f64)
        self.map_cycle = bool(map_cycle)
        self.map_random = bool(map_random)     # randomize_maps_on_reset: a uniformly drawn map, reloaded, at every reset
        self.seed_value = seed

        flags = 0
        flags |= _ffi.F_RENDER if render else 0
        flags |= _ffi.F_DISTORTION if (render and distortion) else 0
        # per_env_camera: render through the per-env camera / light path (the device flag of domain randomisation) while the reset draws stay
        # those of domain_rand=False -- for callers that write a per-env light or camera themselves (the gym facade's GL light capture)
        self.per_env_camera = bool(per_env_camera)
        if self.per_env_camera and not domain_rand and (device_reset or auto_reset):
            # the device sampler (physics.hip sample_init) always draws a camera_noise, and the per-env render path applies it: with
            # domain_rand off the frames after a device-side reset would be jittered while the reference's are not
            raise ValueError("per_env_camera with domain_rand=False needs host-side resets (device_reset / auto_reset draw a camera noise "
                             "that this render path would apply)")
        flags |= _ffi.F_DOMAIN_RAND if (domain_rand or per_env_camera) else 0
        flags |= _ffi.F_AUTO_RESET if auto_reset else 0
        # light_capture (DTSIM_F_LIGHT_CAPTURE): device-side resets take the new episode's light through the camera of the pose the previous
        # episode ended at, as GL does with reset()'s glLightfv (simulator.py:565-584).  Only the per-env render path has a per-env light.
        self.light_capture = bool(light_capture)
        if self.light_capture and not domain_rand:
            raise ValueError("light_capture needs domain_rand=True: the shared-camera render path lights every env alike (DESIGN.md section 5)")
        flags |= _ffi.F_LIGHT_CAPTURE if self.light_capture else 0
        flags |= _ffi.F_ACTIONS_F64 if actions_f64 else 0
        flags |= _ffi.F_PROFILE if profile else 0
        cfg = _ffi.Config()
        cfg.struct_size = C.sizeof(_ffi.Config)
        cfg.flags, cfg.num_envs, cfg.device = flags, self.num_envs, int(device)
        cfg.cam_width, cfg.cam_height = self.camera_width, self.camera_height
        cfg.frame_skip, cfg.max_steps, cfg.delay_steps = self.frame_skip, int(max_steps), int(delay_steps)
        cfg.action_mode = {"wheels": _ffi.ACTION_WHEELS, "vel_steer": _ffi.ACTION_VEL_STEER}[action_mode]
        cfg.delta_time, cfg.robot_speed = self.delta_time, float(robot_speed)
        cfg.gain, cfg.trim, cfg.radius, cfg.k, cfg.limit = float(gain), float(trim), float(radius), float(k), float(limit)
        cfg.stream = C.c_void_p(stream) if stream else None
        _ffi.check(self._lib, self._lib.dtsim_create(C.byref(cfg), C.byref(self._h)))

        # ---- maps + assets (one-time host prep)
        # Assets come from `asset_root` (or $DTSIM_ASSET_ROOT: a duckietown-world style data tree with
        # MapFormat1 YAML maps, tiles-processed/<style>/<kind>/texture.* and <kind>.obj/.mtl meshes)
        # with the deterministic fixtures of dtsim/assets.py as the fallback.
        self.state_version = 0
        self.library = assets.AssetLibrary(asset_root, style)
        use_lib = self.library if self.library.root else None
        names = [map_name] if isinstance(map_name, str) else list(map_name)
        datas = [map_data] if (map_data is not None) else [self.library.map_data(n) for n in names]
        self.map_names = [assets.map_basename(n) for n in names]
        self.map_datas, self._ctor_map_names = datas, names
        self.meshes: Dict[str, assets.MeshData] = {"duckie": assets.get_mesh("duckie"), "*": assets.get_mesh("*")}
        first = [maps.interpret_map(d, n, self.meshes, transform_uses_width, library=use_lib)
                 for d, n in zip(datas, self.map_names)]
        mesh_order = list(self.meshes)                 # "duckie", "*", then every object mesh the library loaded
        self._mesh_order, self._have_segment_assets = mesh_order, False
        tex_kinds: List[str] = []
        for mt in first:
            for kd in mt.texture_kinds:
                if kd not in tex_kinds:
                    tex_kinds.append(kd)
        self.texture_kinds = tex_kinds
        tile_tex = [self.library.tile_texture(kd) for kd in tex_kinds]
        if tile_tex:                                   # the raster needs one size for all tile textures
            side = max(max(t.shape[0], t.shape[1]) for t in tile_tex)
            tile_tex = [assets.to_pow2(t, side) if t.shape[:2] != (side, side) else t for t in tile_tex]
        self.textures = list(tile_tex)
        mesh_tex_base: Dict[str, int] = {}             # mesh key -> index of its first texture
        for mk in mesh_order:
            mesh_tex_base[mk] = len(self.textures)
            self.textures.extend(self.meshes[mk].textures)
        self.light_tex = (-1, -1)                      # traffic-light cards (only when a map has lights and the tree has cards)
        if any(o.light_freq > 0 for mt in first for o in mt.objects):
            cards = self.library.light_cards()
            if cards is not None:
                self.light_tex = (len(self.textures), len(self.textures) + 1)
                self.textures.extend(cards)
        tex_ids = {kd: i for i, kd in enumerate(tex_kinds)} if render else None
        self.maps: List[maps.MapTables] = [
            maps.interpret_map(d, n, self.meshes, transform_uses_width, texture_ids=tex_ids, library=use_lib)
            for d, n in zip(datas, self.map_names)]
        if render:
            keep = []
            tarr = (_ffi.Texture * max(len(self.textures), 1))()
            for i, t in enumerate(self.textures):
                tarr[i].width, tarr[i].height = t.shape[1], t.shape[0]
                tarr[i].rgba = t.ctypes.data_as(C.POINTER(C.c_uint8))
            marr = (_ffi.Mesh * len(mesh_order))()
            for i, mk in enumerate(mesh_order):
                m = self.meshes[mk]
                marr[i].n_tris = m.n_tris
                marr[i].verts = m.verts.ctypes.data_as(C.POINTER(C.c_float))
                marr[i].normals = m.normals.ctypes.data_as(C.POINTER(C.c_float))
                marr[i].colors = m.colors.ctypes.data_as(C.POINTER(C.c_float))
                if m.textures:
                    gt = np.where(m.tri_tex >= 0, m.tri_tex + mesh_tex_base[mk], -1).astype(np.int32)
                    keep.append(gt)
                    marr[i].uvs = m.uvs.ctypes.data_as(C.POINTER(C.c_float))
                    marr[i].tri_tex = gt.ctypes.data_as(C.POINTER(C.c_int32))
            _ffi.check(self._lib, self._lib.dtsim_set_assets(self._h, tarr, len(self.textures), marr, len(mesh_order)))
        mesh_ids = {mk: i for i, mk in enumerate(mesh_order)} if render else {}
        farr = (_ffi.Map * len(self.maps))()
        for i, mt in enumerate(self.maps):
            farr[i] = mt.to_ffi(mesh_ids, self.light_tex if render else (-1, -1))
        _ffi.check(self._lib, self._lib.dtsim_set_maps(self._h, farr, len(self.maps)))
        self.undistort = bool(undistort and distortion)
        self._skip_distort = False
        if render and distortion:
            # undistort=True: UndistortWrapper(env) -- the simulator's fisheye is skipped (env.undistort, simulator.py:1969)
            # and the wrapper's rectify map is the per-pixel source map instead (wrappers.py:209-227)
            rmx, rmy = (dist_mod.undistort_wrapper_maps if self.undistort else dist_mod.distortion_maps)(
                self.camera_width, self.camera_height)
            self.rmapx, self.rmapy = rmx, rmy
            _ffi.check(self._lib, self._lib.dtsim_set_distortion_lut(
                self._h, rmx.ctypes.data_as(C.POINTER(C.c_float)), rmy.ctypes.data_as(C.POINTER(C.c_float))))

        # ---- per-env host reset state (RNG order lives on the host; reset.py)
        self.env_state = [R.EnvResetState(None if seed is None else seed + e) for e in range(self.num_envs)]
        self.env_map = np.zeros(self.num_envs, np.int32)
        self.init_states = (_ffi.InitState * self.num_envs)()
        self._have_reset = False
        self.device_reset = bool(device_reset)
        if self.device_reset:
            self.install_reset_sampler()
        if do_reset:
            self.reset()

    # ------------------------------------------------------------------ reset --
    def install_reset_sampler(self, seed: Optional[int] = None):
        """Device-side reset sampling (dtsim_reset_sampler): reset() and auto-reset then draw DR values and
        spawn poses on the GPU (the reference's distributions and acceptance test, Philox stream) instead
        of on the host in numpy's RNG order."""
        rs = _ffi.ResetSampler()
        sv = self.seed_value if seed is None else seed
        rs.seed = int(sv if sv is not None else np.random.SeedSequence().entropy) & 0xFFFFFFFFFFFFFFFF
        rs.domain_rand, rs.dynamics_rand = int(self.domain_rand), int(self.dynamics_rand)
        rs.map_cycle = 2 if self.map_random else int(self.map_cycle and len(self.maps) > 1)
        rs.max_attempts = R.MAX_SPAWN_ATTEMPTS
        rs.accept_start_angle_deg = float(self.accept_start_angle_deg)
        rs.color_sky[:] = [float(v) for v in self.color_sky]
        rs.color_ground[:] = [float(v) for v in self.color_ground]
        for m in range(_ffi.MAX_MAPS):
            tile = (-1, -1)
            if m < len(self.maps):
                if self.user_tile_start:
                    tile = tuple(int(v) for v in self.user_tile_start)
                elif self.maps[m].start_tile is not None:
                    tile = tuple(int(v) for v in self.maps[m].start_tile)
            rs.start_tile[m][0], rs.start_tile[m][1] = tile
            sp = self.maps[m].start_pose if m < len(self.maps) else None
            rs.has_start_pose[m] = 0 if sp is None else 1
            if sp is not None:                          # [[x, y, z], angle] relative to the start tile (simulator.py:679-688)
                rs.start_pose[m][0], rs.start_pose[m][1], rs.start_pose[m][2] = float(sp[0][0]), float(sp[0][2]), float(sp[1])
        _ffi.check(self._lib, self._lib.dtsim_set_reset_sampler(self._h, C.byref(rs)))
        self._sampler = rs

    def _map_for_reset(self, e: int) -> int:
        if self.map_random:                    # simulator.py:541-544: np_random.choice(map_names), the first draw of reset()
            return int(self.env_state[e].np_random.integers(0, len(self.maps)))
        if len(self.maps) == 1:
            return 0
        if not self.map_cycle:
            return e % len(self.maps)          # fixed assignment
        es = self.env_state[e]
        es.map_slot = (es.map_slot + 1) % len(self.maps)   # envs/multimap_env.py:46 (first reset -> 1)
        return es.map_slot

    def sample_states(self, envs: Sequence[int]) -> None:
        """Simulator.reset() for the given envs: fills self.init_states[e] (host draws in
        the reference order; geometry on the device via dtsim_query)."""
        envs = [int(e) for e in envs]
        pend = {}
        vis_all = None
        for e in envs:
            mi = self._map_for_reset(e)
            mt = self.maps[mi]
            st, tile, visible = R.draw_prefix(
                self.env_state[e], mt, domain_rand=self.domain_rand, camera_rand=self.camera_rand,
                dynamics_rand=self.dynamics_rand, color_sky=self.color_sky, color_ground=self.color_ground,
                num_tris_distractors=self.num_tris_distractors, n_visible_draw=(), user_tile_start=self.user_tile_start)
            st.map_id = mi | (_ffi.MAP_RELOAD if self.map_random else 0)
            if self.per_env_camera and not self.domain_rand:
                st.camera_noise[:] = [0.0, 0.0, 0.0]             # drawn, but only applied under domain_rand (simulator.py:1768-1769)
            self.init_states[e] = st
            self.env_state[e].spawn_attempts = 0
            if mt.start_pose is not None:                       # simulator.py:679-688
                i, j = tile
                sp = mt.start_pose
                st.pos[:] = [i * mt.tile_size + sp[0][0], 0.0, j * mt.tile_size + sp[0][2]]
                st.angle = float(sp[1])
                self.init_states[e] = st
            else:
                pend[e] = (mt, tile, visible)
        if not pend:
            return
        # The device evaluates candidates against each env's *current* world; a map switch
        # or a first reset must create that world first: provisional reset at a dummy pose.
        need_world = [e for e in pend if (not self._have_reset) or self.map_random
                      or self.env_map[e] != (self.init_states[e].map_id & ~_ffi.MAP_RELOAD)]
        if need_world:
            mask = np.zeros(self.num_envs, np.uint8)
            for e in need_world:
                mask[e] = 1
                self.init_states[e].pos[:] = [0.0, 0.0, 0.0]
                self.init_states[e].angle = 0.0
            self._reset_device(mask)
            for e in need_world:                               # the objects are fresh now: the final reset must keep them
                self.init_states[e].map_id &= ~_ffi.MAP_RELOAD
        self._write_visibility({e: pend[e][2] for e in pend})
        active = dict(pend)
        while active:
            q_env, q_pose, spans = [], [], {}
            for e, (mt, tile, _) in active.items():
                blk = R.attempt_block(self.env_state[e], tile, mt.tile_size)
                spans[e] = (len(q_env), blk)
                q_env.extend([e] * len(blk))
                q_pose.append(blk)
            probes = self.query(np.array(q_env, np.int32), np.concatenate(q_pose, axis=0), safety_factor=1.3)
            nxt = {}
            for e, (off, blk) in spans.items():
                es = self.env_state[e]
                k = len(blk)
                M = self.accept_start_angle_deg
                p = probes[off:off + k]
                ok = (~p["inconvenient"].astype(bool)) & p["valid"].astype(bool) & p["in_lane"].astype(bool) \
                    & (-M < p["angle_deg"]) & (p["angle_deg"] < M)
                budget = R.MAX_SPAWN_ATTEMPTS - es.spawn_attempts
                idx = np.flatnonzero(ok[:budget])
                st = self.init_states[e]
                if idx.size:
                    a = int(idx[0])
                    R.commit_attempts(es, a + 1)
                    st.pos[:] = [float(blk[a, 0]), 0.0, float(blk[a, 1])]
                    st.angle = float(blk[a, 2])
                elif budget <= k:                              # simulator.py:732-736 fallback pose
                    R.commit_attempts(es, budget)
                    st.pos[:] = [1.0, 0.0, 1.0]
                    st.angle = 1.0
                else:
                    R.commit_attempts(es, k)
                    nxt[e] = active[e]
                self.init_states[e] = st
            active = nxt

    def _write_visibility(self, vis: Dict[int, List[bool]]):
        if all(all(v) for v in vis.values()) and not getattr(self, "_vis_dirty", False):
            return
        self._vis_dirty = True
        arr = self.read(_ffi.FIELD_OBJ_VISIBLE)
        for e, v in vis.items():
            arr[e, :] = 1
            arr[e, :len(v)] = np.asarray(v, np.uint8)
        self.write(_ffi.FIELD_OBJ_VISIBLE, arr)

    def _reset_device(self, mask: Optional[np.ndarray]):
        mp = mask.ctypes.data_as(C.POINTER(C.c_uint8)) if mask is not None else None
        _ffi.check(self._lib, self._lib.dtsim_reset(self._h, mp, self.init_states))
        sel = range(self.num_envs) if mask is None else np.flatnonzero(mask)
        for e in sel:
            self.env_map[e] = self.init_states[e].map_id & ~_ffi.MAP_RELOAD
        self._have_reset = True

    def reset(self, mask: Optional[np.ndarray] = None, states=None):
        """Simulator.reset() for the masked envs (None = all).  `states`: optional
        ctypes array / list of _ffi.InitState to use instead of sampling (parity mode)."""
        self.state_version += 1                       # any cached read-back of the state is stale now
        if not self._have_reset:
            mask = None                        # the first reset creates every env's world
        if self.device_reset and states is None:
            m = None if mask is None else np.ascontiguousarray(np.asarray(mask).astype(np.uint8))
            mp = m.ctypes.data_as(C.POINTER(C.c_uint8)) if m is not None else None
            _ffi.check(self._lib, self._lib.dtsim_reset(self._h, mp, None))
            self.env_map[:] = self.read(_ffi.FIELD_MAP_ID)
            self._have_reset = True
            return
        sel = list(range(self.num_envs)) if mask is None else [int(e) for e in np.flatnonzero(mask)]
        if states is not None:
            for e in sel:
                self.init_states[e] = states[e]
        else:
            self.sample_states(sel)
        m = None
        if mask is not None:
            m = np.ascontiguousarray(np.asarray(mask).astype(np.uint8))
        self._reset_device(m)

    def reset_done(self):
        """Restart (device sampler) every env whose done flag is set; asynchronous, no host round trip."""
        self.state_version += 1                       # any cached read-back of the state is stale now
        _ffi.check(self._lib, self._lib.dtsim_reset_done(self._h))

    def field_device(self, field: int) -> DeviceArray:
        """Zero-copy device view of one SoA field ([N, ...] as documented in include/dtsim.h)."""
        dt, shp = self._FIELD_SHAPES[field]
        if shp != ():
            raise ValueError("device views are offered for the scalar per-env fields (the state is SoA: [component][N])")
        ptr = self._lib.dtsim_field_devptr(self._h, field)
        if not ptr:
            raise ValueError(f"field {field} has no contiguous device view")
        ts = {"f8": "<f8", "u1": "|u1", "i4": "<i4"}[dt]
        return DeviceArray(ptr, (self.num_envs,), ts, self)

    def make_spawn_pool(self, n_pool: int):
        """Pre-sample `n_pool` spawn states (env e%N's RNG stream) for DTSIM_F_AUTO_RESET."""
        pool = (_ffi.InitState * n_pool)()
        saved = [(_ffi.InitState.from_buffer_copy(self.init_states[e])) for e in range(self.num_envs)]
        done = 0
        while done < n_pool:
            batch = min(self.num_envs, n_pool - done)
            self.sample_states(range(batch))
            for e in range(batch):
                pool[done + e] = self.init_states[e]
            done += batch
        for e in range(self.num_envs):
            self.init_states[e] = saved[e]
        _ffi.check(self._lib, self._lib.dtsim_set_spawn_pool(self._h, pool, n_pool))
        self._pool = pool
        return pool

    def skip_distort(self, flag: bool):
        """`env.undistort = True` (simulator.py:1968-1970, 2001; set by UndistortWrapper, wrappers.py:209): render_obs
        returns the rectilinear image, i.e. the per-pixel source map becomes the identity; False restores the fisheye."""
        flag = bool(flag) and self.distortion and self.render_enabled
        if flag == self._skip_distort:
            return
        self._skip_distort = flag
        fp = C.POINTER(C.c_float)
        if flag:
            _ffi.check(self._lib, self._lib.dtsim_set_distortion_lut(self._h, fp(), fp()))
        else:
            _ffi.check(self._lib, self._lib.dtsim_set_distortion_lut(self._h, self.rmapx.ctypes.data_as(fp), self.rmapy.ctypes.data_as(fp)))

    # ------------------------------------------------------------------- step --
    def step(self, actions, n_steps: int = 1, flags: int = 0):
        """actions: [n_steps, N, 2] or [N, 2] (float32, or float64 with actions_f64);
        numpy array (host) or an object with __cuda_array_interface__ (device).
        flags: _ffi.STEP_ONE_UPDATE (one update_physics, no frame_skip) / _ffi.STEP_POSE_ONLY (`_update_pos`)."""
        self.state_version += 1                       # any cached read-back of the state is stale now
        if hasattr(actions, "__cuda_array_interface__"):
            ptr = actions.__cuda_array_interface__["data"][0]
            _ffi.check(self._lib, self._lib.dtsim_step_ex(self._h, C.c_void_p(ptr), int(n_steps), 1, int(flags)))
            return
        dt = np.float64 if self.actions_f64 else np.float32
        a = np.ascontiguousarray(np.asarray(actions, dtype=dt))
        if a.size != n_steps * self.num_envs * 2:
            raise ValueError(f"actions has {a.size} elements, expected {n_steps}*{self.num_envs}*2")
        _ffi.check(self._lib, self._lib.dtsim_step_ex(self._h, a.ctypes.data_as(C.c_void_p), int(n_steps), 0, int(flags)))

    def render(self, segment: bool = False, gl_filter: bool = False):
        """render_obs() of every env into the frame batch; `segment=True` is the reference's segmentation render
        (simulator.py:1730-1737,1753,1808,1879).  `gl_filter=True` (DTSIM_RENDER_GL_FILTER): tile textures filtered with the arithmetic of the
        reference's renderer (Mesa llvmpipe's 8-bit GL_LINEAR) by the generic raster -- frames bit-identical to the reference's on 99.2 - 99.96 % of
        the pixels, 2 - 4 x slower than the quad-record kernels: for validation, not for throughput."""
        flags = _ffi.RENDER_GL_FILTER if gl_filter else 0
        if not segment:
            _ffi.check(self._lib, self._lib.dtsim_render_ex(self._h, flags))
            return
        if not self._have_segment_assets:
            self._install_segment_assets()
        _ffi.check(self._lib, self._lib.dtsim_render_ex(self._h, _ffi.RENDER_SEGMENT | flags))

    def segment_assets(self):
        """(segmented textures mirroring self.textures, per-mesh flat colours [n_meshes,3]) -- host prep of the
        segmentation render: tile textures through load_texture(segment=True) (graphics.py:70-126, into black),
        mesh textures and meshes through gen_segmentation_color(mesh_name) (objmesh.py:255-292)."""
        import os
        lib = self.library
        seg_tex = []
        for i, kd in enumerate(self.texture_kinds):
            p = lib.tile_texture_file(kd) if lib.root else None
            hint = p if p else os.path.join("tiles-processed", lib.style, kd, "texture.png")
            seg_tex.append(assets.segment_texture(self.textures[i], hint))
        rgb = np.zeros((len(self._mesh_order), 3), np.uint8)
        k = len(self.texture_kinds)
        for mi, mk in enumerate(self._mesh_order):
            m = self.meshes[mk]
            col = assets.gen_segmentation_color(getattr(m, "seg_name", None) or (mk if len(mk) >= 3 else "object"))
            rgb[mi] = col
            for t in m.textures:                   # only ever sampled through the flat colour (st.tex = -1 on the device)
                seg_tex.append(assets.segment_texture(t, "mesh", col))
                k += 1
        while len(seg_tex) < len(self.textures):   # traffic-light cards: a segmented mesh has no switching card
            seg_tex.append(assets.segment_texture(self.textures[len(seg_tex)], "trafficlight"))
        return seg_tex, rgb

    def _install_segment_assets(self):
        seg_tex, rgb = self.segment_assets()
        tarr = (_ffi.Texture * max(len(seg_tex), 1))()
        for i, t in enumerate(seg_tex):
            tarr[i].width, tarr[i].height = t.shape[1], t.shape[0]
            tarr[i].rgba = t.ctypes.data_as(C.POINTER(C.c_uint8))
        _ffi.check(self._lib, self._lib.dtsim_set_segment_assets(
            self._h, tarr, len(seg_tex), rgb.ctypes.data_as(C.POINTER(C.c_uint8)), len(self._mesh_order)))
        self._have_segment_assets = True

    def frames_device(self) -> DeviceArray:
        ptr = self._lib.dtsim_frames_devptr(self._h)
        return DeviceArray(ptr, (self.num_envs, self.camera_height, self.camera_width, 3), "|u1", self)

    def draw_lines(self, lines, env_idx=None):
        """GL_LINE overlays (the reference's draw_curve / draw_bbox) as a post-pass on the frames of the last render():
        lines [n, 9] = world-space segment (ax, ay, az, bx, by, bz) + colour (r, g, b in 0..1); env_idx [n] (non-decreasing) or None =
        env 0.  dtsim_draw_lines (include/dtsim.h)."""
        a = np.ascontiguousarray(np.asarray(lines, dtype=np.float32).reshape(-1, 9))
        n = a.shape[0]
        if n == 0:
            return
        ip = None
        if env_idx is not None:
            ei = np.ascontiguousarray(np.asarray(env_idx, dtype=np.int32).reshape(-1))
            if ei.shape[0] != n:
                raise ValueError("env_idx needs one entry per segment")
            ip = ei.ctypes.data_as(C.POINTER(C.c_int32))
        _ffi.check(self._lib, self._lib.dtsim_draw_lines(self._h, a.ctypes.data_as(C.POINTER(C.c_float)), ip, int(n)))

    _LED_POS = ((0.1, 0.05, -0.05), (0.1, 0.05, 0.05), (0.1, 0.05, 0.0), (-0.1, 0.05, -0.05), (-0.1, 0.05, 0.05))   # glTranslatef(px, pz, py): front_left,
    # front_right, center, back_left, back_right in the dict's order (objects.py:74-80, 96)
    _LED_FOLLOWER = ((0.5, 0.5, 0.5), (0.5, 0.5, 0.5), (0.0, 0.0, 0.2), (0.5, 0.0, 0.0), (0.5, 0.0, 0.0))           # DuckiebotObj.leds_color (objects.py:218-224)
    _LED_STATIC = ((0.0, 0.0, 1.0),) * 5                                                                             # a static duckiebot-kind WorldObj (objects.py:86-92)

    def led_spheres(self, envs=None):
        """The spheres WorldObj.render_mesh draws with enable_leds (objects.py:68-121) for the given envs (default: all), from the current object
        states: per visible object of kind "duckiebot" and LED, the 1 cm sphere at alpha 1 and the halo of radius mean(colour) x 4 cm at alpha 0.2,
        inside the object's translate / scale / rotate, in draw order.  Returns (spheres [n, 8] float32, env_idx [n] int32) for draw_leds()."""
        envs = range(self.num_envs) if envs is None else [int(e) for e in envs]
        vis_all, cen_all = self.read(_ffi.FIELD_OBJ_VISIBLE), self.read(_ffi.FIELD_OBJ_CENTER)
        yrot_all, cy_all = self.read(_ffi.FIELD_OBJ_YROT), self.read(_ffi.FIELD_OBJ_Y)
        out, idx = [], []
        for e in envs:
            for k, o in enumerate(self.maps[int(self.env_map[e])].objects):
                if o.kind != "duckiebot" or not vis_all[e][k]:
                    continue
                if o.dyn_slot >= 0:
                    pos = np.array([cen_all[e][o.dyn_slot, 0], cy_all[e][o.dyn_slot], cen_all[e][o.dyn_slot, 1]], dtype=np.float64)
                    th = math.radians(float(yrot_all[e][o.dyn_slot]))
                else:
                    pos, th = np.asarray(o.pos, dtype=np.float64), float(o.angle)
                c, s_ = math.cos(th), math.sin(th)
                for (lx, ly, lz), col in zip(self._LED_POS, self._LED_STATIC if o.static else self._LED_FOLLOWER):
                    col = np.clip(np.asarray(col, dtype=np.float64), 0.0, 1.0)
                    x, y, z = lx * o.scale, ly * o.scale, lz * o.scale
                    cw = np.array([x * c + z * s_, y, -x * s_ + z * c]) + pos         # glRotatef(y_rot, 0, 1, 0) (objects.py:140-146)
                    out.append([*cw, 0.01 * o.scale, *col, 1.0]); idx.append(e)
                    out.append([*cw, float(np.mean(col)) * 0.04 * o.scale, *col, 0.2]); idx.append(e)
        return np.asarray(out, dtype=np.float32).reshape(-1, 8), np.asarray(idx, dtype=np.int32)

    def draw_leds(self, spheres, env_idx=None):
        """The LED spheres of the reference's enable_leds (objects.py:68-121) as a post-pass on the frames of the last render(): spheres [n, 8] =
        world-space centre (x, y, z), radius, colour (r, g, b in 0..1), alpha, in draw order; env_idx [n] (non-decreasing) or None = env 0.
        dtsim_draw_leds (include/dtsim.h)."""
        a = np.ascontiguousarray(np.asarray(spheres, dtype=np.float32).reshape(-1, 8))
        n = a.shape[0]
        if n == 0:
            return
        ip = None
        if env_idx is not None:
            ei = np.ascontiguousarray(np.asarray(env_idx, dtype=np.int32).reshape(-1))
            if ei.shape[0] != n:
                raise ValueError("env_idx needs one entry per sphere")
            ip = ei.ctypes.data_as(C.POINTER(C.c_int32))
        _ffi.check(self._lib, self._lib.dtsim_draw_leds(self._h, a.ctypes.data_as(C.POINTER(C.c_float)), ip, int(n)))

    def bind_frames(self, devptr: Optional[int]):
        _ffi.check(self._lib, self._lib.dtsim_bind_frames(self._h, C.c_void_p(devptr) if devptr else None))

    def allgather_frames(self, nccl_comm: int, recv, send=None):
        """RCCL all-gather of this rank's frame batch (or of `send`, any object with __cuda_array_interface__, e.g. the
        observe() output) into `recv` ([n_ranks, ...] device memory), enqueued on the library's stream behind the last
        render: dtsim_allgather_frames (include/dtsim.h).  `nccl_comm`: the ncclComm_t as an integer (address)."""
        rp = recv.__cuda_array_interface__["data"][0]
        sp, sb = None, 0
        if send is not None:
            ai = send.__cuda_array_interface__
            sp = ai["data"][0]
            sb = int(np.prod(ai["shape"])) * np.dtype(ai["typestr"]).itemsize
        _ffi.check(self._lib, self._lib.dtsim_allgather_frames(self._h, C.c_void_p(nccl_comm), C.c_void_p(rp), C.c_void_p(sp), C.c_size_t(sb)))

    def observe(self, height: int, width: int, chw: bool = False, normalize: bool = False, out=None, interpolation: str = "pil_bilinear"):
        """Learner-side observation of the last rendered batch, on the device: PIL-exact bilinear resize
        (learning/utils/wrappers.py ResizeWrapper) or, with interpolation="cv_cubic", the cv2 INTER_CUBIC resize of the
        reference's own ResizeWrapper (src/gym_duckietown/wrappers.py:129-138); optional HWC->CHW (ImgWrapper) and /255
        float32 (NormalizeWrapper).  Returns a device array ([N,h,w,3] or [N,3,h,w]); `out` may be any object with
        __cuda_array_interface__ of that shape/dtype (e.g. the send buffer of the frame all-gather)."""
        import torch
        from . import resample
        if interpolation not in ("pil_bilinear", "cv_cubic"):
            raise ValueError(f"interpolation {interpolation!r}: 'pil_bilinear' or 'cv_cubic'")
        h, w = int(height), int(width)
        shape = (self.num_envs, 3, h, w) if chw else (self.num_envs, h, w, 3)
        if out is None:
            key = (h, w, chw, normalize)
            if getattr(self, "_obs_key", None) != key:
                self._obs_buf = torch.empty(shape, dtype=torch.float32 if normalize else torch.uint8,
                                            device=f"cuda:{self.device_index}")
                self._obs_key = key
            out = self._obs_buf
        ptr = out.__cuda_array_interface__["data"][0]
        ip = C.POINTER(C.c_int32)
        flags = (_ffi.OBS_CHW if chw else 0) | (_ffi.OBS_F32 if normalize else 0)
        if interpolation == "cv_cubic":
            fx, tx = resample.cubic_coeffs(self.camera_width, w)
            fy, ty = resample.cubic_coeffs(self.camera_height, h)
            fx, tx, fy, ty = (np.ascontiguousarray(a, dtype=np.int32) for a in (fx, tx, fy, ty))
            _ffi.check(self._lib, self._lib.dtsim_observe_cubic(self._h, C.c_void_p(ptr), h, w, flags, fx.ctypes.data_as(ip), tx.ctypes.data_as(ip),
                                                                fy.ctypes.data_as(ip), ty.ctypes.data_as(ip)))
            return out
        bx = kx = by = ky = None
        nkx = nky = 0
        if w != self.camera_width:
            b, k = resample.coeffs(self.camera_width, w)
            bx, kx, nkx = b.ctypes.data_as(ip), k.ctypes.data_as(ip), k.shape[1]
        if h != self.camera_height:
            b, k = resample.coeffs(self.camera_height, h)
            by, ky, nky = b.ctypes.data_as(ip), k.ctypes.data_as(ip), k.shape[1]
        _ffi.check(self._lib, self._lib.dtsim_observe(self._h, C.c_void_p(ptr), h, w, flags, bx, kx, nkx, by, ky, nky))
        return out

    def frames_host(self) -> np.ndarray:
        """Synchronous copy of the frame batch to the host (tests / N=1 facade)."""
        import torch
        self.sync()
        t = torch.as_tensor(self.frames_device(), device=f"cuda:{self.device_index}")
        return t.cpu().numpy()

    @property
    def device_index(self) -> int:
        return self._device

    # ----------------------------------------------------------------- fields --
    _FIELD_SHAPES = {
        _ffi.FIELD_POS: ("f8", (3,)), _ffi.FIELD_ANGLE: ("f8", ()), _ffi.FIELD_REWARD: ("f8", ()),
        _ffi.FIELD_DONE: ("u1", ()), _ffi.FIELD_DONE_CODE: ("u1", ()), _ffi.FIELD_STEP_COUNT: ("i4", ()),
        _ffi.FIELD_TILE: ("i4", (2,)), _ffi.FIELD_LANE: ("f8", (4,)), _ffi.FIELD_IN_LANE: ("u1", ()),
        _ffi.FIELD_PROX: ("f8", ()), _ffi.FIELD_SPEED: ("f8", ()), _ffi.FIELD_TIMESTAMP: ("f8", ()),
        _ffi.FIELD_WHEELS: ("f8", (2,)), _ffi.FIELD_MAP_ID: ("i4", ()),
        _ffi.FIELD_OBJ_CENTER: ("f8", (_ffi.MAX_DYNAMIC, 2)), _ffi.FIELD_OBJ_ACTIVE: ("u1", (_ffi.MAX_DYNAMIC,)),
        _ffi.FIELD_OBJ_YROT: ("f8", (_ffi.MAX_DYNAMIC,)), _ffi.FIELD_OBJ_PARAMS: ("f8", (_ffi.MAX_DYNAMIC, 3)),
        _ffi.FIELD_OBJ_VISIBLE: ("u1", (_ffi.MAX_OBJECTS,)), _ffi.FIELD_EPISODE: ("i4", ()),
        _ffi.FIELD_OBJ_LIGHT: ("u1", (_ffi.MAX_OBJECTS,)), _ffi.FIELD_OBJ_Y: ("f8", (_ffi.MAX_DYNAMIC,)),
        _ffi.FIELD_OBJ_EXTRA: ("f8", (_ffi.MAX_DYNAMIC, 5)), _ffi.FIELD_CAMERA: ("f4", (6,)), _ffi.FIELD_COLORS: ("f4", (16,)),
        _ffi.FIELD_WHEEL_DIST: ("f8", ()), _ffi.FIELD_RENDER_POS: ("i4", ()),
    }

    def read(self, field: int) -> np.ndarray:
        if field == _ffi.FIELD_STATE_BLOB:
            out = np.empty(self._lib.dtsim_state_bytes(self._h), np.uint8)
        else:
            dt, shp = self._FIELD_SHAPES[field]
            out = np.empty((self.num_envs,) + shp, dtype=dt)
        _ffi.check(self._lib, self._lib.dtsim_read(self._h, field, out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def read_agent(self, env: int = 0) -> "_ffi.AgentInfo":
        """dtsim_read_agent: pose, speed, wheels, lane pose, proximity, reward / done of one env in one transfer."""
        out = _ffi.AgentInfo()
        _ffi.check(self._lib, self._lib.dtsim_read_agent(self._h, int(env), C.byref(out)))
        return out

    def write(self, field: int, arr: np.ndarray):
        self.state_version += 1                       # any cached read-back of the state is stale now
        if field == _ffi.FIELD_STATE_BLOB:
            a = np.ascontiguousarray(arr, np.uint8)
        else:
            dt, shp = self._FIELD_SHAPES[field]
            a = np.ascontiguousarray(np.asarray(arr, dtype=dt).reshape((self.num_envs,) + shp))
        _ffi.check(self._lib, self._lib.dtsim_write(self._h, field, a.ctypes.data_as(C.c_void_p), a.nbytes))

    def query(self, env_idx, poses, safety_factor: float = 1.0) -> np.ndarray:
        """Reference geometry queries at arbitrary poses [n,3] = (x, z, angle), evaluated
        on the device; returns a structured array mirroring dtsim_probe."""
        env_idx = np.ascontiguousarray(env_idx, np.int32)
        poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 3)
        n = poses.shape[0]
        out = np.zeros(n, dtype=_ffi.probe_dtype())
        if n:
            _ffi.check(self._lib, self._lib.dtsim_query(
                self._h, n, env_idx.ctypes.data_as(C.POINTER(C.c_int32)), poses.ctypes.data_as(C.POINTER(C.c_double)),
                float(safety_factor), C.cast(out.ctypes.data, C.POINTER(_ffi.Probe))))
        return out

    def sync(self):
        _ffi.check(self._lib, self._lib.dtsim_sync(self._h))

    @property
    def stream(self) -> int:
        return int(self._lib.dtsim_stream(self._h) or 0)

    def profile_read(self, kernel: int):
        n, ms = C.c_int(), C.c_double()
        _ffi.check(self._lib, self._lib.dtsim_profile_read(self._h, kernel, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    @property
    def state_bytes(self) -> int:
        return int(self._lib.dtsim_state_bytes(self._h))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.dtsim_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
