#!/usr/bin/env python3
"""bench.py -- headline benchmark of the dtsim hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (config.workload): BASELINE.json configs[2] -- Duckietown-small_loop-v0, 4096
batched envs PER GPU, 640x480 RGB raster + fisheye distortion, domain_rand off, random
DuckietownEnv (vel, steer) actions resident in HBM, auto-reset from a pre-sampled spawn
pool.  One "step" = one pass of the hot path over the batch: dtsim_step (kinematics,
dynamics, lane pose, SAT collision, proximity, reward/done) + dtsim_render (raster + fisheye).
Envs are independent, so GPUs shard them with NO data-path collective: `value` is the
sim throughput, scaling "weak".  The north star's RCCL all-gather of the uint8 frame batch
is measured separately (`gather`, xGMI-link bound) and never part of `value`.

Output: ONE JSON line on rank 0 (contract in the task statement) with `roofline`
(dominant kernel = the raster; achieved = algorithmic bytes / HIP-event kernel time) and
`cpu_baseline` (the oracle on 1 host core, plus `all_cores`: one env per process; bounded samples; N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "gym-duckietown_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

PEAK_HBM = 8.0e12          # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
W, H = 640, 480
FRAME_BYTES = W * H * 3    # algorithmic bytes per env-step: the frame, written once


def _oracle_env_steps(job):
    """n oracle env-steps of one env (full Simulator.step incl. software raster and fisheye remap); returns the
    seconds the stepping took.  Top-level so that multiprocessing's spawn context can import it."""
    n_steps, seed = job
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ.setdefault(var, "1")               # one core per env, as the reference scales (one env per process)
    from dtsim import assets
    from dtsim import distortion as pdist
    from oracle import raster, sim as osim
    ext = assets.mesh_extents(("duckie",))
    o = osim.OracleSim(assets.get_map("small_loop"), ext, domain_rand=False, seed=seed)
    kinds = {t["kind"] for t in o.map.grid if t is not None}
    scene = raster.Scene(o.map, {k: assets.get_texture(k) for k in kinds},
                         {"duckie": assets.get_mesh("duckie"), "*": assets.get_mesh("*")})
    rmap = pdist.distortion_maps(W, H)
    rng = np.random.default_rng(1234 + seed)
    t0 = time.perf_counter()
    for _ in range(n_steps):
        a = rng.uniform(-1, 1, 2)
        _, done, _ = o.step_vel_steer(a)
        cam = raster.Camera(o.cur_pos, o.cur_angle, width=W, height=H, horizon_color=o.horizon_color,
                            ground_color=o.ground_color)
        raster.render_obs(cam, scene, "gouraud", rmap)
        if done:
            o.reset()
    return time.perf_counter() - t0


def _glport_env_steps(job):
    """n env-steps of one env through the reference's CPU path as it can travel: oracle/sim.py physics (restated reference step) + the
    reference's OpenGL call stream on Mesa llvmpipe (oracle/gl/glport.py, byte-identical to the reference's frames on the GL goldens) +
    the fisheye remap.  One core: LP_NUM_THREADS=1.  Returns the seconds the stepping took."""
    n_steps, seed, fisheye = job
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "LP_NUM_THREADS"):
        os.environ.setdefault(var, "1")
    from dtsim import assets
    from dtsim import distortion as pdist
    from oracle import raster, sim as osim
    from oracle.gl import glport
    ext = assets.mesh_extents(("duckie",))
    o = osim.OracleSim(assets.get_map("small_loop"), ext, domain_rand=False, seed=seed)
    kinds = {t["kind"] for t in o.map.grid if t is not None}
    scene = raster.Scene(o.map, {k: assets.get_texture(k) for k in kinds}, {"duckie": assets.get_mesh("duckie"), "*": assets.get_mesh("*")})
    r = glport.GLRenderer(scene, W, H)
    r.set_light(list(o.light_pos) + [0.0] * (4 - len(o.light_pos)), o.light_ambient, o.light_diffuse)
    take = None
    if fisheye:                                                          # cv2.remap(INTER_NEAREST, BORDER_CONSTANT 0) as one gather (raster.distort, indices precomputed)
        rmap = pdist.distortion_maps(W, H)
        sx, sy = np.rint(rmap[0].astype(np.float64)).astype(np.int64), np.rint(rmap[1].astype(np.float64)).astype(np.int64)
        ok = (sx >= 0) & (sx < W) & (sy >= 0) & (sy < H)
        take = np.where(ok, sy * W + sx, W * H).reshape(-1)              # index W * H = one black pixel appended to the frame
    rng = np.random.default_rng(1234 + seed)
    t0 = time.perf_counter()
    for _ in range(n_steps):
        _, done, _ = o.step_vel_steer(rng.uniform(-1, 1, 2))
        img = r.render(o.cur_pos, o.cur_angle, horizon=o.horizon_color, ground=o.ground_color)
        if take is not None:
            img = np.concatenate([img.reshape(-1, 3), np.zeros((1, 3), np.uint8)])[take].reshape(H, W, 3)
        if done:
            o.reset()
            r.set_light(list(o.light_pos) + [0.0] * (4 - len(o.light_pos)), o.light_ambient, o.light_diffuse)
    return time.perf_counter() - t0


def _pool_rate(fn, jobs, steps_each):
    import multiprocessing as mp
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(len(jobs)) as pool:                # spawn: the parent holds a HIP context
        spent = pool.map(fn, jobs)
    return len(jobs) * steps_each / max(spent), time.perf_counter() - t0


def cpu_baseline(n_steps: int, all_cores_steps: int = 8):
    """The reference's CPU path on the bench box's host cores, kind "port" (the reference itself -- Python under /root/reference -- does not
    exist here): the reference's OpenGL call stream on Mesa llvmpipe, the reference CI's renderer (oracle/gl/glport.py; pinned
    byte-identical to the reference's own frames by tests/test_gl_golden.py), with oracle/sim.py's physics -- ONE core (`value`) and, SURVEY
    8(d), all cores with one env per process, the only scaling the reference supports (`all_cores`).  `numpy_port`: the software-rasteriser
    oracle of the earlier rounds (oracle/raster.py), for continuity.  `reference_recorded`: the unmodified reference's own step loop, timed
    in the build container (profiles/reference_render_timings.json).  Same map / resolution / action distribution as the GPU workload."""
    out = {}
    procs = min(os.cpu_count() or 1, 64)
    gl_reason = None
    try:
        from oracle.gl import glport, glshim
        if not glport.available():
            gl_reason = "Mesa's swrast_dri.so / GL headers are not on this host"
    except Exception as ex:                                              # pragma: no cover
        gl_reason = repr(ex)[:200]
    if gl_reason is None:
        try:
            import subprocess
            # the single-core sample in a child process: llvmpipe's context must not live in the process that holds the HIP context
            n_gl = max(60, 12 * n_steps)
            code = ("import sys, json; sys.path[:0] = [%r, %r]; import bench; from oracle.gl import glshim; "
                    "dt = bench._glport_env_steps((%d, 1000, True)); print(json.dumps({'dt': dt, 'renderer': glshim.renderer()}))" % (ROOT, os.path.join(ROOT, "gym-duckietown_amd"), n_gl))
            res = subprocess.run([sys.executable, "-W", "ignore", "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, LP_NUM_THREADS="1"))
            rec = json.loads(res.stdout.strip().splitlines()[-1])
            out = {"value": n_gl / rec["dt"], "unit": "env-steps/s", "cores": 1, "kind": "port", "renderer": rec["renderer"],
                   "sample": f"{n_gl} env-steps of small_loop 640x480 + fisheye: oracle/sim.py physics + the reference's OpenGL call stream "
                             "(simulator.py:1707-1951 restated in oracle/gl/glport.py, byte-identical to the reference's frames on the GL goldens) on Mesa "
                             "llvmpipe with LP_NUM_THREADS=1 + a numpy remap; /root/reference itself (Python) does not exist on this host"}
            if all_cores_steps > 0 and procs > 1:
                per = max(40, 5 * all_cores_steps)
                rate, wall = _pool_rate(_glport_env_steps, [(per, 2000 + i, True) for i in range(procs)], per)
                out["all_cores"] = {"value": rate, "unit": "env-steps/s", "cores": procs,
                                    "sample": f"{procs} processes x {per} env-steps, one env and one llvmpipe context per process (LP_NUM_THREADS=1); "
                                              f"stepping time of the slowest process (pool start-up excluded; wall {wall:.1f} s)"}
        except Exception as ex:
            gl_reason = "the GL port failed here: " + repr(ex)[:200]
            out = {}
    dt = _oracle_env_steps((n_steps, 1000))
    numpy_port = {"value": n_steps / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
                  "sample": f"{n_steps} env-steps of small_loop 640x480 + fisheye on the numpy oracle (oracle/sim.py + the software rasteriser oracle/raster.py)"}
    if not out:
        out = dict(numpy_port)
        out["renderer"] = None
        out["gl_port_skipped"] = gl_reason
        if all_cores_steps > 0 and procs > 1:
            try:
                rate, wall = _pool_rate(_oracle_env_steps, [(all_cores_steps, 2000 + i) for i in range(procs)], all_cores_steps)
                out["all_cores"] = {"value": rate, "unit": "env-steps/s", "cores": procs,
                                    "sample": f"{procs} processes x {all_cores_steps} env-steps, one env per process; stepping time of the slowest process (wall {wall:.1f} s)"}
            except Exception as ex:
                out["all_cores"] = {"error": repr(ex)[:200]}
    else:
        out["numpy_port"] = numpy_port
    try:
        with open(os.path.join(ROOT, "profiles", "reference_render_timings.json")) as f:
            rr = json.load(f)
        out["reference_recorded"] = {"status": "recorded in the build container, not measured in this run", "what": rr["what"], "where": rr["where"],
                                     "renderer": rr["renderer"], "env_steps_per_s": {k: v["env_steps_per_s"] for k, v in rr["cases"].items()},
                                     "glport_same_host": rr.get("glport_same_host"), "recipe": "oracle/time_reference_render.py"}
    except Exception:
        pass
    return out


def main_c2(args):
    """BASELINE.json configs[1]: small_loop, render off.  Latency-bound at N=4096 (working set in L2);
    pass --envs 1048576 for the HBM-roofline reading."""
    import torch
    from dtsim import BatchedSimulator, _ffi
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    N, K, Wm, F = args.envs, args.steps, args.warmup, args.fuse
    sim = BatchedSimulator("small_loop", min(N, 4096), render=False, domain_rand=False, seed=1000, action_mode="vel_steer",
                           auto_reset=True, profile=True, device=local_rank, do_reset=False)
    pool = sim.make_spawn_pool(min(N, 4096))
    if N > 4096:   # large-N reading: replicate the sampled spawn states (reset sampling is host-side)
        big = BatchedSimulator("small_loop", N, render=False, domain_rand=False, seed=1000, action_mode="vel_steer",
                               auto_reset=True, profile=True, device=local_rank, do_reset=False)
        states = (_ffi.InitState * N)()
        for e in range(N):
            states[e] = pool[e % len(pool)]
        big._lib.dtsim_set_spawn_pool(big._h, pool, len(pool))
        sim.close()
        sim = big
        sim.reset(states=states)
    else:
        sim.reset(states=pool)
    dev = torch.device("cuda", local_rank)
    acts = torch.from_numpy(np.random.default_rng(1234).uniform(-1.0, 1.0, (F, N, 2)).astype(np.float32)).to(dev)   # SURVEY 8(d) protocol
    for _ in range(Wm):
        sim.step(acts, n_steps=F)
    sim.sync(); sim.profile_read(_ffi.KERNEL_STEP)
    t0 = time.perf_counter()
    for _ in range(K):
        sim.step(acts, n_steps=F)
    sim.sync()
    dt = time.perf_counter() - t0
    n_s, ms_s = sim.profile_read(_ffi.KERNEL_STEP)
    # state actually touched per env-step on a map without dynamic objects: 31 f64 arrays (pose, SE(2) state,
    # velocities, 2x5 delay ring, gains, wheel_dist, timestamp, speed, reward, lane[4], prox, wheels[2]),
    # 4 int32 (ring head, step count, tile i/j), 3 uint8 (done, code, in_lane); read + written once per step,
    # plus the 8-byte f32 action.  (dtsim_state_bytes()/N also counts the 8 unused DuckieObj slots.)
    S = 31 * 8 + 4 * 4 + 3
    b_phys = 2 * S + 8
    achieved = N * F * b_phys / (ms_s / n_s * 1e-3)
    print(json.dumps({
        "metric": "env-steps/sec (dynamics+collision only, render off)", "value": N * F * K / dt, "unit": "env-steps/s",
        "n_gpus": 1, "steps": K, "warmup": Wm, "ms_per_step": 1e3 * dt / K, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "Duckietown-small_loop-v0 (fixture), batched envs, render off [BASELINE.json configs[1]]",
                   "envs_per_gpu": N, "fused_steps_per_launch": F},
        "roofline": {"bound": "hbm", "kernel": "k_step", "achieved": achieved / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                     "frac": achieved / PEAK_HBM, "traffic": None, "kernel_ms": ms_s / n_s,
                     "algorithmic_bytes_per_env_step": b_phys, "state_bytes_touched_per_env": S,
                     "state_bytes_allocated_per_env": sim.state_bytes / N,
                     "note": "with fused steps the state stays in L2 between steps: read frac as HBM-bound only at --fuse 1"},
        "cpu_baseline": None}), flush=True)
    sim.close()


def main_c1(args):
    """BASELINE.json configs[0]: Duckietown-small_loop-v0, 1 env, random actions, 84x84 observations -- the reference's
    CPU-runnable plumbing case.  Here: the drop-in gym_duckietown.DuckietownEnv (an N = 1 view of the HIP library) stepped
    with reset-on-done like a gym loop, observations copied to the host every step as gym requires; beside it the CPU
    oracle doing the same work (oracle physics + software raster at 84x84)."""
    from gym_duckietown.envs import DuckietownEnv
    K, Wm = args.steps, args.warmup
    env = DuckietownEnv(map_name="small_loop", domain_rand=False, seed=1000, camera_width=84, camera_height=84,
                        device=int(os.environ.get("LOCAL_RANK", "0")))
    rng = np.random.default_rng(1234)
    env.reset()
    for _ in range(Wm):
        _, _, d, _ = env.step(rng.uniform(-1, 1, 2))
        if d:
            env.reset()
    t0 = time.perf_counter()
    for _ in range(K):
        obs, _, d, _ = env.step(rng.uniform(-1, 1, 2))
        if d:
            env.reset()
    dt = time.perf_counter() - t0
    assert obs.shape == (84, 84, 3)
    cpu = None
    if args.cpu_steps > 0:
        from dtsim import assets
        from oracle import raster, sim as osim
        ext = assets.mesh_extents(("duckie",))
        o = osim.OracleSim(assets.get_map("small_loop"), ext, domain_rand=False, seed=1000)
        kinds = {t["kind"] for t in o.map.grid if t is not None}
        scene = raster.Scene(o.map, {k: assets.get_texture(k) for k in kinds},
                             {"duckie": assets.get_mesh("duckie"), "*": assets.get_mesh("*")})
        n = max(4, args.cpu_steps)
        t1 = time.perf_counter()
        for _ in range(n):
            _, dn, _ = o.step_vel_steer(rng.uniform(-1, 1, 2))
            cam = raster.Camera(o.cur_pos, o.cur_angle, width=84, height=84, horizon_color=o.horizon_color, ground_color=o.ground_color)
            raster.render_obs(cam, scene, "gouraud", None)
            if dn:
                o.reset()
        cpu = {"value": n / (time.perf_counter() - t1), "unit": "env-steps/s", "cores": 1, "kind": "port",
               "sample": f"{n} env-steps of small_loop at 84x84 on the numpy oracle (the software rasteriser; bench.py --config c3 times the reference's GL call stream on llvmpipe)"}
    k_ms = 1e3 * dt / K
    print(json.dumps({
        "metric": "env-steps/sec (1 env, 84x84 obs, gym loop)", "value": K / dt, "unit": "env-steps/s", "n_gpus": 1, "steps": K,
        "warmup": Wm, "ms_per_step": k_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 raster -> u8 frames; f64 physics", "data": "synthetic",
        "config": {"workload": "Duckietown-small_loop-v0 (fixture), 1 env through gym_duckietown.DuckietownEnv, random (vel, steer) "
                               "actions, 84x84 observations copied to the host every step [BASELINE.json configs[0]]",
                   "envs_per_gpu": 1, "camera": [84, 84]},
        "roofline": {"bound": "hbm", "achieved": 84 * 84 * 3 / (k_ms * 1e-3) / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                     "frac": 84 * 84 * 3 / (k_ms * 1e-3) / PEAK_HBM, "traffic": None,
                     "note": "one env is launch- and host-round-trip bound (several kernel launches + a D2H copy per step); "
                             "the roofline fraction is reported for completeness only"},
        "cpu_baseline": cpu}), flush=True)
    env.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--cpu-steps", type=int, default=32, help="oracle env-steps for cpu_baseline (about 15 s on one host core)")
    ap.add_argument("--windows", type=int, default=5, help="timed regions of --steps steps each; value = the median one")
    ap.add_argument("--no-gather", action="store_true", help="skip the RCCL frame all-gather measurement")
    ap.add_argument("--gather-timeout", type=float, default=120.0, help="give up on the frame exchange after this many seconds")
    ap.add_argument("--config", default="c3", choices=["c3", "c1", "c2", "c4", "c5", "c3dr"],
                    help="c3 (default, the headline): raster + fisheye; c1: ONE env through the gym facade, 84x84 observations, "
                         "random actions, with the CPU oracle beside it (BASELINE.json configs[0], the plumbing case); "
                         "c2: dynamics+collision only, render off "
                         "(BASELINE.json configs[1]); a step is then `--fuse` physics steps in one launch; "
                         "c4: loop_pedestrians + domain randomisation (configs[3]); c5: MultiMap, two maps alternating "
                         "per env slot (configs[4]); c3dr: what gym.make('Duckietown-small_loop-v0') runs -- the registered defaults "
                         "domain_rand=True, distortion=False (simulator.py:213,223) -- at N = 4096: context for users, not a BASELINE config")
    ap.add_argument("--fuse", type=int, default=32, help="c2: physics steps fused per dtsim_step launch")
    args = ap.parse_args()
    if args.config == "c2":
        return main_c2(args)
    if args.config == "c1":
        return main_c1(args)

    import torch
    import torch.distributed as dist
    from dtsim import BatchedSimulator, _ffi

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    # Test hook for 1-GPU boxes: DTSIM_BENCH_ONE_GPU=1 puts every rank on device 0 and uses gloo for the
    # (control-plane only) collectives, so the multi-rank code path can be exercised without 2 GPUs.
    one_gpu = os.environ.get("DTSIM_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1 or "RANK" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)   # (the exchange legs then stage through pinned host memory)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    N, K, Wm = args.envs, args.steps, args.warmup
    variant = {
        "c3": dict(maps="small_loop", dr=False, label="Duckietown-small_loop-v0 (fixture)", ref="configs[2]",
                   kern="k_raster_v3<OBJ=0> (exact path of the edge pixels inside; k_pix_setup tables cached)", extra={}),
        "c4": dict(maps="loop_pedestrians", dr=True, label="Duckietown-loop_pedestrians-v0 (stand-in: loop_only_duckies with "
                   "static: False, 8 walking duckies), dynamic obstacles + domain randomisation", ref="configs[3]",
                   kern="k_obj_setup + k_raster_v3dr<OBJ=1> (domain randomisation on the quad records) + k_resolve + k_resolve_obj", extra={}),
        "c5": dict(maps=["loop_only_duckies", "small_loop_only_duckies"], dr=False, label="MultiMap-v0 (loop_only_duckies / "
                   "small_loop_only_duckies alternating per env slot, multimap_env.py:17,44-49)", ref="configs[4]",
                   kern="k_obj_setup + k_raster_v3<OBJ=1> + k_resolve_obj", extra=dict(map_cycle=True)),
        "c3dr": dict(maps="small_loop", dr=True, distortion=False, label="Duckietown-small_loop-v0 (fixture) with the registered defaults of "
                     "gym.make: domain_rand on, NO fisheye", ref="none: the gym.make defaults, simulator.py:213,223",
                     kern="k_raster_v3dr<OBJ=0> + k_resolve_dr", extra={}),
    }[args.config]
    sim = BatchedSimulator(variant["maps"], N, domain_rand=variant["dr"], distortion=variant.get("distortion", True), camera_width=W, camera_height=H,
                           seed=1000 + rank * N, action_mode="vel_steer", auto_reset=True, profile=True,
                           device=local_rank, do_reset=False, light_capture=bool(variant["dr"]),   # (DR: resets light the new episode as GL does; free)
                           **variant["extra"])
    t_setup = time.perf_counter()
    sim.make_spawn_pool(N)                     # reference-order resets, geometry evaluated on the GPU
    sim.reset(states=sim._pool)                # start from the first pool entry of each env
    t_setup = time.perf_counter() - t_setup

    dev = torch.device("cuda", local_rank)
    # SURVEY 8(d) input protocol: np.random.default_rng(1234).uniform(-1, 1, (T, N_total, 2)) as float32, uploaded once; a rank takes
    # the columns of ITS global env range (the slice ShardedSimulator.local_actions makes); T = warm-up + K, cycled over the windows
    acts_all = np.random.default_rng(1234).uniform(-1.0, 1.0, (K + Wm, world * N, 2)).astype(np.float32)
    acts = torch.from_numpy(np.ascontiguousarray(acts_all[:, rank * N:(rank + 1) * N, :])).to(dev)   # resident in HBM
    del acts_all

    def one_step(t):
        sim.step(acts[t])
        sim.render()

    def sync_all():
        sim.sync()
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
            torch.cuda.synchronize()

    for t in range(Wm):
        one_step(t)
    sync_all()
    sim.profile_read(_ffi.KERNEL_RENDER)
    sim.profile_read(_ffi.KERNEL_STEP)
    # The timed region: EXACTLY K steps between two (barrier + synchronize) brackets, MAX over ranks.  It is repeated
    # args.windows times (each window is such a region of K steps; the box-to-box and run-to-run spread of a 40 ms region is
    # the size of most deltas this repo reports) and `value` comes from the MEDIAN window; all of them are in `windows_ms`.
    win = []
    for wi in range(max(1, args.windows)):
        sync_all()
        t0 = time.perf_counter()
        for t in range(Wm, Wm + K):
            one_step(t)
        sim.sync()
        torch.cuda.synchronize()
        tw = torch.tensor([time.perf_counter() - t0], device="cpu" if one_gpu else dev, dtype=torch.float64)
        if dist.is_initialized():
            dist.barrier()
            dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        win.append(float(tw.item()))
    n_r, ms_r = sim.profile_read(_ffi.KERNEL_RENDER)
    n_s, ms_s = sim.profile_read(_ffi.KERNEL_STEP)
    win_sorted = sorted(win)
    t_local = win_sorted[len(win_sorted) // 2]
    # run-to-run spread of the render pass: a separate short series AFTER the timed region, one HIP-event reading per launch
    per_launch = []
    for t in range(min(K, 16)):
        one_step(Wm + (t % K))
        sim.sync()
        n1, ms1 = sim.profile_read(_ffi.KERNEL_RENDER)
        if n1 == 1:
            per_launch.append(ms1)
    sim.profile_read(_ffi.KERNEL_STEP)
    per_launch.sort()
    t_max = t_local                            # already the max over ranks, per window

    guard = {"marker": None}

    def emit(gather_, cpu_=None, to=None):
        """Print THE json line (to=None), or return it as a string (to="str": what the guard process holds back)."""
        k_ms = ms_r / max(n_r, 1)
        achieved = N * FRAME_BYTES / (k_ms * 1e-3)
        # HBM-side traffic and VALU instruction counts cannot be collected inside this process (rocprofv3 PMC passes
        # wrap the whole command): they are read from the committed summary of the same command's counter runs and
        # labelled with where they came from; null when that file is for another workload.
        traffic = traffic_source = valu = traffic_split = None
        traffic_why = "no committed counter summary for this config under profiles/ (tools/prof_bench_short.sh + tools/make_traffic_json.py)"
        pj = os.path.join(ROOT, "profiles", "raster_pmc_latest.json" if args.config == "c3" else f"raster_pmc_{args.config}.json")
        if os.path.exists(pj):
            try:
                d = json.load(open(pj))
                traffic_why = (f"the committed counter summary is for config {d.get('config', 'c3')} at {d.get('envs')} envs per GPU; this run is "
                               f"{args.config} at {N}: PMC passes wrap the whole command (rocprofv3), they cannot be taken in-process")
                if d.get("envs") == N and d.get("config", "c3") == args.config:
                    traffic_why = None
                    traffic = d.get("hbm_bytes_per_launch")
                    traffic_source = d.get("source")
                    if d.get("write_KB") is not None and d.get("fetch_KB_raw") is not None:
                        # what the one number is made of: WRITE_SIZE is the frame (1.04-1.05 x the algorithmic bytes: the exact
                        # path's byte patches); 2 x FETCH_SIZE counts L2 FILLS, nine tenths of them 16-byte record gathers on a
                        # 7 MB pool (4 MB of L2 per XCD) that the 256 MB Infinity Cache holds -- an upper bound of the HBM reads
                        # (profiles/r03_fetch_calibration.txt, profiles/r03_variants_ab.txt blocks H and I)
                        traffic_split = {"write_bytes": d["write_KB"] * 1024.0, "l2_read_fill_bytes": 2.0 * d["fetch_KB_raw"] * 1024.0,
                                         "write_over_algorithmic": d["write_KB"] * 1024.0 / (N * FRAME_BYTES),
                                         "note": "traffic = write_bytes + l2_read_fill_bytes over every kernel of the render pass; the read fills are an UPPER "
                                                 "bound of HBM reads (record gathers of a 7 MB pool resident in the 256 MB Infinity Cache; the counter "
                                                 "cannot tell a fill from the Infinity Cache from one from HBM)"
                                                 + ("" if args.config == "c3" else "; with mesh objects also the per-env screen-space triangles and the queue")}
                    if d.get("valu_per_pixel") is not None:
                        clk = d.get("clock_ghz", 2.1) * 1e9
                        # issue peak as MEASURED on this part (profiles/r02_ubench_valu_rates.txt: the opcodes of this kernel
                        # sustain one wave64 instruction per ~4.8 cycles per SIMD at 8 wavefronts per SIMD, ~5.4 at 4), at the
                        # clock the chip holds under this kernel (GRBM_GUI_ACTIVE / duration)
                        peak = 1024 * clk / 4.8
                        ach = d["valu_per_pixel"] * N * W * H / 64.0 / (k_ms * 1e-3)
                        valu = {"ops_per_pixel": d["valu_per_pixel"], "source": d.get("source"), "unit": "wave64 instructions/s",
                                "peak": peak, "achieved": ach, "frac": ach / peak,
                                "peak_basis": "1024 SIMDs x clock / 4.8 cycles per instruction (measured issue rate, profiles/r02_ubench_valu_rates.txt)"}
            except Exception:
                pass
        # the reference's own lane / collision / reward functions per call.  They need /root/reference,
        # which the bench box does not have: the numbers recorded in the build container are reported, labelled as such.
        ref_fn = {"status": "not runnable on the bench box (no /root/reference)"}
        rj = os.path.join(ROOT, "profiles", "reference_function_timings.json")
        if os.path.exists(rj):
            try:
                rd = json.load(open(rj))
                ref_fn = {"status": "recorded, not measured in this run", "us_per_call": rd.get("us_per_call"), "where": rd.get("where"),
                          "what": rd.get("what"), "recipe": "oracle/time_reference_functions.py"}
            except Exception:
                pass
        cfg_extra = {}
        if args.config in ("c4", "c5"):   # object placement of the 8x7 loop maps depends on the reading of get_transform (absent package)
            cfg_extra["get_transform"] = ("README semantics (duckietown_world absent: object rows of the 8x7 loop maps differ by one tile under the "
                                          "reference's grid_width call, tests/test_transform_readings.py)")
        line = {
            "metric": "env-steps/sec (4096 envs, 640x480 RGB) at 1/2/4/8 MI355X; % HBM roofline",
            "value": world * N * K / t_max,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": Wm,
            "ms_per_step": 1e3 * t_max / K,
            "windows_ms": {"min": 1e3 * win_sorted[0], "median": 1e3 * t_max, "max": 1e3 * win_sorted[-1], "n": len(win), "all": [1e3 * w for w in win],
                           "note": f"{len(win)} timed regions of exactly {K} steps each (barrier + synchronize on both sides, max over ranks); "
                                   "value and ms_per_step are those of the median region"},
            "actions": "np.random.default_rng(1234).uniform(-1, 1, (warmup + steps, n_gpus * envs, 2)).astype(float32), uploaded once; rank r takes columns [r*envs, (r+1)*envs)",
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 raster -> u8 frames; f64 physics",
            "data": "synthetic",
            "config": {"workload": f"{variant['label']}, {N} batched envs per GPU, 640x480 RGB raster{' + fisheye distortion' if variant.get('distortion', True) else ''}"
                                   f", domain_rand {'on' if variant['dr'] else 'off'}, random (vel, steer) actions, "
                                   f"auto-reset from spawn pool [BASELINE.json {variant['ref']}]",
                       "envs_per_gpu": N, "camera": [W, H], "distortion": bool(variant.get("distortion", True)), "domain_rand": variant["dr"],
                       "parallelism": f"env-sharded x{world}, no data-path collective", **cfg_extra},
            "roofline": {"bound": "hbm", "kernel": f"dtsim_render pass = k_cam_setup + {variant['kern']} (HIP events around the launches)", "achieved": achieved / 1e9, "peak": PEAK_HBM / 1e9,
                         "unit": "GB/s", "frac": achieved / PEAK_HBM, "traffic": traffic, "traffic_source": traffic_source,
                         "traffic_null_reason": traffic_why, "traffic_split": traffic_split,
                         "kernel_ms": k_ms, "launches": n_r, "algorithmic_bytes_per_launch": N * FRAME_BYTES,
                         "kernel_ms_per_launch": ({"min": per_launch[0], "median": per_launch[len(per_launch) // 2], "max": per_launch[-1],
                                                   "n": len(per_launch), "source": "separate series after the timed region, one HIP-event reading per launch"}
                                                  if per_launch else None),
                         "step_kernel_ms": ms_s / max(n_s, 1), "valu": valu},
            "cpu_baseline": cpu_,
            "reference_functions": ref_fn,
            "gather": gather_,
            "episodes_per_env": done_frac,
            "setup_s": t_setup,
        }
        try:                                   # RCCL's version banner sits in the C stdio buffer until exit: push it out
            import ctypes                      # first, so that the JSON line is the last line of stdout
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        if to == "str":
            return json.dumps(line)
        if guard["marker"]:                    # the guard process (below) stands down: this process prints the line itself
            try:
                open(guard["marker"], "w").close()
            except OSError:
                pass
        print(json.dumps(line), flush=True)

    # ---- optional: frame all-gather over RCCL/xGMI (north star; link-bound, not in `value`)
    gather = None
    force_gather = os.environ.get("DTSIM_BENCH_FORCE_GATHER") == "1" and dist.is_initialized()
    done_frac = float(sim.read(_ffi.FIELD_EPISODE).mean())
    watchdog = None
    if (world > 1 or force_gather) and not args.no_gather:
        # The exchange is reported beside `value`, never inside it: if a collective stalls (a hung link would
        # block inside RCCL, where no exception can be caught) the already-measured line is still printed.
        import threading

        def _give_up():
            if rank == 0:
                emit({"error": f"frame exchange did not finish within {args.gather_timeout:.0f} s; skipped"})
            os._exit(0)

        watchdog = threading.Timer(args.gather_timeout, _give_up)
        watchdog.daemon = True
        watchdog.start()
        if rank == 0:
            # ... and if this process DIES inside the exchange (the RCCL transport between GPUs has never run on hardware: no multi-GPU
            # box was available to any round), a guard process prints the line measured so far: it waits for this pid to go away and
            # writes the held-back line to the inherited stdout unless the marker says the line was printed here.
            try:
                import subprocess
                import tempfile
                held = tempfile.NamedTemporaryFile("w", delete=False, suffix=".json", prefix="dtsim_bench_")
                held.write(emit({"error": "the process ended inside the frame exchange; this is the line measured before it"}, to="str"))
                held.close()
                guard["marker"] = held.name + ".done"
                code = ("import os,sys,time\npid=%d\nwhile True:\n try:\n  os.kill(pid,0)\n except OSError:\n  break\n time.sleep(0.2)\n"
                        "if not os.path.exists(%r):\n sys.stdout.write(open(%r).read()+'\\n'); sys.stdout.flush()\n"
                        "for f in (%r,%r):\n try:\n  os.remove(f)\n except OSError:\n  pass\n" % (os.getpid(), guard["marker"], held.name, held.name, guard["marker"]))
                subprocess.Popen([sys.executable, "-c", code], stdin=subprocess.DEVNULL)
            except Exception:
                guard["marker"] = None
        if os.environ.get("DTSIM_BENCH_DIE_IN_GATHER") == "1" and rank == 0:
            os._exit(17)                           # test hook: what a crash inside the exchange looks like (tests/test_gpu_bench_two_ranks.py)
        frames = torch.as_tensor(sim.frames_device(), device=dev)
        rccl = dist.get_backend() == "nccl"
        rdev = dev if rccl else "cpu"                   # where the control-plane scalars of this leg live
        from dtsim.sharding import ShardedSimulator, gather_batch
        try:
            # the all-gather of the frame batch through the product API: on RCCL the library's own collective
            # (dtsim_allgather_frames, enqueued on the simulator's stream behind the render pass); on gloo (DTSIM_BENCH_ONE_GPU)
            # staged through host memory
            ssa = ShardedSimulator.wrap(sim, world * N, rank, world)
            ks = min(K, 3)
            sync_all()
            tg = time.perf_counter()
            for t in range(ks):
                one_step(Wm + t)
                out = ssa.gather_frames()
            torch.cuda.synchronize()
            tg = time.perf_counter() - tg
            tgt = torch.tensor([tg], device=rdev, dtype=torch.float64)
            dist.all_reduce(tgt, op=dist.ReduceOp.MAX)
            gather = {"value": world * N * ks / float(tgt.item()), "unit": "env-steps/s", "steps": ks,
                      "collective": ("ShardedSimulator.gather_frames(): dtsim_allgather_frames (ncclAllGather of the uint8 frames on the simulator's stream)"
                                     if rccl else "ShardedSimulator.gather_frames(): gloo all_gather of the uint8 frames through host memory"),
                      "bytes_per_rank_per_step": int(frames.numel()), "transport": "rccl" if rccl else "gloo (host staged)",
                      "checksum_ok": bool(int(out[rank * N:(rank + 1) * N].to(torch.int64).sum().item()) == int(frames.to(torch.int64).sum().item()))}
            del out, ssa
            # what a learner on rank 0 needs: gather-to-root instead of all-gather, double-buffered so
            # that the exchange of step t overlaps the simulation of step t+1 (the sim runs on its own HIP stream;
            # only the buffer about to be overwritten is waited for)
            try:
                from dtsim.sharding import ShardedSimulator
                ss = ShardedSimulator.wrap(sim, world * N, rank, world)     # the product API: dtsim/sharding.py
                kp = min(K, 6)
                sync_all()
                tg = time.perf_counter()
                for t in range(kp):
                    ss.step_render_gather(acts[Wm + t], overlap=True, dst=0, local_actions=True, what="frames", slots=2)   # (2 x world x 3.77 GB on the root)
                ss.flush_gather(dst=0)
                torch.cuda.synchronize()
                tg = time.perf_counter() - tg
                sim.bind_frames(None)
                tgt = torch.tensor([tg], device=rdev, dtype=torch.float64)
                dist.all_reduce(tgt, op=dist.ReduceOp.MAX)
                gather["to_root_overlapped"] = {"value": world * N * kp / float(tgt.item()), "unit": "env-steps/s", "steps": kp,
                                                "collective": "ShardedSimulator.step_render_gather(what='frames', slots=2, overlap=True): gather(uint8 frames, dst=0), "
                                                              "two buffers rotating through dtsim_bind_frames, the exchange of step t behind step t+1, stream-ordered by events",
                                                "root_receive_bytes": ShardedSimulator.exchange_root_bytes(2, world, (N, H, W, 3), 1)}
                del ss
            except Exception as ex:
                sim.bind_frames(None)
                gather["to_root_overlapped"] = {"error": repr(ex)[:200]}
            # the same exchange on what learners consume: 160x120 observations made on the device
            # (dtsim_observe, PIL-exact bilinear): 16x fewer bytes over xGMI
            obs = torch.as_tensor(sim.observe(120, 160), device=dev)
            out = torch.empty((world,) + tuple(obs.shape), dtype=torch.uint8, device=dev) if rccl else None
            sync_all()
            tg = time.perf_counter()
            for t in range(ks):
                one_step(Wm + t)
                sim.observe(120, 160)
                sim.sync()
                if rccl:
                    dist.all_gather_into_tensor(out, obs)
                else:
                    out = gather_batch(obs.cpu(), world)
            torch.cuda.synchronize()
            tg = time.perf_counter() - tg
            tgt = torch.tensor([tg], device=rdev, dtype=torch.float64)
            dist.all_reduce(tgt, op=dist.ReduceOp.MAX)
            gather["observations"] = {"value": world * N * ks / float(tgt.item()), "unit": "env-steps/s", "shape": [120, 160, 3],
                                      "collective": "all_gather_into_tensor(uint8 160x120 observations from dtsim_observe)",
                                      "bytes_per_rank_per_step": int(obs.numel())}
            del out
            # ... and through the product API: the learner's loop on observations (gather-to-root of step t behind step t+1)
            try:
                from dtsim.sharding import ShardedSimulator
                ss = ShardedSimulator.wrap(sim, world * N, rank, world)
                kp = min(K, 6)
                sync_all()
                tg = time.perf_counter()
                for t in range(kp):
                    ss.step_render_gather(acts[Wm + t], overlap=True, dst=0, local_actions=True)
                ss.flush_gather(dst=0)
                torch.cuda.synchronize()
                tg = time.perf_counter() - tg
                tgt = torch.tensor([tg], device=rdev, dtype=torch.float64)
                dist.all_reduce(tgt, op=dist.ReduceOp.MAX)
                gather["observations_to_root_overlapped"] = {"value": world * N * kp / float(tgt.item()), "unit": "env-steps/s", "steps": kp,
                                                             "collective": "ShardedSimulator.step_render_gather(overlap=True, dst=0) with its defaults: what='observe', obs=(120, 160), three slots",
                                                             "root_receive_bytes": ShardedSimulator.exchange_root_bytes(3, world, (N, 120, 160, 3), 1)}
                gather["learner_path"] = "observations_to_root_overlapped (the default payload of step_render_gather: what a learner can keep up with over xGMI; the full-frame legs are context)"
                del ss
            except Exception as ex:
                gather["observations_to_root_overlapped"] = {"error": repr(ex)[:200]}
        except Exception as ex:  # e.g. OOM on small-memory parts
            gather = {"error": repr(ex)[:200]}
        watchdog.cancel()

    cpu = None
    if rank == 0 and world == 1 and args.cpu_steps > 0 and args.config == "c3":
        cpu = cpu_baseline(args.cpu_steps)

    if rank == 0:
        emit(gather, cpu)
    sim.close()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
