"""GPU parity AT the BASELINE.json configurations themselves (not at reduced sizes / with parts switched off):

  C3  configs[2]: small_loop, 4096 envs, 640x480 + fisheye -- >= 64 envs of the 4096-env batch, stratified over the
      render order (one per (tile under the camera, heading quadrant) bin of k_env_sort: the first and the last of the
      order, every XCD slice), rendered by the oracle DIRECTLY (no transitive small batch);
  C4  configs[3]: loop_pedestrians, domain randomisation + fisheye, 640x480, after 260 steps (duckies mid-walk);
  C5  configs[4]: MultiMap, both *_only_duckies maps alternating per env slot, 640x480 + fisheye, shared camera:
      k_raster_v3<OBJ> + k_resolve_obj over more than one env chunk.

  C4 / C5 at N = 4096 (round 4): the same two configurations at the batch size the bench lines are quoted on -- 32 (C4) and 16 (C5)
      envs of the 4096-env batch picked like the C3 test (first / last env, both sides of 32-env chunk borders, the middle of
      every XCD's eighth of the chunks, the tail chunk), each rendered by the oracle directly.

Reference: simulator.py:1707-1951 (_render_img), objects.py:384-431 (DuckieObj.step), envs/multimap_env.py:44-49.
Thresholds ("Tolerances"): plane-only scenes 1e-3 / 5e-4 / 0.02 (fraction of pixels beyond +-1, beyond +-2, mean |error| in
1/255); scenes with mesh objects 2e-3 / 1e-3 / 0.03 (silhouette pixels of the meshes flip coverage where float32 edge functions
meet the oracle's float64 ones).  The oracle is oracle/raster.py in its "pixel" lighting mode.
"""
import numpy as np
import pytest

from dtsim import BatchedSimulator, _ffi
from dtsim import distortion as pdist
from oracle import raster
from test_gpu_render import _camera, _obj_states, _scene, _stats

pytestmark = pytest.mark.gpu
W, H = 640, 480
OBJ_TOL = dict(frac_gt1=2e-3, frac_gt2=1e-3, mean=0.03)     # scenes with mesh objects


def _stratified_picks(sim, N, n_min, seed):
    """Env indices of an N-env batch that exercise the raster's work decomposition.  The decomposition goes by POSITION in the
    render order of the pass that just ran (DTSIM_FIELD_RENDER_POS: k_env_sort's order on the quad-record paths, the identity
    elsewhere): 64 consecutive positions (32 before round 6) share a workgroup chunk, XCD x owns the x-th eighth of the chunks (render_v3.inc:
    XCD-affine workgroup map).  Picked positions: the first / last of the order, both sides of chunk borders, the middle of
    every XCD's eighth (a chunk border there), the tail chunk -- mapped back to env indices -- plus envs 0 and N - 1."""
    pos = sim.read(_ffi.FIELD_RENDER_POS)
    assert sorted(pos.tolist()) == list(range(N))          # a permutation of the batch
    env_at = np.argsort(pos, kind="stable")                # position -> env
    at = [0, 1, 31, 32, 33, 63, 64, N - 1, N - 2, N - 32, N - 33]
    for x in range(8):
        m = x * (N // 8) + N // 16
        at += [m - 1, m]
    picks = [int(env_at[p]) for p in at if 0 <= p < N] + [0, N - 1]
    rng = np.random.default_rng(seed)
    while len(set(picks)) < n_min:
        picks.append(int(rng.integers(N)))
    return sorted(set(picks))


def _frames_of(sim, picks):
    import torch
    frames = torch.as_tensor(sim.frames_device(), device="cuda:0")
    return frames[torch.as_tensor(np.array(picks), device="cuda:0")].cpu().numpy()


def test_c3_full_size_batch_matches_oracle_directly():
    N = 4096
    sim = BatchedSimulator("small_loop", N, camera_width=W, camera_height=H, distortion=True, domain_rand=False, seed=5,
                           action_mode="vel_steer")
    acts = np.random.default_rng(0).uniform(-1, 1, (12, N, 2)).astype(np.float32)
    sim.step(acts, n_steps=12)
    sim.render()
    sim.sync()
    pos, ang = sim.read(_ffi.FIELD_POS), sim.read(_ffi.FIELD_ANGLE)
    # the sort key of k_env_sort (render.hip): tile under the camera centre and heading quadrant
    ts = 0.585
    cx, cz = pos[:, 0] + 0.066 * np.cos(ang), pos[:, 2] - 0.066 * np.sin(ang)
    ti, tj = np.clip(np.floor(cx / ts), 0, 31).astype(int), np.clip(np.floor(cz / ts), 0, 31).astype(int)
    quad = np.floor(ang * (2.0 / np.pi) + 0.5).astype(int) & 3
    key = (((tj << 5) | ti) << 2) | quad
    picks = []
    for k in np.unique(key):                               # one env per bin: first, last and every slice of the order
        picks.append(int(np.nonzero(key == k)[0][0]))
    rng = np.random.default_rng(1)
    rpos = sim.read(_ffi.FIELD_RENDER_POS)                 # the order the pass actually ran in: bins as recomputed above, ties broken on the device
    assert sorted(rpos.tolist()) == list(range(N))
    env_at = np.argsort(rpos)
    assert (np.diff(key[env_at]) >= 0).mean() > 0.98       # sorted by the key (the device bins in float32: an env on a bin border may move)
    picks += [int(env_at[0]), int(env_at[-1]), 0, N - 1, int(env_at[31]), int(env_at[32]), int(env_at[63]), int(env_at[64])]
    picks += [int(env_at[i]) for i in range(N // 16, N, N // 8)]    # the middle of each XCD's eighth of the order
    while len(set(picks)) < 64:
        picks.append(int(rng.integers(N)))
    picks = sorted(set(picks))
    assert len(picks) >= 64
    import torch
    frames = torch.as_tensor(sim.frames_device(), device="cuda:0")
    sub = frames[torch.as_tensor(np.array(picks), device="cuda:0")].cpu().numpy()
    scene = _scene("small_loop")
    rmap = pdist.distortion_maps(W, H)
    worst = dict(frac_gt1=0.0, frac_gt2=0.0, mean=0.0)
    for k, e in enumerate(picks):
        ref = raster.render_obs(_camera(sim, e, W, H, False), scene, "pixel", rmap)
        s = _stats(sub[k], ref)
        assert s["frac_gt1"] <= 1e-3 and s["frac_gt2"] <= 5e-4 and s["mean"] <= 0.02, (e, s)
        for f in worst:
            worst[f] = max(worst[f], s[f])
    print("C3 4096-env batch, %d envs against the oracle: worst" % len(picks), worst)
    sim.close()


def test_c4_config_matches_oracle():
    """loop_pedestrians + domain randomisation + fisheye at 640x480 after the duckies started walking."""
    N, steps = 4, 260
    sim = BatchedSimulator("loop_pedestrians", N, camera_width=W, camera_height=H, distortion=True, domain_rand=True,
                           seed=31, max_steps=100000)
    sim.step(np.zeros((steps, N, 2), np.float32), n_steps=steps)
    assert sim.read(_ffi.FIELD_OBJ_ACTIVE).any()           # somebody is walking
    sim.render()
    frames = sim.frames_host()
    scene = _scene("loop_pedestrians")
    rmap = pdist.distortion_maps(W, H)
    n_obj_px = 0
    for e in range(N):
        cam = _camera(sim, e, W, H, True)
        st = _obj_states(sim, e, scene)
        ref = raster.render_obs(cam, scene, "pixel", rmap, obj_states=st)
        no_obj = raster.render_obs(cam, scene, "pixel", rmap, obj_states=[dict(s_, visible=False) for s_ in st])
        n_obj_px += int((np.abs(ref.astype(int) - no_obj.astype(int)).max(-1) > 0).sum())
        s = _stats(frames[e], ref)
        assert s["frac_gt1"] <= 2e-3 and s["frac_gt2"] <= 1e-3 and s["mean"] <= 0.03, (e, s)
    assert n_obj_px > 200, n_obj_px
    sim.close()


def test_c5_config_matches_oracle():
    """MultiMap: two maps alternating per env slot, shared camera, fisheye, more than one env chunk."""
    N = 72
    names = ["loop_only_duckies", "small_loop_only_duckies"]
    sim = BatchedSimulator(names, N, camera_width=W, camera_height=H, distortion=True, domain_rand=False, seed=17,
                           map_cycle=True, max_steps=100000)
    # multimap_env.py:44-49: every slot starts on map index 1 and moves on at each of ITS resets: restart the even slots
    # once more, so that the two maps alternate over the batch as they do in a running MultiMap-v0 job
    sim.reset(mask=(np.arange(N) % 2 == 0))
    acts = np.random.default_rng(4).uniform(0.1, 0.6, (6, N, 2)).astype(np.float32)
    sim.step(acts, n_steps=6)
    sim.render()
    frames = sim.frames_host()
    mid = sim.read(_ffi.FIELD_MAP_ID)
    assert set(np.unique(mid)) == {0, 1} and mid[0] != mid[1]
    scenes = [_scene(n) for n in names]
    rmap = pdist.distortion_maps(W, H)
    n_obj_px = 0
    for e in (0, 1, 30, 31, 32, 33, 63, 64, 71):           # both maps, chunk borders (64 envs per chunk; 32 before round 6), the tail chunk
        scene = scenes[int(mid[e])]
        cam = _camera(sim, e, W, H, False)
        st = _obj_states(sim, e, scene)
        ref = raster.render_obs(cam, scene, "pixel", rmap, obj_states=st)
        no_obj = raster.render_obs(cam, scene, "pixel", rmap, obj_states=[dict(s_, visible=False) for s_ in st])
        n_obj_px += int((np.abs(ref.astype(int) - no_obj.astype(int)).max(-1) > 0).sum())
        s = _stats(frames[e], ref)
        assert s["frac_gt1"] <= 2e-3 and s["frac_gt2"] <= 1e-3 and s["mean"] <= 0.03, (e, int(mid[e]), s)
    assert n_obj_px > 200, n_obj_px
    sim.close()


def test_c4_full_size_batch_matches_oracle_directly():
    """C4 at the size its bench line is quoted on: 4096 envs of loop_pedestrians with domain randomisation and fisheye after 260
    steps; k_raster_v3dr + k_resolve + k_resolve_obj over 128 chunks, every XCD slice, the persistent work lists wrapped many times."""
    N, steps = 4096, 260
    sim = BatchedSimulator("loop_pedestrians", N, camera_width=W, camera_height=H, distortion=True, domain_rand=True,
                           seed=31, max_steps=100000)
    for _ in range(steps // 52):                           # 260 steps, 52 per launch
        sim.step(np.zeros((52, N, 2), np.float32), n_steps=52)
    assert sim.read(_ffi.FIELD_OBJ_ACTIVE).any()
    sim.render()
    sim.sync()
    picks = _stratified_picks(sim, N, 32, 2)
    assert len(picks) >= 32
    sub = _frames_of(sim, picks)
    scene = _scene("loop_pedestrians")
    rmap = pdist.distortion_maps(W, H)
    worst = dict(frac_gt1=0.0, frac_gt2=0.0, mean=0.0)
    n_obj_px = 0
    for k, e in enumerate(picks):
        cam = _camera(sim, e, W, H, True)
        st = _obj_states(sim, e, scene)
        ref = raster.render_obs(cam, scene, "pixel", rmap, obj_states=st)
        if k < 6:
            no_obj = raster.render_obs(cam, scene, "pixel", rmap, obj_states=[dict(s_, visible=False) for s_ in st])
            n_obj_px += int((np.abs(ref.astype(int) - no_obj.astype(int)).max(-1) > 0).sum())
        s = _stats(sub[k], ref)
        assert all(s[f] <= OBJ_TOL[f] for f in OBJ_TOL), (e, s)
        for f in worst:
            worst[f] = max(worst[f], s[f])
    assert n_obj_px > 200, n_obj_px
    print("C4 4096-env batch, %d envs against the oracle: worst" % len(picks), worst)
    sim.close()


def test_c5_full_size_batch_matches_oracle_directly():
    """C5's per-GPU share at the size its bench line is quoted on: 4096 envs over both *_only_duckies maps (MultiMap slot
    alternation), shared camera, fisheye: k_raster_v3<OBJ> + k_resolve_obj."""
    N = 4096
    names = ["loop_only_duckies", "small_loop_only_duckies"]
    sim = BatchedSimulator(names, N, camera_width=W, camera_height=H, distortion=True, domain_rand=False, seed=17,
                           map_cycle=True, max_steps=100000)
    sim.reset(mask=(np.arange(N) % 2 == 0))                # multimap_env.py:44-49 (see test_c5_config_matches_oracle)
    acts = np.random.default_rng(4).uniform(0.1, 0.6, (6, N, 2)).astype(np.float32)
    sim.step(acts, n_steps=6)
    sim.render()
    sim.sync()
    mid = sim.read(_ffi.FIELD_MAP_ID)
    assert set(np.unique(mid)) == {0, 1}
    picks = _stratified_picks(sim, N, 16, 3)
    assert len(picks) >= 16 and {int(mid[e]) for e in picks} == {0, 1}
    sub = _frames_of(sim, picks)
    scenes = [_scene(n) for n in names]
    rmap = pdist.distortion_maps(W, H)
    worst = dict(frac_gt1=0.0, frac_gt2=0.0, mean=0.0)
    for k, e in enumerate(picks):
        scene = scenes[int(mid[e])]
        ref = raster.render_obs(_camera(sim, e, W, H, False), scene, "pixel", rmap, obj_states=_obj_states(sim, e, scene))
        s = _stats(sub[k], ref)
        assert all(s[f] <= OBJ_TOL[f] for f in OBJ_TOL), (e, int(mid[e]), s)
        for f in worst:
            worst[f] = max(worst[f], s[f])
    print("C5 4096-env batch, %d envs against the oracle: worst" % len(picks), worst)
    sim.close()


@pytest.mark.parametrize("cfg", ["c5", "c4"])
def test_render_parts_give_the_same_frames(cfg, monkeypatch):
    """DTSIM_RENDER_PARTS (read at dtsim_create): the chunks of the batch in ranges, the exact-path kernels of one range on a second stream
    beside the raster of the next -- every per-position array addressed relative to the range.  Same frames, bit for bit, as the one-part
    launch: 1024 envs = 16 chunks = two parts of 8 (k_raster_v3<OBJ> in the sorted render order for C5, k_raster_v3dr for C4)."""
    N = 1024
    kw = dict(c5=dict(maps=["loop_only_duckies", "small_loop_only_duckies"], dr=False, extra=dict(map_cycle=True)),
              c4=dict(maps="loop_pedestrians", dr=True, extra={}))[cfg]
    out = []
    for parts in ("1", "2"):
        monkeypatch.setenv("DTSIM_RENDER_PARTS", parts)
        sim = BatchedSimulator(kw["maps"], N, camera_width=W, camera_height=H, distortion=True, domain_rand=kw["dr"], seed=5, max_steps=100000,
                               **kw["extra"])
        acts = np.random.default_rng(9).uniform(0.2, 0.9, (4, N, 2)).astype(np.float32)
        sim.step(acts, n_steps=4)
        sim.render()
        out.append(sim.frames_host().copy())
        sim.close()
    assert out[0].std() > 10.0
    assert np.array_equal(out[0], out[1])
