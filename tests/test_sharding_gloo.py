"""`not gpu`: the N>1 path (env sharding + frame/reward exchange) with world_size 2 on gloo."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dtsim import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_frames(lo, hi, h=6, w=8):
    """Deterministic per-env 'frame': what the raster would write for global env e."""
    e = torch.arange(lo, hi, dtype=torch.int64).view(-1, 1, 1, 1)
    yy = torch.arange(h).view(1, -1, 1, 1)
    xx = torch.arange(w).view(1, 1, -1, 1)
    cc = torch.arange(3).view(1, 1, 1, -1)
    return ((e * 7 + yy * 3 + xx * 5 + cc * 11) % 251).to(torch.uint8)


class _FakeSim:
    """Stand-in for BatchedSimulator on CPU: remembers what it was given; its 'frames' are those of the global envs
    its seed says it owns (sharding.env_seed: seed = base + first global env)."""
    BASE = 5000

    def __init__(self, map_name, n, seed=None, device=0, **kw):
        self.n, self.seed, self.device_index = n, seed, device
        self.last_actions = None

    def step(self, actions, n_steps=1):
        self.last_actions = np.asarray(actions)
        self.n_steps_done += n_steps

    def frames_tensor(self):
        lo = self.seed - self.BASE
        return _fake_frames(lo, lo + self.n)

    # -- what step_render_gather needs: frames that depend on the step, written into the buffer it is handed
    n_steps_done = 0

    def render_into(self, buf):
        """The 'render pass' of step t: every byte is a function of (global env, step), written in two halves with a
        pause in between -- a gather that read this buffer while it is being rendered into (a torn frame) would ship a
        mix of two steps."""
        import time
        lo = self.seed - self.BASE
        f = _fake_frames(lo, lo + self.n) + torch.tensor(self.n_steps_done * 13 % 256, dtype=torch.uint8)
        half = buf.shape[1] // 2
        buf[:, :half].copy_(f[:, :half])
        time.sleep(0.02)
        buf[:, half:].copy_(f[:, half:])


    def observe_into(self, buf, h, w):
        """The 'dtsim_observe' of the current step: [n, h, w, 3] bytes that encode (global env, step, size)."""
        lo = self.seed - self.BASE
        buf.copy_(_fake_frames(lo, lo + self.n, h, w) + torch.tensor((self.n_steps_done * 13 + h) % 256, dtype=torch.uint8))


def _step_frames(lo, hi, t):
    return _fake_frames(lo, hi) + torch.tensor(t * 13 % 256, dtype=torch.uint8)


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = sharding.shard_range(rank, world, total)
        frames = _fake_frames(lo, hi)
        allf = sharding.gather_batch(frames, world)                 # all-gather
        ok = bool(torch.equal(allf, _fake_frames(0, total)))
        rew = torch.arange(lo, hi, dtype=torch.float64) * 0.5
        root = sharding.gather_batch(rew, world, dst=0)             # gather to the learner
        if rank == 0:
            ok = ok and bool(torch.equal(root, torch.arange(0, total, dtype=torch.float64) * 0.5))
        else:
            ok = ok and root is None
        # per-env seeds are a function of the GLOBAL env index
        seeds = [sharding.env_seed(1000, e) for e in range(lo, hi)]
        ok = ok and seeds == list(range(1000 + lo, 1000 + hi))
        # ShardedSimulator itself: rank / world from the environment, its env range, the seed of its first env, the
        # action slice it feeds its simulator, and the frame exchange -- with a stand-in simulator that keys
        # everything by GLOBAL env id (seed - base), so a wrong range or slice shows up in the values
        base = 5000
        ss = sharding.ShardedSimulator("small_loop", total, seed=base, sim_factory=_FakeSim, device=0)
        ok = ok and (ss.rank, ss.world, ss.lo, ss.hi) == (rank, world, lo, hi)
        ok = ok and ss.sim.n == hi - lo and ss.sim.seed == base + lo
        acts = np.zeros((3, total, 2), np.float32)
        acts[..., 0] = np.arange(total)[None, :] * 10 + np.arange(3)[:, None]        # value encodes (global env, step)
        acts[..., 1] = -acts[..., 0]
        ss.step(acts, n_steps=3)
        got = ss.sim.last_actions
        want_env = np.arange(lo, hi)[None, :] * 10 + np.arange(3)[:, None]
        ok = ok and got.shape == (3, hi - lo, 2) and got.flags["C_CONTIGUOUS"] and np.array_equal(got[..., 0], want_env)
        allf2 = ss.gather_frames()
        ok = ok and bool(torch.equal(allf2, _fake_frames(0, total)))
        rootf = ss.gather_frames(dst=0)
        ok = ok and ((rank == 0 and bool(torch.equal(rootf, _fake_frames(0, total)))) or (rank != 0 and rootf is None))
        # the learner's loop: gather-to-root of step t overlapped with step t+1, three rotating buffers.
        # Every frame delivered must be the frame of ITS step, whole (the stand-in renders in two halves with a pause).
        ss2 = sharding.ShardedSimulator("small_loop", total, seed=base, sim_factory=_FakeSim, device=0)
        a1 = np.zeros((1, total, 2), np.float32)
        got_steps = []
        T = 7
        for t in range(T):
            tt, fr = ss2.step_render_gather(a1, overlap=True, dst=0, what="frames")
            if t == 0:
                ok = ok and tt is None and fr is None
                continue
            ok = ok and tt == t - 1
            if rank == 0:
                ok = ok and bool(torch.equal(fr, _step_frames(0, total, tt + 1)))   # step index tt has n_steps_done == tt + 1
                got_steps.append(tt)
            else:
                ok = ok and fr is None
            # zero extra copies on the root: what comes back IS the slot's preallocated [world*n, ...] receive tensor, and
            # the root's own envs were rendered straight into its slice of it
            slot = ss2._gx["slots"][tt % sharding.ShardedSimulator.N_SLOTS]
            if rank == 0:
                ok = ok and fr.data_ptr() == slot["recv"].data_ptr() and tuple(fr.shape) == (total,) + tuple(_fake_frames(0, 1).shape[1:])
                ok = ok and slot["send"].data_ptr() == slot["recv"][lo:hi].data_ptr()
            else:
                ok = ok and slot["recv"] is None
        tt, fr = ss2.flush_gather(dst=0)
        ok = ok and tt == T - 1
        if rank == 0:
            ok = ok and bool(torch.equal(fr, _step_frames(0, total, T))) and got_steps == list(range(T - 1))
        # the same loop on the dtsim_observe output (what the learner takes at 160 x 120): 4 x 5 stand-in observations
        ss4 = sharding.ShardedSimulator("small_loop", total, seed=base, sim_factory=_FakeSim, device=0)
        for t in range(4):
            tt, ob = ss4.step_render_gather(a1, overlap=True, dst=0, what="observe", obs=(4, 5))
            if t == 0:
                ok = ok and tt is None and ob is None
            elif rank == 0:
                want = _fake_frames(0, total, 4, 5) + torch.tensor(((tt + 1) * 13 + 4) % 256, dtype=torch.uint8)
                ok = ok and tt == t - 1 and tuple(ob.shape) == (total, 4, 5, 3) and bool(torch.equal(ob, want))
            else:
                ok = ok and ob is None
        tt, ob = ss4.flush_gather(dst=0)
        ok = ok and tt == 3 and ((rank == 0 and bool(torch.equal(ob, _fake_frames(0, total, 4, 5) + torch.tensor((4 * 13 + 4) % 256, dtype=torch.uint8)))) or (rank != 0 and ob is None))
        # blocking variant: the frames of this very step
        ss3 = sharding.ShardedSimulator("small_loop", total, seed=base, sim_factory=_FakeSim, device=0)
        for t in range(3):
            tt, fr = ss3.step_render_gather(a1, overlap=False, dst=0, what="frames")
            ok = ok and tt == t and ((rank == 0 and bool(torch.equal(fr, _step_frames(0, total, t + 1)))) or (rank != 0 and fr is None))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def _worker_lifetime_and_groups(rank, world, port, total, q):
    """Round 5 (advisor, round 4): a batch handed out by step_render_gather survives the NEXT call (three slots); a change of the
    payload drains the old transfers; `dst` is a global rank, also with an explicit group."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        base = 5000
        a1 = np.zeros((1, total, 2), np.float32)
        for overlap in (True, False):
            ss = sharding.ShardedSimulator("small_loop", total, seed=base, sim_factory=_FakeSim, device=0)
            held = None                                  # (step, tensor) of the call before
            for t in range(8):
                tt, fr = ss.step_render_gather(a1, overlap=overlap, dst=0, what="frames")
                if rank == 0 and held is not None:       # obs_t kept next to obs_t+1: still step held[0]'s bytes, whole
                    assert torch.equal(held[1], _step_frames(0, total, held[0] + 1)), (overlap, t, held[0])
                    assert fr is None or fr.data_ptr() != held[1].data_ptr()
                if rank == 0 and fr is not None:
                    assert torch.equal(fr, _step_frames(0, total, tt + 1)), (overlap, t, tt)
                held = (tt, fr) if fr is not None else None
            tt, fr = ss.flush_gather(dst=0)
            assert tt == 7
            if rank == 0:
                assert torch.equal(fr, _step_frames(0, total, 8))
            # copy=True: a private tensor, not a slot
            tt, fr = ss.step_render_gather(a1, overlap=False, dst=0, copy=True, what="frames")
            if rank == 0:
                assert all(fr.data_ptr() != sl["recv"].data_ptr() for sl in ss._gx["slots"]) and torch.equal(fr, _step_frames(0, total, 9))
            # another payload on the same object: the pending transfers are drained, new buffers of the new shape
            for t in range(3):
                tt, ob = ss.step_render_gather(a1, overlap=True, dst=0, what="observe", obs=(4, 5))
                if rank == 0 and ob is not None:
                    assert tuple(ob.shape) == (total, 4, 5, 3)
                    assert torch.equal(ob, _fake_frames(0, total, 4, 5) + torch.tensor(((tt + 10) * 13 + 4) % 256, dtype=torch.uint8))   # the new exchange counts from 0; 9 steps were made before
            ss.flush_gather(dst=0)
            # two slots: a third less memory on the root; the batch handed out is whole until the next call
            for t in range(5):
                tt, fr = ss.step_render_gather(a1, overlap=True, dst=0, what="frames", slots=2)
                assert len(ss._gx["slots"]) == 2
                if rank == 0 and fr is not None:
                    assert torch.equal(fr, _step_frames(0, total, tt + 13)), (t, tt)       # 12 steps were made before; the new exchange counts from 0
            ss.flush_gather(dst=0)
            with pytest.raises(ValueError):
                ss.step_render_gather(a1, slots=4)
            # the default payload is the observation (DEFAULT_OBS), not the full frames
            tt, ob = ss.step_render_gather(a1, overlap=False, dst=0)
            if rank == 0:
                assert tuple(ob.shape) == (total,) + sharding.ShardedSimulator.DEFAULT_OBS + (3,)
            # and back, to the OTHER root
            tt, fr = ss.step_render_gather(a1, overlap=False, dst=1, what="frames")
            assert (fr is not None) == (rank == 1)
            if rank == 1:
                assert torch.equal(fr, _step_frames(0, total, 20))
        # an explicit group: `dst` stays a GLOBAL rank and the peers of the point-to-point transfers are translated with
        # dist.get_global_rank (torch sorts the ranks of a new group, so at world_size 2 group rank == global rank: the
        # translation is the identity here, what is exercised is the group argument on every call of the exchange)
        g = dist.new_group(ranks=[0, 1])
        ss = sharding.ShardedSimulator("small_loop", total, seed=base, sim_factory=_FakeSim, device=0)
        for t in range(3):
            tt, fr = ss.step_render_gather(a1, overlap=False, dst=1, group=g, what="frames")
            assert (fr is not None) == (rank == 1)
            if rank == 1:
                assert torch.equal(fr, _step_frames(0, total, t + 1)), t
        assert ss._gx["peers"] == [0, 1] and ss._gx["is_root"] == (rank == 1)
        q.put((rank, True))
    except Exception as ex:                              # noqa: BLE001
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(ex), ex, ex.__traceback__))[-1500:]))
    finally:
        dist.destroy_process_group()


def test_held_batches_payload_changes_and_groups():
    world, total = 2, 10
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_lifetime_and_groups, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)], res


def test_two_rank_gather_and_sharding():
    world, total = 2, 10
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_shard_range_properties():
    assert sharding.shard_range(0, 8, 32768) == (0, 4096) and sharding.shard_range(7, 8, 32768) == (28672, 32768)
    cover = []
    for r in range(4):
        lo, hi = sharding.shard_range(r, 4, 16)
        cover += list(range(lo, hi))
    assert cover == list(range(16))
    with pytest.raises(ValueError):
        sharding.shard_range(0, 3, 10)
    assert sharding.env_seed(None, 5) is None


def test_root_memory_of_the_overlapped_gather_is_stated():
    """three slots of what="frames" at the north star's size are 90.6 GB of receive tensors on rank 0; the observation
    payload is 16 x smaller; _exchange_state checks the number against the device's free memory before allocating (CUDA only)."""
    S = sharding.ShardedSimulator
    assert S.exchange_root_bytes(3, 8, (4096, 480, 640, 3), 1) == 3 * 8 * 4096 * 480 * 640 * 3 == 90_596_966_400
    assert S.exchange_root_bytes(2, 8, (4096, 120, 160, 3), 1) * 24 == S.exchange_root_bytes(3, 8, (4096, 480, 640, 3), 1)
    assert 0 < S.ROOT_MEMORY_FRACTION < 1 and S.N_SLOTS == 3 and S.DEFAULT_OBS == (120, 160)
