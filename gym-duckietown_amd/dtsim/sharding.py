"""Multi-GPU: env-index sharding + the frame exchange of the north star.

Envs are independent (no cross-env term in step/reset), so rank r of R owns the
contiguous env range [r*n, (r+1)*n) and the sim path needs NO collective.  The only exchange
is the learner's: an RCCL all-gather (or gather to rank 0) of the uint8 frame batch plus
reward/done.  One process per GPU; `torch.distributed` backend "nccl" is RCCL on ROCm, "gloo"
is used by the CPU tests.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np


def shard_range(rank: int, world: int, n_total: int) -> Tuple[int, int]:
    """Contiguous env range of `rank`; n_total must divide evenly (weak scaling: n_total = world * n)."""
    if n_total % world:
        raise ValueError(f"{n_total} envs do not shard evenly over {world} ranks")
    n = n_total // world
    return rank * n, (rank + 1) * n


def env_seed(base_seed: Optional[int], global_env: int) -> Optional[int]:
    """Env e of the whole job is seeded base+e on whichever rank owns it, so a sharded run
    reproduces the single-GPU run env for env."""
    return None if base_seed is None else base_seed + global_env


def gather_batch(local, world: int, group=None, dst: Optional[int] = None):
    """All-gather (dst None) or gather-to-root of a per-rank batch tensor [n, ...] along dim 0.
    Returns the [world*n, ...] tensor (on every rank, or on `dst` only -> others get None).
    On xGMI a direct/one-shot exchange is per-link bound; call this on a side
    stream to overlap it with the next step."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return local
    if dst is None:
        out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        if dist.get_backend(group) == "nccl":
            dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        else:
            parts = list(out.chunk(world, dim=0))
            dist.all_gather(parts, local.contiguous(), group=group)
        return out
    rank = dist.get_rank(group)
    parts = None
    out = None
    if rank == dst:
        out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        parts = list(out.chunk(world, dim=0))
    dist.gather(local.contiguous(), parts, dst=dst, group=group)
    return out


class ShardedSimulator:
    """BatchedSimulator for the env range this rank owns (rank/world from torch.distributed)."""

    def __init__(self, map_name, total_envs: int, *, seed: Optional[int] = None, device: Optional[int] = None,
                 rank: Optional[int] = None, world: Optional[int] = None, sim_factory=None, **kw):
        """sim_factory(map_name, n_local, seed=, device=, **kw): the per-rank simulator; BatchedSimulator unless a test
        injects a stand-in (the CPU tests run this class on gloo without a GPU)."""
        import os
        if sim_factory is None:
            from .batched import BatchedSimulator as sim_factory
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
        self.lo, self.hi = shard_range(self.rank, self.world, total_envs)
        dev = int(os.environ.get("LOCAL_RANK", "0")) if device is None else device
        self.sim = sim_factory(map_name, self.hi - self.lo, seed=env_seed(seed, self.lo), device=dev, **kw)

    @classmethod
    def wrap(cls, sim, total_envs: int, rank: int, world: int):
        """A ShardedSimulator around an existing per-rank simulator (bench.py builds its own)."""
        self = cls.__new__(cls)
        self.rank, self.world = int(rank), int(world)
        self.lo, self.hi = shard_range(self.rank, self.world, total_envs)
        self.sim = sim
        return self

    def local_actions(self, global_actions: np.ndarray) -> np.ndarray:
        """Slice [..., N_total, 2] actions to this rank's envs."""
        return np.ascontiguousarray(global_actions[..., self.lo:self.hi, :])

    def step(self, global_actions: np.ndarray, n_steps: int = 1):
        """Step this rank's envs with their slice of the job-wide action array."""
        self.sim.step(self.local_actions(global_actions), n_steps)

    def local_frames(self):
        """This rank's frame batch as a torch tensor (device memory of the simulator, no copy)."""
        import torch
        if hasattr(self.sim, "frames_tensor"):           # stand-in simulators of the CPU tests
            return self.sim.frames_tensor()
        frames = torch.as_tensor(self.sim.frames_device(), device=f"cuda:{self.sim.device_index}")
        self.sim.sync()
        return frames

    def gather_frames(self, dst: Optional[int] = None, group=None):
        return gather_batch(self.local_frames(), self.world, group, dst)

    # ---- the learner's exchange, overlapped with the simulation ---------------------------------------
    # Layout (round 4): per buffer slot ONE preallocated [world*n, ...] tensor on the root; the root's own envs are rendered
    # (or observed) straight into its slice of it, every other rank's batch is received into ITS slice (point-to-point,
    # one transfer per xGMI link into the root) -- no torch.cat, no self-copy, and what the learner gets IS that tensor.
    def _produce(self, buf, what: str, obs):
        """This rank's payload of the current step into `buf` (device memory), complete on return (host-level wait on the
        library's stream: the exchange that reads `buf` is enqueued on another stream).  what = "frames": the render pass
        writes there directly (dtsim_bind_frames); "observe": render into the library's buffer, dtsim_observe into `buf`."""
        if what == "frames":
            if hasattr(self.sim, "render_into"):         # stand-in simulators of the CPU tests
                self.sim.render_into(buf)
                return
            self.sim.bind_frames(buf.data_ptr())
            self.sim.render()
            self.sim.sync()
            return
        if hasattr(self.sim, "observe_into"):            # stand-in simulators of the CPU tests
            self.sim.observe_into(buf, *obs[:2])
            return
        self.sim.render()
        self.sim.observe(obs[0], obs[1], out=buf, **(obs[2] if len(obs) > 2 else {}))
        self.sim.sync()

    def _payload_shape(self, what: str, obs):
        n = self.hi - self.lo
        if hasattr(self.sim, "frames_tensor"):           # stand-in simulators of the CPU tests
            f = self.sim.frames_tensor()
            return ((n,) + tuple(f.shape[1:]) if what == "frames" else (n, obs[0], obs[1], 3)), f.dtype, f.device
        import torch
        dev = torch.device(f"cuda:{self.sim.device_index}")
        if what == "frames":
            return (n, self.sim.camera_height, self.sim.camera_width, 3), torch.uint8, dev
        kw = obs[2] if len(obs) > 2 else {}
        shape = (n, 3, obs[0], obs[1]) if kw.get("chw") else (n, obs[0], obs[1], 3)
        return shape, (torch.float32 if kw.get("normalize") else torch.uint8), dev

    def _exchange_state(self, what: str, obs, dst: int, group):
        import torch
        import torch.distributed as dist
        key = (what, tuple(obs[:2]) if obs else None, dst)
        gx = getattr(self, "_gx", None)
        if gx is not None and gx["key"] == key:
            return gx
        shape, dtype, dev = self._payload_shape(what, obs)
        rank = dist.get_rank(group) if self.world > 1 else 0
        n = shape[0]
        slots = []
        for _ in range(2):
            if rank == dst:
                recv = torch.empty((self.world * n,) + tuple(shape[1:]), dtype=dtype, device=dev)
                send = recv[rank * n:(rank + 1) * n]       # the root produces in place
            else:
                recv, send = None, torch.empty(shape, dtype=dtype, device=dev)
            slots.append({"recv": recv, "send": send, "works": None})
        gx = self._gx = {"key": key, "slots": slots, "t": 0, "rank": rank, "n": n, "what": what}
        return gx

    def _start_exchange(self, slot, gx, dst: int, group):
        """Asynchronous gather-to-root of slot's payload: receives into the slices of the root's tensor, one send per other rank."""
        import torch.distributed as dist
        slot["waited"] = False
        if self.world == 1:
            slot["works"] = []
            return
        n, rank = gx["n"], gx["rank"]
        if rank == dst:
            ops = [dist.P2POp(dist.irecv, slot["recv"][r * n:(r + 1) * n], r, group) for r in range(self.world) if r != dst]
        else:
            ops = [dist.P2POp(dist.isend, slot["send"], dst, group)]
        slot["works"] = list(dist.batch_isend_irecv(ops))
        slot["waited"] = False

    @staticmethod
    def _wait(slot, host: bool):
        """Wait for the exchange that uses `slot`.  On RCCL a Work.wait() only orders torch's CURRENT STREAM behind the
        transfer; the library renders on its own stream, so before a buffer is produced into again the HOST waits
        (host=True) -- stream-level ordering is enough for handing the tensor to torch consumers (host=False)."""
        works = slot["works"]
        if works is None:
            return
        if not slot.get("waited"):                       # (a second wait() on a gloo receive blocks for ever)
            for w in works:
                w.wait()
            slot["waited"] = True
        if host:
            t = slot["send"]
            if t.is_cuda:
                import torch
                torch.cuda.current_stream(t.device).synchronize()
            slot["works"] = None                         # only now: nothing reads or writes the slot's buffers any more

    def step_render_gather(self, global_actions, n_steps: int = 1, *, overlap: bool = True, dst: int = 0, group=None,
                           local_actions: bool = False, what: str = "frames", obs=None):
        """One learner iteration: step this rank's envs, render them, and gather the batch to rank `dst`.

        what="frames": the [n, H, W, 3] uint8 frame batch (3.77 GB per rank at the BASELINE size: xGMI-link bound);
        what="observe", obs=(h, w[, observe() keywords]): the dtsim_observe output of the step instead (57.6 KB per env at
        160 x 120: the exchange the north star's learner can actually keep up with).
        overlap=False: blocking; returns (t, batch) -- batch = [world*n, ...] of THIS step on `dst`, None elsewhere.
        overlap=True: the gather of step t runs while step t+1 is simulated and rendered.  Two buffer slots
        rotate; the call returns the batch of step t-1 (the learner runs one step behind the simulator; `flush_gather()`
        hands out the last one).  The tensor returned on `dst` is the slot's preallocated receive tensor itself (no copy):
        it stays valid until the next-but-one call.  A slot is produced into again only after the HOST has seen the end of the
        transfer that read it.  Returns (t-1, batch) or (None, None) on the first call."""
        if what not in ("frames", "observe") or (what == "observe" and not obs):
            raise ValueError("what = 'frames' or 'observe' (with obs=(height, width[, keywords]))")
        if local_actions:                                # already this rank's slice (e.g. a device tensor)
            self.sim.step(global_actions, n_steps)
        else:
            self.step(global_actions, n_steps)
        gx = self._exchange_state(what, obs, dst, group)
        if not overlap:
            t = getattr(self, "_gather_t", 0)
            self._gather_t = t + 1
            slot = gx["slots"][0]
            self._wait(slot, host=True)
            self._produce(slot["send"], what, obs)
            self._start_exchange(slot, gx, dst, group)
            self._wait(slot, host=True)
            return t, slot["recv"]
        t, b = gx["t"], gx["t"] % 2
        slot = gx["slots"][b]
        self._wait(slot, host=True)                      # the transfer of step t-2 read / wrote this slot
        self._produce(slot["send"], what, obs)
        self._start_exchange(slot, gx, dst, group)
        gx["t"] = t + 1
        if t == 0:
            return None, None
        prev = gx["slots"][1 - b]
        self._wait(prev, host=False)
        return t - 1, prev["recv"]

    def flush_gather(self, dst: int = 0):
        """Batch of the last step issued by step_render_gather(overlap=True): (t, batch on `dst` / None)."""
        gx = getattr(self, "_gx", None)
        if gx is None or gx["t"] == 0:
            return None, None
        t = gx["t"] - 1
        slot = gx["slots"][t % 2]
        self._wait(slot, host=True)
        self._wait(gx["slots"][1 - t % 2], host=True)
        if gx["what"] == "frames" and hasattr(self.sim, "bind_frames"):
            self.sim.bind_frames(None)                   # back to the library's own buffer
        return t, slot["recv"]
