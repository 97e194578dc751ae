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
    def _render_into(self, buf):
        """Render this rank's envs into `buf` ([n, H, W, 3] uint8, device memory) and wait for the pass: the library
        writes its frames wherever dtsim_bind_frames points, so the send buffer IS the frame buffer."""
        if hasattr(self.sim, "render_into"):             # stand-in simulators of the CPU tests
            self.sim.render_into(buf)
            return
        self.sim.bind_frames(buf.data_ptr())
        self.sim.render()
        self.sim.sync()

    def _frame_like(self):
        import torch
        if hasattr(self.sim, "frames_tensor"):
            return torch.empty_like(self.sim.frames_tensor())
        dev = f"cuda:{self.sim.device_index}"
        shape = (self.hi - self.lo, self.sim.camera_height, self.sim.camera_width, 3)
        return torch.empty(shape, dtype=torch.uint8, device=dev)

    def step_render_gather(self, global_actions, n_steps: int = 1, *, overlap: bool = True, dst: int = 0, group=None,
                           local_actions: bool = False):
        """One learner iteration: step this rank's envs, render them, and gather the frame batch to rank `dst`.

        overlap=False: blocking; returns (t, frames) -- frames = [world*n, H, W, 3] of THIS step on `dst`, None elsewhere.
        overlap=True (the design of SURVEY 8e): the gather of step t runs while step t+1 is simulated and rendered.
        Two frame buffers rotate through dtsim_bind_frames: step t renders into buffer t % 2 and its asynchronous
        gather-to-root starts right away; the call returns the frames of step t-1 (whose gather is waited for here),
        i.e. the learner runs one step behind the simulator, and `flush_gather()` hands out the last step's.  A buffer
        is only rendered into again after the gather that reads it has completed (no torn frames): the wait on
        `works[b]` below.  Returns (t-1, frames of step t-1 on `dst` / None elsewhere), or (None, None) on the first call.
        3.77 GB per rank per step is xGMI-link bound: this hides the simulation behind the exchange, it
        does not make the exchange faster -- gather `observe()` output when the learner takes 160x120."""
        import torch.distributed as dist
        if local_actions:                                # already this rank's slice (e.g. a device tensor)
            self.sim.step(global_actions, n_steps)
        else:
            self.step(global_actions, n_steps)
        if not overlap or self.world == 1:
            t = getattr(self, "_gather_t", 0)
            self._gather_t = t + 1
            if hasattr(self.sim, "render_into"):         # stand-in simulators of the CPU tests
                if getattr(self, "_gbuf", None) is None:
                    self._gbuf = self._frame_like()
                self.sim.render_into(self._gbuf)
                return t, gather_batch(self._gbuf, self.world, group, dst)
            self.sim.render()                            # into the library's own frame buffer
            return t, self.gather_frames(dst, group)
        gx = getattr(self, "_gx", None)
        if gx is None:
            rank = dist.get_rank(group)
            gx = self._gx = {"bufs": [self._frame_like() for _ in range(2)], "works": [None, None], "t": 0, "rank": rank,
                             "roots": [[self._frame_like() for _ in range(self.world)] if rank == dst else None for _ in range(2)]}
        t, b = gx["t"], gx["t"] % 2
        if gx["works"][b] is not None:                   # the gather of step t-2 read this buffer: it must be done
            gx["works"][b].wait()
        self._render_into(gx["bufs"][b])
        gx["works"][b] = dist.gather(gx["bufs"][b], gx["roots"][b], dst=dst, group=group, async_op=True)
        gx["t"] = t + 1
        if t == 0:
            return None, None
        return t - 1, self._collect(1 - b, dst)

    def _collect(self, b: int, dst: int):
        import torch
        gx = self._gx
        if gx["works"][b] is not None:
            gx["works"][b].wait()
            gx["works"][b] = None
        if gx["rank"] != dst:
            return None
        return torch.cat(gx["roots"][b], dim=0)

    def flush_gather(self, dst: int = 0):
        """Frames of the last step issued by step_render_gather(overlap=True): (t, frames on `dst` / None)."""
        gx = getattr(self, "_gx", None)
        if gx is None or gx["t"] == 0:
            return None, None
        t = gx["t"] - 1
        out = self._collect(t % 2, dst)
        if hasattr(self.sim, "bind_frames"):
            self.sim.bind_frames(None)                   # back to the library's own buffer
        return t, out
