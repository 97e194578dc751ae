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
