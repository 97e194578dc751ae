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
    parts = None
    out = None
    if dist.get_rank() == dst:                           # dst is a GLOBAL rank (torch.distributed.gather)
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
        """The frame batches of every rank: [world*n, H, W, 3] on every rank (dst None) or on `dst` only.
        On RCCL the all-gather is the library's own (`dtsim_allgather_frames`, include/dtsim.h): enqueued on the simulator's stream
        right behind the render pass, no host synchronisation in between; torch's current stream is ordered behind it by an event.
        Elsewhere (gloo; gather-to-root) it goes through torch.distributed."""
        import torch
        import torch.distributed as dist
        if dst is None and self._rccl_ready(group):
            n = self.hi - self.lo
            out = torch.empty((self.world * n, self.sim.camera_height, self.sim.camera_width, 3), dtype=torch.uint8,
                              device=f"cuda:{self.sim.device_index}")
            self._lib_wait_torch(out.device)            # `out` is ready on torch's stream: order the library's stream behind it
            self.sim.allgather_frames(self._rccl_comm(group), out)
            self._torch_wait_lib(out.device)
            return out
        if self.world > 1 and dist.get_backend(group) == "gloo" and not hasattr(self.sim, "frames_tensor"):
            return gather_batch(self.local_frames().cpu(), self.world, group, dst)    # gloo moves host memory
        return gather_batch(self.local_frames(), self.world, group, dst)

    # ---- RCCL through the C-ABI (dtsim_allgather_frames) ---------------------------------------------------------
    def _rccl_ready(self, group) -> bool:
        import torch.distributed as dist
        if hasattr(self.sim, "frames_tensor") or not hasattr(self.sim, "allgather_frames"):
            return False                                 # stand-in simulators of the CPU tests
        if self.world == 1:
            return bool(getattr(self, "force_collective", False))
        return dist.is_initialized() and dist.get_backend(group) == "nccl"

    def _rccl_comm(self, group) -> int:
        """An ncclComm_t over the ranks of `group` for the library's all-gather, made once per group: rank 0 of the group draws the
        unique id (librccl through ctypes), torch.distributed carries it to the others, every rank joins with ITS group rank."""
        import ctypes as C
        import torch.distributed as dist
        comms = self.__dict__.setdefault("_comms", {})
        if group in comms:
            return comms[group][0]
        import torch
        rccl = C.CDLL(_find_rccl())

        class UniqueId(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]

        uid = UniqueId()
        grank = dist.get_rank(group) if self.world > 1 else 0
        if grank == 0 and rccl.ncclGetUniqueId(C.byref(uid)) != 0:
            raise RuntimeError("ncclGetUniqueId failed")
        if self.world > 1:
            box = [C.string_at(C.byref(uid), 128)]
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(box, src=src, group=group)
            C.memmove(C.byref(uid), box[0], 128)
        comm = C.c_void_p()
        rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
        torch.cuda.set_device(self.sim.device_index)
        if rccl.ncclCommInitRank(C.byref(comm), self.world, uid, grank) != 0:
            raise RuntimeError("ncclCommInitRank failed")
        comms[group] = (comm.value, rccl)
        return comm.value

    def _lib_stream(self, device):
        """The simulator's HIP stream as a torch stream (dtsim_stream), or None (stand-ins, old torch)."""
        import torch
        if not hasattr(self.sim, "stream") or not getattr(device, "type", None) == "cuda":
            return None
        ls = self.__dict__.get("_ls")
        if ls is None:
            try:
                ls = self._ls = torch.cuda.ExternalStream(self.sim.stream(), device=device)
            except Exception:
                ls = self._ls = False
        return ls or None

    def _torch_wait_lib(self, device):
        """Order torch's current stream behind everything enqueued on the simulator's stream so far (no host wait)."""
        import torch
        ls = self._lib_stream(device)
        if ls is None:
            if hasattr(self.sim, "sync"):
                self.sim.sync()
            return
        ev = torch.cuda.Event()
        ev.record(ls)
        torch.cuda.current_stream(device).wait_event(ev)

    def _lib_wait_torch(self, device):
        """Order the simulator's stream behind torch's current stream (where Work.wait() put the end of a transfer)."""
        import torch
        ls = self._lib_stream(device)
        if ls is None:
            if getattr(device, "type", None) == "cuda":
                torch.cuda.current_stream(device).synchronize()
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        ls.wait_event(ev)

    # ---- the learner's exchange, overlapped with the simulation ---------------------------------------
    # Layout (round 4): per buffer slot ONE preallocated [world*n, ...] tensor on the root; the root's own envs are rendered
    # (or observed) straight into its slice of it, every other rank's batch is received into ITS slice (point-to-point,
    # one transfer per xGMI link into the root) -- no torch.cat, no self-copy, and what the learner gets IS that tensor.
    # Round 5: THREE slots rotate (a batch handed out survives the next call: a learner may keep obs_t next to obs_t+1), and
    # the ordering between the simulator's stream and the exchange is by EVENTS, not host waits: the simulator can run ahead of
    # the host while a transfer is in flight.  On gloo (no device transport) the payload is staged through pinned host memory.
    N_SLOTS = 3                                          # default number of rotating slots (step_render_gather(slots=2) for two)
    ROOT_MEMORY_FRACTION = 0.8                           # of the root device's FREE memory the receive tensors of all slots may take
    DEFAULT_OBS = (120, 160)                             # what = "observe" without obs=: the 160 x 120 observation of the reference's wrappers

    @staticmethod
    def exchange_root_bytes(slots: int, world: int, shape, itemsize: int) -> int:
        """Bytes of receive tensors the ROOT of the overlapped exchange allocates: slots x world x this rank's payload.  At the north
        star's size (8 ranks x 4096 envs x 640 x 480 x 3 B) that is 30.2 GB per slot -- 90.6 GB for three slots of what="frames";
        what="observe" at 160 x 120 is 16 x smaller."""
        n = 1
        for d in shape:
            n *= int(d)
        return int(slots) * int(world) * n * int(itemsize)

    def _produce(self, slot, what: str, obs):
        """This rank's payload of the current step into the slot's send buffer.  Device path: enqueued on the simulator's stream;
        torch's current stream (from which the exchange is issued) is ordered behind it by an event.  what = "frames": the render
        pass writes there directly (dtsim_bind_frames); "observe": render into the library's buffer, dtsim_observe into the slot."""
        buf = slot["dev"]
        if what == "frames":
            if hasattr(self.sim, "render_into"):         # stand-in simulators of the CPU tests
                self.sim.render_into(buf)
                return
            self.sim.bind_frames(buf.data_ptr())
            self.sim.render()
        elif hasattr(self.sim, "observe_into"):          # stand-in simulators of the CPU tests
            self.sim.observe_into(buf, *obs[:2])
            return
        else:
            self.sim.render()
            self.sim.observe(obs[0], obs[1], out=buf, **(obs[2] if len(obs) > 2 else {}))
        if slot["staged"]:                               # gloo: device -> pinned host memory, then the host exchanges it
            self._torch_wait_lib(buf.device)
            slot["send"].copy_(buf, non_blocking=True)
            import torch
            torch.cuda.current_stream(buf.device).synchronize()
        else:
            self._torch_wait_lib(buf.device)

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

    def _exchange_state(self, what: str, obs, dst: int, group, n_slots: int = None):
        """Buffers of the overlapped exchange, made once per (payload, root, group, slots).  `dst` is a GLOBAL rank (as in
        torch.distributed.gather); slices of the root's tensor are in GROUP-rank order.  The root's memory need is checked BEFORE
        anything is allocated (exchange_root_bytes against ROOT_MEMORY_FRACTION of the device's free memory): MemoryError with the numbers."""
        import torch
        import torch.distributed as dist
        n_slots = self.N_SLOTS if n_slots is None else int(n_slots)
        if n_slots not in (2, 3):
            raise ValueError("slots = 2 (a handed-out batch is valid until the next call) or 3 (through the next call)")
        okw = tuple(sorted((obs[2] if obs and len(obs) > 2 else {}).items()))
        key = (what, tuple(obs[:2]) if obs else None, okw, dst, group, n_slots)
        gx = getattr(self, "_gx", None)
        if gx is not None and gx["key"] == key:
            return gx
        if gx is not None:                               # another payload / root / group: let the old transfers finish first
            for sl in gx["slots"]:
                self._wait(sl, reuse=True)
        shape, dtype, dev = self._payload_shape(what, obs)
        multi = self.world > 1
        grank = dist.get_rank(group) if multi else 0     # position of this rank's slice
        is_root = (dist.get_rank() if multi else 0) == dst
        peers = [dist.get_global_rank(group, r) if (multi and group is not None) else r for r in range(self.world)]
        staged = bool(multi and dev.type == "cuda" and dist.get_backend(group) == "gloo")
        xdev = torch.device("cpu") if staged else dev
        n = shape[0]
        if is_root and xdev.type == "cuda":
            need = self.exchange_root_bytes(n_slots, self.world, shape, torch.empty((), dtype=dtype).element_size())
            free, _total = torch.cuda.mem_get_info(xdev)
            if need > self.ROOT_MEMORY_FRACTION * free:
                raise MemoryError(f"the overlapped gather of what={what!r} needs {need / 2**30:.1f} GiB of receive tensors on the root ({n_slots} slots x "
                                  f"{self.world} ranks x {need / n_slots / self.world / 2**20:.0f} MiB) but the device has {free / 2**30:.1f} GiB free: use "
                                  "slots=2, what='observe', or fewer envs per rank")
        slots = []
        for _ in range(n_slots):
            if is_root:
                recv = torch.empty((self.world * n,) + tuple(shape[1:]), dtype=dtype, device=xdev, pin_memory=staged)
                send = recv[grank * n:(grank + 1) * n]     # the root produces in place
            else:
                recv, send = None, torch.empty(shape, dtype=dtype, device=xdev, pin_memory=staged)
            devbuf = torch.empty(shape, dtype=dtype, device=dev) if staged else send
            slots.append({"recv": recv, "send": send, "dev": devbuf, "staged": staged, "works": None})
        gx = self._gx = {"key": key, "slots": slots, "t": 0, "grank": grank, "is_root": is_root, "peers": peers, "n": n, "what": what}
        return gx

    def _start_exchange(self, slot, gx, dst: int, group):
        """Asynchronous gather-to-root of slot's payload: receives into the slices of the root's tensor, one send per other rank."""
        import torch.distributed as dist
        slot["waited"] = False
        if self.world == 1:
            slot["works"] = []
            return
        n = gx["n"]
        if gx["is_root"]:
            ops = [dist.P2POp(dist.irecv, slot["recv"][r * n:(r + 1) * n], peer, group)
                   for r, peer in enumerate(gx["peers"]) if peer != dst]
        else:
            ops = [dist.P2POp(dist.isend, slot["send"], dst, group)]
        slot["works"] = list(dist.batch_isend_irecv(ops))

    def _wait(self, slot, reuse: bool):
        """Wait for the exchange that uses `slot`.  Work.wait() orders torch's CURRENT STREAM behind the transfer on RCCL (and blocks
        the host on gloo): enough for handing the tensor to torch consumers (reuse=False).  Before the slot is produced into again
        (reuse=True) the SIMULATOR's stream, which is not torch's, is ordered behind that point by an event."""
        works = slot["works"]
        if works is None:
            return
        if not slot.get("waited"):                       # (a second wait() on a gloo receive blocks for ever)
            for w in works:
                w.wait()
            slot["waited"] = True
        if reuse:
            t = slot["dev"]
            if t.is_cuda and not slot["staged"]:
                self._lib_wait_torch(t.device)
            slot["works"] = None                         # only now: nothing reads or writes the slot's buffers any more

    def step_render_gather(self, global_actions, n_steps: int = 1, *, overlap: bool = True, dst: int = 0, group=None,
                           local_actions: bool = False, what: str = "observe", obs=None, copy: bool = False, slots: int = None):
        """One learner iteration: step this rank's envs, render them, and gather the batch to (global) rank `dst`.

        what="observe" (the default), obs=(h, w[, observe() keywords]) (default DEFAULT_OBS = 120 x 160): the dtsim_observe output of the step --
        57.6 KB per env at 160 x 120: the exchange a learner can keep up with over xGMI;
        what="frames": the full [n, H, W, 3] uint8 frame batch (3.77 GB per rank at the BASELINE size: xGMI-link bound; and
        slots x world x 3.77 GB of receive tensors on the root -- checked against the free device memory before allocating).
        slots=2 | 3 (default N_SLOTS = 3): rotating receive buffers on the root, see "Lifetime" below.
        overlap=False: blocking; returns (t, batch) -- batch = [world*n, ...] of THIS step on `dst`, None elsewhere.
        overlap=True: the gather of step t runs while step t+1 is simulated and rendered; the call returns the batch
        of step t-1 (the learner runs one step behind the simulator; `flush_gather()` hands out the last one), or (None, None) on
        the first call.
        Lifetime of the returned tensor: it is one of the rotating slots' preallocated receive tensor itself (no copy).  With three slots it
        stays valid through the NEXT call and is overwritten by the one after: a learner can hold obs_t while it receives obs_t+1, no
        longer; with slots=2 (a third less memory on the root) it is valid only until the next call -- pass copy=True (or clone it) for a
        replay buffer.  A slot is produced into again only after the simulator's stream
        has been ordered behind the end of the transfer that read it; consumers must run on torch's current stream (or synchronise
        with it) -- the ordering of a handed-out batch is stream-level, not host-level."""
        if what not in ("frames", "observe"):
            raise ValueError("what = 'observe' (with obs=(height, width[, keywords]); default 120 x 160) or 'frames'")
        if what == "observe" and not obs:
            obs = self.DEFAULT_OBS
        if local_actions:                                # already this rank's slice (e.g. a device tensor)
            self.sim.step(global_actions, n_steps)
        else:
            self.step(global_actions, n_steps)
        gx = self._exchange_state(what, obs, dst, group, slots)
        S = len(gx["slots"])
        t = gx["t"]
        slot = gx["slots"][t % S]
        self._wait(slot, reuse=True)                     # the transfer of step t-3 read / wrote this slot
        self._produce(slot, what, obs)
        self._start_exchange(slot, gx, dst, group)
        gx["t"] = t + 1
        if not overlap:
            self._wait(slot, reuse=False)
            out = slot["recv"]
            return t, (out.clone() if (copy and out is not None) else out)
        if t == 0:
            return None, None
        prev = gx["slots"][(t - 1) % S]
        self._wait(prev, reuse=False)
        out = prev["recv"]
        return t - 1, (out.clone() if (copy and out is not None) else out)

    def flush_gather(self, dst: int = 0):
        """Batch of the last step issued by step_render_gather: (t, batch on `dst` / None); every transfer has ended on return."""
        gx = getattr(self, "_gx", None)
        if gx is None or gx["t"] == 0:
            return None, None
        t = gx["t"] - 1
        last = gx["slots"][t % len(gx["slots"])]
        self._wait(last, reuse=False)
        for sl in gx["slots"]:
            self._wait(sl, reuse=True)
        dev = last["dev"].device
        if getattr(dev, "type", None) == "cuda":
            import torch
            torch.cuda.current_stream(dev).synchronize()
        if gx["what"] == "frames" and hasattr(self.sim, "bind_frames"):
            if hasattr(self.sim, "sync"):
                self.sim.sync()
            self.sim.bind_frames(None)                   # back to the library's own buffer
        return t, last["recv"]


def _find_rccl() -> str:
    """librccl as torch loaded it (so that one RCCL serves both), else the system's."""
    import os
    try:
        import torch
        p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if os.path.exists(p):
            return p
    except Exception:
        pass
    return "librccl.so.1"
