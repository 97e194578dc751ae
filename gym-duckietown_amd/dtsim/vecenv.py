"""A vectorised, device-resident env for learners: the gym.Env contract of the reference
(`reset() -> obs`, `step(a) -> obs, reward, done, info`), batched over N envs, with every tensor living in HBM.

    env = DuckietownVecEnv("small_loop", num_envs=4096, obs_shape=(120, 160))
    obs = env.reset()                                  # torch.float32 [N, 3, 120, 160] on the GPU
    obs, reward, done, info = env.step(actions)        # actions: torch / numpy [N, 2]

One step is: `dtsim_step` (kinematics, collisions, reward / done) -> reward and done are copied out ->
`dtsim_reset_done` restarts the finished episodes with the device-side sampler (the reference's reset
distributions and spawn test) -> `dtsim_render` -> `dtsim_observe` (the learners' ResizeWrapper /
ImgWrapper / NormalizeWrapper, bit-identical to PIL).  So, as with gym's vector envs, the observation returned
with `done[e] = True` is already the first observation of env e's next episode.  Everything is launched on one
HIP stream shared with torch and ordered against the caller's stream on the GPU: no host synchronisation in the loop.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from . import _ffi
from .batched import BatchedSimulator


class DuckietownVecEnv:
    def __init__(self, map_name="small_loop", num_envs: int = 1024, obs_shape: Optional[Tuple[int, int]] = (120, 160),
                 chw: bool = True, normalize: bool = True, action_mode: str = "vel_steer", device: int = 0,
                 seed: Optional[int] = 0, **sim_kwargs):
        import torch
        self.torch = torch
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.stream = torch.cuda.Stream(device=self.device)      # the simulator's launches and our copies share it
        self.sim = BatchedSimulator(map_name, num_envs, action_mode=action_mode, device=device, seed=seed,
                                    device_reset=True, auto_reset=False, stream=self.stream.cuda_stream, do_reset=False,
                                    **sim_kwargs)
        self.num_envs = num_envs
        self.obs_shape, self.chw, self.normalize = obs_shape, chw, normalize
        sim = self.sim
        self._reward = torch.as_tensor(sim.field_device(_ffi.FIELD_REWARD), device=self.device)
        self._done = torch.as_tensor(sim.field_device(_ffi.FIELD_DONE), device=self.device)
        self._code = torch.as_tensor(sim.field_device(_ffi.FIELD_DONE_CODE), device=self.device)
        self._steps = torch.as_tensor(sim.field_device(_ffi.FIELD_STEP_COUNT), device=self.device)
        self._frames = torch.as_tensor(sim.frames_device(), device=self.device)
        self.action_shape = (num_envs, 2)

    # ------------------------------------------------------------------------------------------------
    def _observe(self):
        self.sim.render()
        if self.obs_shape is None:
            return self._frames                                     # [N, H, W, 3] uint8, the raw camera frames
        return self.torch.as_tensor(self.sim.observe(self.obs_shape[0], self.obs_shape[1], chw=self.chw,
                                                     normalize=self.normalize), device=self.device)

    def _enter(self):
        self.stream.wait_stream(self.torch.cuda.current_stream(self.device))     # e.g. the policy that produced the actions

    def _leave(self):
        self.torch.cuda.current_stream(self.device).wait_stream(self.stream)     # consumers run after our launches (GPU-side)

    def reset(self):
        self._enter()
        with self.torch.cuda.stream(self.stream):
            self.sim.reset()                                        # device sampler (dtsim_reset(states = NULL))
            obs = self._observe()
        self._leave()
        return obs

    def step(self, actions):
        t = self.torch
        self._enter()
        with t.cuda.stream(self.stream):
            if isinstance(actions, np.ndarray):
                self.sim.step(np.ascontiguousarray(actions, np.float32).reshape(1, self.num_envs, 2))
            else:
                a = actions.to(device=self.device, dtype=t.float32).contiguous()
                a.record_stream(self.stream)
                self.sim.step(a)
            reward = self._reward.to(t.float32)                     # copies: the reset below clears the fields
            done = self._done.to(t.bool)
            info = {"done_code": self._code.clone(), "episode_steps": self._steps.clone()}
            self.sim.reset_done()
            obs = self._observe()
        self._leave()
        return obs, reward, done, info

    def close(self):
        self.sim.close()
