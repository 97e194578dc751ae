"""Adapters above the env boundary, with the reference's names and semantics.

  src/gym_duckietown/wrappers.py : DiscreteWrapper :8, SteeringToWheelVelWrapper :36, PyTorchObsWrapper :92,
                                   ResizeWrapper :111 (cv2.INTER_CUBIC on the axis-swapped image), UndistortWrapper :145
  learning/utils/wrappers.py     : ResizeWrapper (imresize == PIL bilinear) :38, NormalizeWrapper :57,
                                   ImgWrapper :72, DtRewardWrapper :89, ActionWrapper :105

These are host-side, N = 1 conveniences over `gym_duckietown.simulator.Simulator`; the batched,
on-device equivalent of the observation wrappers is `dtsim.BatchedSimulator.observe()` /
`dtsim_observe` (same arithmetic: `dtsim/resample.py`, pinned bit-exact against PIL).

Parity notes: the PIL-bilinear path is pinned (tests/test_observe_host.py).  OpenCV is not available
offline, so `ResizeWrapper`'s INTER_CUBIC and `UndistortWrapper`'s rectify map are restatements of
OpenCV's documented algorithms (bicubic a = -0.75, 11-bit fixed-point taps, replicated border;
`initUndistortRectifyMap` + `remap(INTER_NEAREST)`) -- parity unpinned.
"""
from __future__ import annotations

import numpy as np

from dtsim import resample

try:  # gym is optional
    import gym
    from gym import spaces
    _Base = gym.Wrapper
except Exception:  # pragma: no cover
    gym = None

    class _Base:  # minimal gym.Wrapper
        def __init__(self, env):
            self.env = env
            self.observation_space = getattr(env, "observation_space", None)
            self.action_space = getattr(env, "action_space", None)
            self.reward_range = getattr(env, "reward_range", None)

        @property
        def unwrapped(self):
            return getattr(self.env, "unwrapped", self.env)

        def __getattr__(self, name):
            return getattr(self.env, name)

        def reset(self, **kw):
            return self.env.reset(**kw)

        def step(self, action):
            return self.env.step(action)

        def render(self, *a, **kw):
            return self.env.render(*a, **kw)

        def close(self):
            return self.env.close()

    class _Space:
        def __init__(self, low, high, shape, dtype):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

    class _Discrete:
        def __init__(self, n):
            self.n = n

    class spaces:  # noqa: N801
        Box = _Space
        Discrete = _Discrete


def _box(low, high, shape, dtype):
    if gym is not None:
        return spaces.Box(low, high, shape, dtype=dtype)
    return spaces.Box(low, high, shape, dtype)


class _ObservationWrapper(_Base):
    def reset(self, **kw):
        return self.observation(self.env.reset(**kw))

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        return self.observation(obs), reward, done, info

    def observation(self, obs):
        raise NotImplementedError


class _ActionWrapper(_Base):
    def step(self, action):
        return self.env.step(self.action(action))

    def action(self, action):
        raise NotImplementedError


# ------------------------------------------------------------- src/gym_duckietown/wrappers.py ----
class DiscreteWrapper(_ActionWrapper):
    """left / right / forward instead of continuous control (wrappers.py:8-33)."""

    def __init__(self, env):
        super().__init__(env)
        self.action_space = spaces.Discrete(3)

    def action(self, action):
        table = {0: [0.6, +1.0], 1: [0.6, -1.0], 2: [0.7, 0.0]}
        assert int(action) in table, "unknown action"
        return np.array(table[int(action)])

    def reverse_action(self, action):
        raise NotImplementedError()


class SteeringToWheelVelWrapper(_ActionWrapper):
    """[velocity | heading] -> [wheelvel_left | wheelvel_right] (wrappers.py:36-89)."""

    def __init__(self, env, gain=1.0, trim=0.0, radius=0.0318, k=27.0, limit=1.0):
        super().__init__(env)
        self.gain, self.trim, self.radius, self.k, self.limit = gain, trim, radius, k, limit

    def action(self, action):
        vel, angle = action
        baseline = self.unwrapped.wheel_dist
        k_r_inv = (self.gain + self.trim) / self.k
        k_l_inv = (self.gain - self.trim) / self.k
        omega_r = (vel + 0.5 * angle * baseline) / self.radius
        omega_l = (vel - 0.5 * angle * baseline) / self.radius
        u_r = max(min(omega_r * k_r_inv, self.limit), -self.limit)
        u_l = max(min(omega_l * k_l_inv, self.limit), -self.limit)
        return np.array([u_l, u_r])

    def reverse_action(self, action):
        raise NotImplementedError()


class PyTorchObsWrapper(_ObservationWrapper):
    """observation.transpose(2, 1, 0) (wrappers.py:92-108)."""

    def __init__(self, env=None):
        super().__init__(env)
        h, w, c = self.observation_space.shape
        self.observation_space = _box(0, 255, [c, w, h], self.observation_space.dtype)

    def observation(self, observation):
        return observation.transpose(2, 1, 0)


def resize_cubic(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_CUBIC) for uint8 images: the statement in
    dtsim/resample.py (also what dtsim_observe_cubic computes on the device for a whole frame batch)."""
    return resample.resize_cubic(img, out_h, out_w)


class ResizeWrapper(_ObservationWrapper):
    """wrappers.py:111-142: cv2.resize(obs.swapaxes(0, 2), dsize=(resize_w, resize_h), INTER_CUBIC).swapaxes(0, 2).
    (The reference resizes the axis-swapped image, so the result has shape [resize_w?..] exactly as there:
    input [H,W,3] -> swap -> [3,W,H] treated by OpenCV as a 3-row, W-column, H-channel image.)"""

    def __init__(self, env=None, resize_w=80, resize_h=80):
        super().__init__(env)
        self.resize_h, self.resize_w = resize_h, resize_w
        obs_shape = self.observation_space.shape
        self.observation_space = _box(0, 255, [obs_shape[0], resize_h, resize_w], self.observation_space.dtype)

    def observation(self, observation):
        sw = observation.swapaxes(0, 2)                       # [3, W, H]: rows = 3, cols = W, channels = H
        return resize_cubic(sw, self.resize_h, self.resize_w).swapaxes(0, 2)


class UndistortWrapper(_ObservationWrapper):
    """wrappers.py:145-227: sets env.undistort and remaps with initUndistortRectifyMap(K, D, I, P) / INTER_NEAREST."""

    K = np.array([[305.5718893575089, 0, 303.0797142544728], [0, 308.8338858195428, 231.8845403702499], [0, 0, 1]])
    D = np.array([-0.2, 0.0305, 0.0005859930422629722, -0.0006697840226199427, 0])
    P = np.array([[220.2460277141687, 0, 301.8668918355899, 0], [0, 238.6758484095299, 227.0880056118307, 0], [0, 0, 1, 0]])

    def __init__(self, env=None):
        super().__init__(env)
        assert env.unwrapped.distortion, "Distortion is false, no need for this wrapper"
        self.env.unwrapped.undistort = True
        self.mapx = self.mapy = None

    def _maps(self, H, W):
        fx, fy, cx, cy = self.P[0, 0], self.P[1, 1], self.P[0, 2], self.P[1, 2]
        k1, k2, p1, p2, k3 = self.D
        u, v = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
        x, y = (u - cx) / fx, (v - cy) / fy
        r2 = x * x + y * y
        kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
        xd = x * kr + p1 * 2 * x * y + p2 * (r2 + 2 * x * x)
        yd = y * kr + p1 * (r2 + 2 * y * y) + p2 * 2 * x * y
        return (self.K[0, 0] * xd + self.K[0, 2]).astype(np.float32), (self.K[1, 1] * yd + self.K[1, 2]).astype(np.float32)

    def observation(self, observation):
        H, W = observation.shape[:2]
        if self.mapx is None:
            self.mapx, self.mapy = self._maps(H, W)
        sx = np.rint(self.mapx.astype(np.float64)).astype(np.int64)
        sy = np.rint(self.mapy.astype(np.float64)).astype(np.int64)
        ok = (sx >= 0) & (sx < W) & (sy >= 0) & (sy < H)
        out = np.zeros_like(observation)
        out[ok] = observation[sy[ok], sx[ok]]
        return out


# ----------------------------------------------------------------- learning/utils/wrappers.py ----
class PILResizeWrapper(_ObservationWrapper):
    """learning/utils/wrappers.py:38-54 `ResizeWrapper(shape=(120, 160, 3))`: scipy imresize == PIL bilinear."""

    def __init__(self, env=None, shape=(120, 160, 3)):
        super().__init__(env)
        self.shape = tuple(shape)
        self.observation_space = _box(0, 255, self.shape, self.observation_space.dtype)

    def observation(self, observation):
        return resample.resize_bilinear(observation, self.shape[0], self.shape[1])


class NormalizeWrapper(_ObservationWrapper):
    """(obs - lo) / (hi - lo) -> float32 in [0, 1] (learning/utils/wrappers.py:57-69)."""

    def __init__(self, env=None):
        super().__init__(env)
        self.obs_lo, self.obs_hi = 0.0, 255.0
        self.observation_space = _box(0.0, 1.0, self.observation_space.shape, np.float32)

    def observation(self, obs):
        return (obs.astype(np.float32) - np.float32(self.obs_lo)) / np.float32(self.obs_hi - self.obs_lo)


class ImgWrapper(_ObservationWrapper):
    """HWC -> CHW (learning/utils/wrappers.py:72-86)."""

    def __init__(self, env=None):
        super().__init__(env)
        h, w, c = self.observation_space.shape
        self.observation_space = _box(0, 255, [c, h, w], self.observation_space.dtype)

    def observation(self, observation):
        return observation.transpose(2, 0, 1)


class DtRewardWrapper(_Base):
    """learning/utils/wrappers.py:89-102."""

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        return obs, self.reward(reward), done, info

    def reward(self, reward):
        if reward == -1000:
            return -10
        return reward + 10 if reward > 0 else reward + 4


class ActionWrapper(_ActionWrapper):
    """learning/utils/wrappers.py:105-112: at max speed the duckie can't turn any more."""

    def action(self, action):
        return [action[0] * 0.8, action[1]]
