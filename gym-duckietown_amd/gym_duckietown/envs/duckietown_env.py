"""DuckietownEnv (reference: src/gym_duckietown/envs/duckietown_env.py:9-72): control with
(velocity, steering) instead of wheel duties.  The inverse kinematics runs inside the HIP step
kernel (DTSIM_ACTION_VEL_STEER); this class only formats `info`."""
import numpy as np

from ..simulator import Simulator, spaces


class DuckietownEnv(Simulator):
    _ACTION_MODE = "vel_steer"

    def __init__(self, gain=1.0, trim=0.0, radius=0.0318, k=27.0, limit=1.0, **kwargs):
        self.gain, self.trim, self.radius, self.k, self.limit = gain, trim, radius, k, limit
        Simulator.__init__(self, gain=gain, trim=trim, radius=radius, k=k, limit=limit, **kwargs)
        self.action_space = spaces.Box(low=np.array([-1, -1]), high=np.array([1, 1]), dtype=np.float32)

    @property
    def unwrapped(self):
        return self

    def step(self, action):
        vel, angle = float(action[0]), float(action[1])
        obs, reward, done, info = Simulator.step(self, np.array([vel, angle], dtype=np.float64))
        baseline = self.wheel_dist
        info["DuckietownEnv"] = {
            "k": self.k, "gain": self.gain, "train": self.trim, "radius": self.radius,
            "omega_r": (vel + 0.5 * angle * baseline) / self.radius,
            "omega_l": (vel - 0.5 * angle * baseline) / self.radius,
        }
        return obs, reward, done, info


class DuckietownLF(DuckietownEnv):
    """Lane following task (duckietown_env.py:75-86): identical to DuckietownEnv."""
