"""Drop-in for the reference package `gym_duckietown` (v6.1.34 surface of the hot path):
`Simulator`, `DuckietownEnv`, `MultiMapEnv`, the exception names and the gym ids
`Duckietown-<map>-v0` / `MultiMap-v0` (src/gym_duckietown/__init__.py:30-50) -- all backed
by the HIP library through dtsim.BatchedSimulator (N=1 views).  No OpenGL, no pyglet.
"""
__version__ = "6.1.34+dtsim"

import logging

logger = logging.getLogger("gym-duckietown")

from .exceptions import GymDuckietownException, InvalidMapException, NotInLane  # noqa: E402,F401


def _register():
    try:
        from gym.envs.registration import register
    except Exception:       # gym is optional: the classes work without it
        return
    from dtsim import assets
    for name in assets.MAPS:
        try:
            register(id=f"Duckietown-{name}-v0", entry_point="gym_duckietown.envs:DuckietownEnv",
                     reward_threshold=400.0, kwargs={"map_name": name})
        except Exception:
            pass
    try:
        register(id="MultiMap-v0", entry_point="gym_duckietown.envs:MultiMapEnv", reward_threshold=400.0)
    except Exception:
        pass


_register()
