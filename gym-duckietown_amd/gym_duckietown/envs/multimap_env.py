"""MultiMapEnv (reference: src/gym_duckietown/envs/multimap_env.py:8-91): round-robin over the
two `*_only_duckies` maps on every reset; first reset selects index 1."""
from .duckietown_env import DuckietownEnv

try:
    import gym
    _Base = gym.Env
except Exception:  # pragma: no cover
    _Base = object


class MultiMapEnv(_Base):
    def __init__(self, **kwargs):
        self.env_list = []
        self.window = None
        for map_name in ["loop_only_duckies", "small_loop_only_duckies"]:
            env = DuckietownEnv(map_name=map_name, **kwargs)
            self.action_space, self.observation_space = env.action_space, env.observation_space
            self.reward_range = env.reward_range
            self.env_list.append(env)
        self.cur_env_idx = 0
        self.cur_reward_sum = 0
        self.cur_num_steps = 0

    def seed(self, seed=None):
        for env in self.env_list:
            env.seed(seed)
        return [seed]

    def reset(self):
        self.cur_env_idx = (self.cur_env_idx + 1) % len(self.env_list)
        return self.env_list[self.cur_env_idx].reset()

    def step(self, action):
        obs, reward, done, info = self.env_list[self.cur_env_idx].step(action)
        self.cur_reward_sum += reward
        self.cur_num_steps += 1
        if done:
            self.cur_reward_sum = 0
            self.cur_num_steps = 0
        return obs, reward, done, info

    def render(self, mode="human", close=False):
        return self.env_list[self.cur_env_idx].render(mode, close)

    def close(self):
        for env in self.env_list:
            env.close()
        self.cur_env_idx = 0
        self.env_list = None

    @property
    def step_count(self):
        return self.env_list[self.cur_env_idx].step_count
