"""Stand-in for ray.rllib.env.vector_env.VectorEnv (see ../../../README.md)."""
import os

if os.environ.get("PNR_STUB_RAY_STYLE", "old") == "new":
    class VectorEnv:
        def __init__(self, observation_space, action_space, num_envs):
            self.observation_space, self.action_space, self.num_envs = observation_space, action_space, num_envs

        def vector_reset(self):
            raise NotImplementedError

        def reset_at(self, index=None):
            raise NotImplementedError

        def vector_step(self, actions):
            raise NotImplementedError

        def get_sub_environments(self):
            return []
else:
    class VectorEnv:
        def vector_reset(self):
            raise NotImplementedError

        def reset_at(self, index):
            raise NotImplementedError

        def vector_step(self, actions):
            raise NotImplementedError

        def get_unwrapped(self):
            raise NotImplementedError
