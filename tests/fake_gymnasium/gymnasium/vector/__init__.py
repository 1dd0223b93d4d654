"""TEST INFRASTRUCTURE (see gymnasium/__init__.py of this stand-in): VectorEnv's constructor contract of gymnasium 0.29."""
from . import utils  # noqa: F401


class VectorEnv:
    def __init__(self, num_envs, observation_space, action_space):
        self.num_envs = num_envs
        self.is_vector_env = True
        self.observation_space = utils.batch_space(observation_space, n=num_envs)
        self.action_space = utils.batch_space(action_space, n=num_envs)
        self.closed = False
        self.viewer = None
        self.single_observation_space = observation_space
        self.single_action_space = action_space

    @property
    def unwrapped(self):
        return self

    def close_extras(self, **kwargs):
        pass

    def close(self, **kwargs):
        if self.closed:
            return
        self.close_extras(**kwargs)
        self.closed = True
