"""TEST INFRASTRUCTURE (see gymnasium/__init__.py of this stand-in): register() / make() / spec() as gymnasium 0.29 runs them."""
import copy
import importlib
import inspect
from dataclasses import dataclass, field

import numpy as np

from .. import spaces
from ..core import Wrapper

registry = {}


@dataclass
class EnvSpec:
    id: str
    entry_point: object = None
    reward_threshold: float = None
    nondeterministic: bool = False
    max_episode_steps: int = None
    order_enforce: bool = True
    autoreset: bool = False
    disable_env_checker: bool = False
    apply_api_compatibility: bool = False
    kwargs: dict = field(default_factory=dict)


def register(id, entry_point=None, **kw):
    if id in registry:
        raise ValueError("id %s is registered already" % id)
    registry[id] = EnvSpec(id=id, entry_point=entry_point, **kw)


def spec(id):
    return registry[id]


class PassiveEnvChecker(Wrapper):
    """The checks gymnasium's passive checker applies to the first reset and step (spaces, signature, return types)."""

    def __init__(self, env):
        Wrapper.__init__(self, env)
        assert isinstance(env.action_space, spaces.Space), "action_space must be a gymnasium space"
        assert isinstance(env.observation_space, spaces.Space), "observation_space must be a gymnasium space"
        sig = inspect.signature(env.reset)
        assert "seed" in sig.parameters and "options" in sig.parameters, "reset() must accept seed and options"
        self._checked_reset = self._checked_step = False

    def reset(self, *, seed=None, options=None):
        out = self.env.reset(seed=seed, options=options)
        if not self._checked_reset:
            self._checked_reset = True
            assert isinstance(out, tuple) and len(out) == 2, "reset() must return (obs, info)"
            assert self.env.observation_space.contains(out[0]), "the reset observation is not in observation_space"
            assert isinstance(out[1], dict)
        return out

    def step(self, action):
        if not self._checked_step:
            assert self.env.action_space.contains(action), "action %r is not in action_space" % (action,)
        out = self.env.step(action)
        if not self._checked_step:
            self._checked_step = True
            assert isinstance(out, tuple) and len(out) == 5
            obs, reward, terminated, truncated, info = out
            assert self.env.observation_space.contains(obs), "the step observation is not in observation_space"
            assert isinstance(reward, (int, float, np.integer, np.floating)) and not np.isnan(reward)
            assert isinstance(terminated, (bool, np.bool_)) and isinstance(truncated, (bool, np.bool_))
            assert isinstance(info, dict)
        return out


class OrderEnforcing(Wrapper):
    def __init__(self, env):
        Wrapper.__init__(self, env)
        self._has_reset = False

    def reset(self, *, seed=None, options=None):
        self._has_reset = True
        return self.env.reset(seed=seed, options=options)

    def step(self, action):
        if not self._has_reset:
            raise RuntimeError("Cannot call env.step() before calling env.reset()")
        return self.env.step(action)


class TimeLimit(Wrapper):
    def __init__(self, env, max_episode_steps):
        Wrapper.__init__(self, env)
        self._max, self._t = max_episode_steps, 0

    def reset(self, *, seed=None, options=None):
        self._t = 0
        return self.env.reset(seed=seed, options=options)

    def step(self, action):
        obs, r, term, trunc, info = self.env.step(action)
        self._t += 1
        return obs, r, term, trunc or self._t >= self._max, info


def make(id, max_episode_steps=None, disable_env_checker=None, **kwargs):
    spec_ = copy.deepcopy(registry[id])
    kw = dict(spec_.kwargs, **kwargs)
    creator = spec_.entry_point
    if isinstance(creator, str):
        mod, attr = creator.split(":")
        creator = getattr(importlib.import_module(mod), attr)
    render_mode = kw.get("render_mode")
    if render_mode is not None:
        modes = getattr(creator, "metadata", {}).get("render_modes", [])
        assert render_mode in modes, "render_mode %r is not in the environment's metadata (%s)" % (render_mode, modes)
    env = creator(**kw)
    spec_.kwargs = kw
    env.unwrapped.spec = spec_
    if disable_env_checker is False or (disable_env_checker is None and not spec_.disable_env_checker):
        env = PassiveEnvChecker(env)
    if spec_.order_enforce:
        env = OrderEnforcing(env)
    steps = max_episode_steps if max_episode_steps is not None else spec_.max_episode_steps
    if steps is not None:
        env = TimeLimit(env, steps)
    return env
