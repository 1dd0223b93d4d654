"""gymnasium.vector.VectorEnv-shaped front end: what a trainer written against gymnasium 0.29's
vector API (the version the reference pins, setup.py:35) expects from `gymnasium.vector.make(id, num_envs)`:

    obs, infos = envs.reset(seed=..., options=...)
    obs, rewards, terminations, truncations, infos = envs.step(actions)

* sub-environments that terminate are reset in the same call; `obs[i]` is then the first observation of the new
  episode, `infos["final_observation"][i]` the terminal observation and `infos["final_info"][i]` the reference's
  end-of-episode info dict (e.g. mortar_mayhem_grid.py:356-362); `infos["_final_observation"]` / `["_final_info"]`
  are the boolean masks.  `truncations` is all False (the reference never truncates).
* `reset(seed=s)` seeds sub-environment i with `s + i`, like gymnasium's vector envs.

By default everything stays on the GPU as torch tensors (`final_observation` is a full [N, ...] tensor whose rows are
valid where the mask is set; `final_info` a dict of [N] tensors).  With `as_numpy=True` the outputs are converted to
exactly gymnasium's host-side layout (numpy arrays, object arrays of per-env entries / None); that costs a device
synchronisation and a copy of the observations per step and is meant for small `num_envs` and plumbing tests.
"""
import numpy as np
import torch

from .vec_env import VecMemoryGym

try:  # inherit when the host has gymnasium, so isinstance checks of trainers pass
    from gymnasium.vector import VectorEnv as _Base
except Exception:  # gymnasium is not a dependency of the hot path
    _Base = object


class GymnasiumVectorEnv(_Base):
    def __init__(self, env_id, num_envs, device=None, obs_format="u8_xyc", as_numpy=False, on_capacity="raise", capacity=None):
        # as_numpy: gymnasium's host-side layout with the reference's dtypes -- rewards and info["ground_truth"] as float64
        # on_capacity="truncate": a sub-environment whose episode this build ended on one of its capacities (VecMemoryGym) comes back with
        # terminations[i] = False and truncations[i] = True, like a time limit; its final_observation / final_info are set as for any end
        self.env = VecMemoryGym(env_id, num_envs=num_envs, device=device, obs_format=obs_format, final_observation=True,
                                ground_truth64=bool(as_numpy), on_capacity=on_capacity, capacity=capacity)
        self.as_numpy = bool(as_numpy)
        done = False
        if _Base is not object:  # gymnasium 0.29: VectorEnv.__init__(num_envs, observation_space, action_space) batches the spaces
            try:
                _Base.__init__(self, int(num_envs), self.env.observation_space, self.env.action_space)
                done = True
            except TypeError:  # gymnasium 1.x: no constructor arguments, attributes are set by the subclass
                _Base.__init__(self)
        if not done:
            self.num_envs = int(num_envs)
            self.is_vector_env = True
            self.single_observation_space = self.env.observation_space
            self.single_action_space = self.env.action_space
            self.observation_space, self.action_space = self._batched_spaces()
            self.closed = False
        self.metadata = self.env.metadata
        self.spec = None

    def _batched_spaces(self):
        try:
            from gymnasium.vector.utils import batch_space
            return (batch_space(self.single_observation_space, self.num_envs),
                    batch_space(self.single_action_space, self.num_envs))
        except Exception:
            return self.single_observation_space, self.single_action_space

    # ------------------------------------------------------------------ conversions
    def _host(self, x):
        if isinstance(x, dict):
            return {k: self._host(v) for k, v in x.items()}
        return x.cpu().numpy() if self.as_numpy else x

    def reset(self, seed=None, options=None):
        obs, info = self.env.reset(seed=seed, options=options)
        return self._host(obs), self._host(info)

    def step(self, actions):
        obs, reward, done, trunc, info = self.env.step(actions)
        ep = {"reward": info["reward"], "length": info["length"]}
        for nm in self.env.info_names:
            ep[nm] = info[nm]
        infos = {}
        if "ground_truth" in info:
            infos["ground_truth"] = info["ground_truth"]
        term = done if self.env.capacity_u8 is None else (done & ~trunc)  # gymnasium: an episode ends EITHER terminated or truncated
        if "capacity_exceeded" in info:
            infos["capacity_exceeded"] = info["capacity_exceeded"]
        if not self.as_numpy:
            infos.update(final_observation=info["final_observation"], _final_observation=done, final_info=ep, _final_info=done)
            return obs, reward, term, trunc, infos
        d = done.cpu().numpy()
        out = self._host(infos)
        if d.any():
            idx = np.nonzero(d)[0]
            fobs = np.full(self.num_envs, None, dtype=object)
            finfo = np.full(self.num_envs, None, dtype=object)
            rows = info["final_observation"][torch.as_tensor(idx, device=done.device)].cpu().numpy()
            host_ep = {k: v.cpu().numpy() for k, v in ep.items()}
            for j, i in enumerate(idx):
                fobs[i] = rows[j]
                finfo[i] = {k: (int(v[i]) if k == "length" else float(v[i])) for k, v in host_ep.items()}
            out.update(final_observation=fobs, _final_observation=d.copy(), final_info=finfo, _final_info=d.copy())
        return self._host(obs), self.env.reward64.cpu().numpy(), term.cpu().numpy(), trunc.cpu().numpy(), out  # (the reference's Python floats: doubles)

    # gymnasium.vector.VectorEnv API surface used by trainers
    def step_async(self, actions):
        self._pending = actions

    def step_wait(self):
        return self.step(self._pending)

    def reset_async(self, seed=None, options=None):
        self._pending_reset = (seed, options)

    def reset_wait(self, seed=None, options=None):
        s, o = getattr(self, "_pending_reset", (seed, options))
        return self.reset(seed=s, options=o)

    def close(self, **kwargs):
        if not self.closed:
            self.env.close()
            self.closed = True

    def close_extras(self, **kwargs):
        self.env.close()

    @property
    def unwrapped(self):
        return self
