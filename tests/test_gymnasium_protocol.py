"""`gymnasium.make(id)` must hand a trainer a real gymnasium environment (BASELINE.json north_star: "behind the repo's
existing gymnasium.make() ids"; reference: memory_gym/__init__.py:13-61, environment.py:3-7).  gymnasium is not installed
on the build / GPU boxes, so these tests run in a subprocess whose path holds tests/fake_gymnasium -- a stand-in that
restates gymnasium 0.29's register() / make() sequence (entry-point resolution, `env.unwrapped.spec`, passive checker,
order enforcing).  CPU part: registration and class structure; GPU part: make -> reset -> step under the wrappers."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATHS = os.pathsep.join([os.path.join(ROOT, "tests", "fake_gymnasium"), os.path.join(ROOT, "endless-memory-gym_amd"), os.path.join(ROOT, "tests")])

# class names of the reference (memory_gym/__init__.py:2-11) by id (:13-61)
REFERENCE_CLASSES = {
    "SearingSpotlights-v0": "SearingSpotlightsEnv", "Endless-SearingSpotlights-v0": "EndlessSearingSpotlightsEnv",
    "MortarMayhem-v0": "MortarMayhemEnv", "Endless-MortarMayhem-v0": "EndlessMortarMayhemEnv",
    "MortarMayhem-Grid-v0": "GridMortarMayhemEnv", "MortarMayhemB-v0": "MortarMayhemTaskBEnv",
    "MortarMayhemB-Grid-v0": "GridMortarMayhemTaskBEnv", "MysteryPath-v0": "MysteryPathEnv",
    "Endless-MysteryPath-v0": "EndlessMysteryPathEnv", "MysteryPath-Grid-v0": "GridMysteryPathEnv"}


def run(code):
    env = dict(os.environ, PYTHONPATH=PATHS)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    return out.stdout


def test_registration_mirrors_the_reference():
    code = """
import importlib, json
import gymnasium as gym
import memory_gym_amd  # registers as a side effect of the import, like `import memory_gym`
out = {}
for env_id, spec in gym.registry.items():
    mod, attr = spec.entry_point.split(":")
    cls = getattr(importlib.import_module(mod), attr)
    out[env_id] = [attr, issubclass(cls, gym.Env), cls.metadata.get("render_fps"), spec.order_enforce, spec.disable_env_checker, spec.max_episode_steps]
print(json.dumps(out))
"""
    import json
    got = json.loads(run(code).strip().splitlines()[-1])
    assert set(got) == set(REFERENCE_CLASSES)
    for env_id, (attr, is_env, fps, order, no_check, limit) in got.items():
        assert attr == REFERENCE_CLASSES[env_id] and is_env and fps == 25
        assert order is True and no_check is False and limit is None  # gymnasium's defaults, as in the reference's register() calls


@pytest.mark.gpu
def test_make_reset_step_under_gymnasiums_wrappers():
    code = """
import numpy as np
import gymnasium as gym
import memory_gym_amd
import oracle_lib
from memory_gym_amd import envs
for env_id in %r:
    env = gym.make(env_id)
    assert isinstance(env.unwrapped, gym.Env) and type(env.unwrapped).__name__ == %r[env_id]
    assert env.unwrapped.spec.id == env_id and env.spec.id == env_id
    try:
        env.step(env.action_space)  # order enforcing: step before reset
        raise SystemExit("step before reset did not raise")
    except RuntimeError:
        pass
    ref = oracle_lib.OracleEnv(env_id)
    obs, info = env.reset(seed=1)
    want = ref.reset(1)
    vis = obs["visual_observation"] if isinstance(obs, dict) else obs
    assert vis.dtype == np.uint8 and np.array_equal(vis, want), env_id
    assert isinstance(env.unwrapped.np_random, np.random.Generator)
    g = np.random.Generator(np.random.PCG64(5))
    disc = ref.discrete
    for t in range(60):
        a = int(g.integers(0, 4)) if disc else g.integers(0, 3, 2)
        obs, r, term, trunc, info = env.step(a)
        o2, r2, d2 = ref.step(a)
        vis = obs["visual_observation"] if isinstance(obs, dict) else obs
        assert np.array_equal(vis, o2) and term == d2 and trunc is False
        assert type(r) is float and r == r2, (env_id, t, r, r2)  # the reference's Python float, not its float32 rounding
        if term:
            obs, info = env.reset()
            ref.reset(None)
    env.close()
envs_v = memory_gym_amd.GymnasiumVectorEnv("MortarMayhem-Grid-v0", 4, as_numpy=True)
assert isinstance(envs_v, gym.vector.VectorEnv) and envs_v.num_envs == 4 and envs_v.unwrapped is envs_v
assert envs_v.observation_space.shape == (4, 84, 84, 3) and envs_v.single_action_space.n == 4
obs, info = envs_v.reset(seed=3)
assert envs_v.observation_space.contains(obs)
obs, rew, term, trunc, info = envs_v.step(np.zeros(4, np.int64))
assert obs.shape == (4, 84, 84, 3) and rew.dtype == np.float64
envs_v.close()
print("ok")
""" % (sorted(REFERENCE_CLASSES), REFERENCE_CLASSES)
    assert run(code).strip().endswith("ok")
