"""GPU (-m gpu): mg_render draws the current frames again (into any buffer, without stepping) and the placement probe of the
Python mirror leaves the first observation and the following trajectory untouched."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("env_id,adim,n_act", [("MortarMayhem-Grid-v0", 1, 4), ("Endless-SearingSpotlights-v0", 2, 3), ("Endless-MysteryPath-v0", 1, 4)])
def test_render_reproduces_the_current_frames(env_id, adim, n_act):
    import memory_gym_amd
    import torch
    from memory_gym_amd import _native

    n = 300
    env = memory_gym_amd.make(env_id, num_envs=n, device=0)
    env.reset(seed=4)
    g = torch.Generator(device="cuda").manual_seed(1)
    for t in range(70):
        a = torch.randint(0, n_act, (n,) if adim == 1 else (n, adim), device="cuda", generator=g, dtype=torch.int32)
        obs = env.step(a)[0]
        if t % 10 == 9:
            words = [env.rng_words(i).copy() for i in (0, n - 1)]
            again = torch.full_like(obs, 7)
            _native.check(_native.LIB.mg_render(env._h, again.data_ptr(), env._stream()), "mg_render")
            assert torch.equal(again, obs), "step %d" % t
            assert all((env.rng_words(i) == w).all() for i, w in zip((0, n - 1), words)), "mg_render must not touch the state"
    env.close()


def test_placement_probe_is_transparent():
    import memory_gym_amd
    import torch

    n = 4096  # 86 MB of observations: above the probe's threshold
    a = memory_gym_amd.make("Endless-MortarMayhem-v0", num_envs=n, device=0, tune_placement=True)
    b = memory_gym_amd.make("Endless-MortarMayhem-v0", num_envs=n, device=0, tune_placement=False)
    oa, _ = a.reset(seed=3)
    ob, _ = b.reset(seed=3)
    assert getattr(a, "placement_probe_ms", None) and len(a.placement_probe_ms) >= 2 and not hasattr(b, "placement_probe_ms")
    assert torch.equal(oa, ob)
    g = torch.Generator(device="cuda").manual_seed(1)
    for t in range(40):
        act = torch.randint(0, 3, (n, 2), device="cuda", generator=g, dtype=torch.int32)
        ra, rb = a.step(act), b.step(act)
        assert torch.equal(ra[0], rb[0]) and torch.equal(ra[1], rb[1]) and torch.equal(ra[2], rb[2]), "step %d" % t
    a.close()
    b.close()
