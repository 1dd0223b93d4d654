"""GPU (-m gpu): render("debug_rgb_array") -- mg_render_debug, the ground-truth view of every instance stretched to
336 x 336 -- against the oracle's debug view (itself pinned to the reference's *_gt.gif recordings,
tests/test_oracle_debug_views.py), for all ten env ids, after the reset, after every step and across same-step auto-resets."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [("MortarMayhem-Grid-v0", None, 90), ("MortarMayhem-v0", None, 120), ("Endless-MortarMayhem-v0", None, 140),
         ("MortarMayhemB-Grid-v0", None, 60), ("MortarMayhemB-v0", None, 60),
         ("MysteryPath-v0", dict(max_steps=40), 100), ("MysteryPath-Grid-v0", dict(max_steps=30), 80),
         ("Endless-MysteryPath-v0", None, 120), ("Endless-MysteryPath-v0", dict(show_background=True, show_stamina=True), 80),
         ("SearingSpotlights-v0", dict(agent_health=30, num_coins=[2, 3]), 140), ("Endless-SearingSpotlights-v0", dict(agent_health=40), 160)]


@pytest.mark.parametrize("env_id,options,steps", CASES, ids=["%s-%d" % (c[0], k) for k, c in enumerate(CASES)])
def test_debug_view_equals_oracle(env_id, options, steps):
    import memory_gym_amd
    import oracle_lib

    n = 24
    env = memory_gym_amd.make(env_id, num_envs=n, device=0, render_mode="debug_rgb_array")
    ref = oracle_lib.OracleBatch(env_id, n, options=options)
    seeds = np.arange(n, dtype=np.int64) * 5 + 2
    env.reset(seed=seeds, options=options)
    ref.reset(seeds)
    disc = env.action_dim == 1
    prng = np.random.Generator(np.random.PCG64(7))

    def check(where):
        got = env.render().cpu().numpy()
        assert got.shape == (n, 336, 336, 3)
        for i in range(n):
            want = ref.envs[i].debug_view()
            if not np.array_equal(got[i], want):
                bad = np.argwhere((got[i] != want).any(-1))
                raise AssertionError("%s: debug view of instance %d differs %s in %d px, first (y, x) = %s: hip %s oracle %s" % (
                    env_id, i, where, len(bad), bad[0], got[i][tuple(bad[0])], want[tuple(bad[0])]))

    check("after reset")
    n_done = 0
    for t in range(steps):
        a = (prng.integers(0, 4, n) if disc else prng.integers(0, 3, (n, 2))).astype(np.int32)
        _, _, done, _, _ = env.step(a)
        _, _, d2 = ref.step(a, autoreset=True, want_obs=False)
        assert np.array_equal(done.cpu().numpy(), d2.astype(bool))
        n_done += int(d2.sum())
        if t % 3 == 0 or d2.any():
            check("at step %d" % t)
    assert n_done > 0 or "Endless" in env_id or env_id.startswith("MortarMayhem")
    env.close()
    ref.close()


def test_single_instance_render_modes():
    import memory_gym_amd

    env = memory_gym_amd.envs.GridMortarMayhemEnv(render_mode="debug_rgb_array")
    env.reset(seed=1)
    img = env.render()
    assert img.shape == (336, 336, 3) and img.dtype == np.uint8 and (img == np.array((0, 255, 0))).all(-1).any()  # the target ring
    env.close()
    env = memory_gym_amd.envs.GridMortarMayhemEnv(render_mode="rgb_array")
    obs, _ = env.reset(seed=1)
    assert np.array_equal(env.render(), obs.transpose(1, 0, 2))
    env.close()
