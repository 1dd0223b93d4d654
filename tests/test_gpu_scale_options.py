"""GPU parity (-m gpu): the *_scale reset options over the range the reference accepts -- free floats
(character_controller.py:21-75, pygame_assets.py:133-220; README tables) -- HIP vs the oracle, observation and debug view.

Stamps larger than what a composer keeps in registers per layer (256 padded pixels for the spotlight family's agent / coin /
exit, 1,024 for the Mystery Path agent) used to be refused; the spotlight composers now read the excess from the atlas while
they compose and a Mystery Path handle switches to MysteryBigComposer (csrc/mg_mystery.hip)."""
import numpy as np
import pytest

from gpu_parity import run_parity
from test_gpu_spot import coin_seeker

pytestmark = pytest.mark.gpu

SPOT = [("SearingSpotlights-v0", dict(agent_scale=0.6)),
        ("SearingSpotlights-v0", dict(exit_scale=1.0, coin_scale=0.75, agent_scale=0.5, num_coins=[2])),
        ("SearingSpotlights-v0", dict(exit_scale=1.3, agent_scale=0.125, coin_scale=0.2, exit_visible=True, coins_visible=True)),
        ("SearingSpotlights-v0", dict(agent_scale=0.45, agent_visible=True, black_background=True, coin_scale=0.9)),
        ("Endless-SearingSpotlights-v0", dict(agent_scale=0.5, coin_scale=0.75)),
        ("Endless-SearingSpotlights-v0", dict(agent_scale=0.8, coin_scale=1.2, coins_visible=True, agent_health=60))]


@pytest.mark.parametrize("env_id,opts", SPOT, ids=["%s-%d" % (c[0], k) for k, c in enumerate(SPOT)])
def test_spotlight_family_scales(env_id, opts):
    run_parity(env_id, opts, n=48, steps=150, policy=coin_seeker, n_policy=24)


MYSTERY = [("MysteryPath-v0", dict(agent_scale=0.5, max_steps=80)), ("MysteryPath-v0", dict(agent_scale=0.29, max_steps=60)),
           ("MysteryPath-v0", dict(agent_scale=0.125, max_steps=60)), ("MysteryPath-Grid-v0", dict(agent_scale=0.6, max_steps=40)),
           ("Endless-MysteryPath-v0", dict(agent_scale=0.5, show_stamina=True)),
           ("Endless-MysteryPath-v0", dict(agent_scale=0.4, show_background=True, max_steps=90))]


@pytest.mark.parametrize("env_id,opts", MYSTERY, ids=["%s-%d" % (c[0], k) for k, c in enumerate(MYSTERY)])
def test_mystery_family_scales(env_id, opts):
    assert run_parity(env_id, opts, n=96, steps=140) > 0


MORTAR = [("MortarMayhem-v0", dict(agent_scale=0.5)), ("Endless-MortarMayhem-v0", dict(agent_scale=0.6)),
          ("MortarMayhem-Grid-v0", dict(agent_scale=0.5)), ("MortarMayhemB-v0", dict(agent_scale=0.4))]


@pytest.mark.parametrize("env_id,opts", MORTAR, ids=[c[0] for c in MORTAR])
def test_mortar_family_scales(env_id, opts):
    run_parity(env_id, opts, n=64, steps=130)


def test_scale_change_between_resets_and_back():
    """A handle whose sprites grow past the register form and shrink again: the composer (and, for Mystery Path, the launch
    arrangement) follows the options of the latest full reset."""
    for env_id in ("MysteryPath-v0", "Endless-MysteryPath-v0", "SearingSpotlights-v0"):
        import memory_gym_amd
        import oracle_lib

        n = 40
        env = memory_gym_amd.make(env_id, num_envs=n, device=0)
        prng = np.random.Generator(np.random.PCG64(3))
        for phase, scale in enumerate((0.25, 0.55, 0.2, 0.5)):
            opts = dict(agent_scale=scale)
            ref = oracle_lib.OracleBatch(env_id, n, options=opts)
            seeds = np.arange(n, dtype=np.int64) + 100 * phase
            obs, _ = env.reset(seed=seeds, options=opts)
            assert np.array_equal(obs.cpu().numpy(), ref.reset(seeds)), (env_id, scale, "reset")
            for t in range(40):
                a = (prng.integers(0, 4, n) if env.action_dim == 1 else prng.integers(0, 3, (n, 2))).astype(np.int32)
                obs, rew, done, _, _ = env.step(a)
                o2, r2, d2 = ref.step(a, autoreset=True)
                assert np.array_equal(done.cpu().numpy(), d2.astype(bool)) and np.array_equal(obs.cpu().numpy(), o2), (env_id, scale, t)
            ref.close()
        env.close()


DEBUG = [("MysteryPath-v0", dict(agent_scale=0.5, max_steps=40)), ("Endless-MysteryPath-v0", dict(agent_scale=0.45)),
         ("SearingSpotlights-v0", dict(agent_scale=0.5, exit_scale=1.0, coin_scale=0.75)), ("MortarMayhem-v0", dict(agent_scale=0.5))]


@pytest.mark.parametrize("env_id,opts", DEBUG, ids=[c[0] for c in DEBUG])
def test_debug_view_with_large_sprites(env_id, opts):
    import memory_gym_amd
    import oracle_lib

    n = 12
    env = memory_gym_amd.make(env_id, num_envs=n, device=0, render_mode="debug_rgb_array")
    ref = oracle_lib.OracleBatch(env_id, n, options=opts)
    seeds = np.arange(n, dtype=np.int64) + 9
    env.reset(seed=seeds, options=opts)
    ref.reset(seeds)
    prng = np.random.Generator(np.random.PCG64(11))
    for t in range(45):
        if t % 4 == 0:
            got = env.render().cpu().numpy()
            for i in range(n):
                assert np.array_equal(got[i], ref.envs[i].debug_view()), (env_id, "debug view of instance %d at step %d" % (i, t))
        a = (prng.integers(0, 4, n) if env.action_dim == 1 else prng.integers(0, 3, (n, 2))).astype(np.int32)
        env.step(a)
        ref.step(a, autoreset=True, want_obs=False)
    env.close()
    ref.close()


def test_stale_exits_keep_the_size_they_were_made_with():
    """use_exit = False keeps the instance's EARLIER exit on screen (searing_spotlights.py:431-435) -- an object of its own in the
    reference, so it keeps its size when exit_scale changes meanwhile.  The handle holds one pair of exit stamps per size still on
    some screen (csrc/mg_spot.hip: exit generations); a checkpoint carries them."""
    import memory_gym_amd
    import oracle_lib
    from memory_gym_amd.reset_params import process_reset_params

    env_id, n = "SearingSpotlights-v0", 24
    env = memory_gym_amd.make(env_id, num_envs=n, device=0)
    ref = oracle_lib.OracleBatch(env_id, n)
    prng = np.random.Generator(np.random.PCG64(17))
    seeds = np.arange(n, dtype=np.int64) + 40
    phases = [dict(exit_scale=0.5, max_steps=30, agent_health=100), dict(use_exit=False, exit_scale=1.0, max_steps=25, agent_health=100),
              dict(use_exit=False, exit_scale=0.3, max_steps=25, agent_health=100, exit_visible=True),
              dict(exit_scale=0.8, max_steps=30, agent_health=100), dict(use_exit=False, exit_scale=0.25, max_steps=20, agent_health=100)]
    for k, opts in enumerate(phases):
        ref.set_options(process_reset_params(env_id, opts))
        obs, _ = env.reset(seed=seeds if k == 0 else None, options=opts)
        assert np.array_equal(obs.cpu().numpy(), ref.reset(seeds if k == 0 else None)), "reset frames of phase %d" % k
        for t in range(45):
            a = np.stack([coin_seeker(ref.envs[i], prng) for i in range(n)]).astype(np.int32)
            obs, rew, done, _, _ = env.step(a)
            o2, r2, d2 = ref.step(a, autoreset=True)
            assert np.array_equal(done.cpu().numpy(), d2.astype(bool)), "phase %d step %d: done" % (k, t)
            assert np.array_equal(obs.cpu().numpy(), o2), "phase %d step %d: frames" % (k, t)
            if k == 2 and t == 20:  # a restore into a fresh handle brings the stamps of the stale exits along
                sd = env.state_dict()
                env.close()
                env = memory_gym_amd.make(env_id, num_envs=n, device=0)
                env.load_state_dict(sd)
    env.check_errors()
    env.close()
    ref.close()
