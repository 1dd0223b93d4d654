"""GPU parity (-m gpu) for the Searing Spotlights family: HIP path through the C ABI vs the CPU oracle."""
import numpy as np
import pytest

from gpu_parity import check_terminal_info, run_parity

pytestmark = pytest.mark.gpu


def _toward(d):
    return 0 if abs(d) < 3 else (1 if d < 0 else 2)


def coin_seeker(e, prng):
    if prng.random() > 0.9:
        return [int(prng.integers(0, 3)), int(prng.integers(0, 3))]
    tx, ty = e.get("coin_x"), e.get("coin_y")
    if tx is None:  # finite variant: first remaining coin, else the exit
        c = e.get_list("coins")
        if c is not None and len(c) >= 2:
            tx, ty = c[0], c[1]
        else:
            tx, ty = e.get("exit_x"), e.get("exit_y")
    return [_toward(tx - e.get("ax")), _toward(ty - e.get("ay"))]


ESS_OPTS = [
    None,
    dict(agent_health=40, steps_per_coin=60, initial_spawns=5, spawn_interval=20, max_steps=400, reward_death=-1.0,
         reward_inside_spotlight=-0.01, reward_outside_spotlight=0.001),
    dict(agent_health=1000, spot_min_speed=0.01, spot_max_speed=0.05, spawn_interval=10),
    dict(sample_agent_position=False, visual_feedback=False, coins_visible=True),
]
SS_OPTS = [
    None,
    dict(num_coins=[1, 2, 3], agent_health=20, initial_spawns=2, max_steps=128, reward_death=-1.0,
         reward_inside_spotlight=-0.01, reward_outside_spotlight=0.001),
    dict(num_coins=[2], agent_health=50, light_dim_off_duration=3),
    dict(sample_agent_position=False, agent_health=100),
    dict(show_last_action=False),  # on its own: moves and widens the last-reward bar (searing_spotlights.py:385-390)
]


@pytest.mark.parametrize("opt_idx", range(len(ESS_OPTS)))
def test_endless_parity(opt_idx):
    n_done = run_parity("Endless-SearingSpotlights-v0", ESS_OPTS[opt_idx], n=160, steps=240, policy=coin_seeker, n_policy=64)
    assert n_done > 0 or opt_idx == 2


@pytest.mark.parametrize("opt_idx", range(len(SS_OPTS)))
def test_finite_parity(opt_idx):
    n_done = run_parity("SearingSpotlights-v0", SS_OPTS[opt_idx], n=160, steps=230, policy=coin_seeker, n_policy=64)
    assert n_done > 0


def test_terminal_info():
    assert check_terminal_info("Endless-SearingSpotlights-v0", steps=220) > 0
    assert check_terminal_info("SearingSpotlights-v0", steps=220) > 0


def test_full_size_sample():
    """BASELINE config C4 size (16,384 instances): a sample of instances must match single-instance oracles."""
    import memory_gym_amd
    import oracle_lib
    import torch

    n = 16384
    env = memory_gym_amd.make("Endless-SearingSpotlights-v0", num_envs=n, device=0)
    obs, _ = env.reset(seed=0)
    sample = [0, 1, 255, 4095, 8192, 16383]
    refs = {i: oracle_lib.OracleEnv("Endless-SearingSpotlights-v0") for i in sample}
    first = obs[sample].cpu().numpy()
    for k, i in enumerate(sample):
        assert np.array_equal(first[k], refs[i].reset(i))
    g = torch.Generator(device="cuda").manual_seed(0)
    for t in range(120):
        a = torch.randint(0, 3, (n, 2), device="cuda", generator=g, dtype=torch.int32)
        obs, rew, done, _, _ = env.step(a)
        ac = a[sample].cpu().numpy()
        got = obs[sample].cpu().numpy()
        for k, i in enumerate(sample):
            o, r, d = refs[i].step(ac[k])
            if d:
                o = refs[i].reset(None)
            assert np.array_equal(got[k], o), "instance %d differs at step %d" % (i, t)
    env.close()


@pytest.mark.parametrize("n", [20480, 40960])
def test_large_launch_store_flavour_sample(n):
    """Launches of more than 16,384 frames stream the observations with NON-TEMPORAL stores (mg_raster.hpp raster_nt(); up to
    16,384: plain stores) and, beyond 24,576 frames, on the larger persistent grid: another instantiation of the raster kernel
    than every small parity test runs.  A sample of instances -- both ends, around the grid sizes -- against single-instance
    oracles, frame by frame."""
    import memory_gym_amd
    import oracle_lib
    import torch

    env = memory_gym_amd.make("Endless-SearingSpotlights-v0", num_envs=n, device=0)
    obs, _ = env.reset(seed=0)
    sample = [0, 1, 9727, 9728, 14335, 14336, 16384, n // 2 + 1, n - 2, n - 1]
    refs = {i: oracle_lib.OracleEnv("Endless-SearingSpotlights-v0") for i in sample}
    first = obs[sample].cpu().numpy()
    for k, i in enumerate(sample):
        assert np.array_equal(first[k], refs[i].reset(i))
    g = torch.Generator(device="cuda").manual_seed(3)
    for t in range(60):
        a = torch.randint(0, 3, (n, 2), device="cuda", generator=g, dtype=torch.int32)
        obs, rew, done, _, _ = env.step(a)
        ac = a[sample].cpu().numpy()
        got = obs[sample].cpu().numpy()
        for k, i in enumerate(sample):
            o, r, d = refs[i].step(ac[k])
            if d:
                o = refs[i].reset(None)
            assert np.array_equal(got[k], o), "instance %d differs at step %d" % (i, t)
    env.check_errors()
    env.close()


def test_slot_capacity_is_checked_at_reset():
    """Option sets that must overflow the 16 slots per instance are refused up front; rarer overflows surface as a
    RuntimeError from step() (host-mapped error word, no synchronisation needed to see it)."""
    import memory_gym_amd
    import torch

    env = memory_gym_amd.make("Endless-SearingSpotlights-v0", num_envs=64, device=0)
    with pytest.raises(RuntimeError, match="spotlights alive at once"):
        env.reset(seed=0, options=dict(initial_spawns=5, spawn_interval=5))
    # passes the static check (fastest spotlight: 5 + 134 // 10 = 18 > 16 is refused, 12 -> 5 + 11 = 16 fits) ...
    with pytest.raises(RuntimeError, match="spotlights alive at once"):
        env.reset(seed=0, options=dict(initial_spawns=5, spawn_interval=10))
    env.reset(seed=0, options=dict(initial_spawns=5, spawn_interval=12, agent_health=100000, steps_per_coin=100000))
    # ... but slow spotlights live up to 400 steps, so an agent that survives overflows: step() must say so
    a = torch.zeros((64, 2), dtype=torch.int32, device="cuda")
    with pytest.raises(RuntimeError, match="more than 16 live spotlights"):
        for _ in range(400):
            env.step(a)
            torch.cuda.synchronize()
    env.check_errors()  # cleared by the raise
    env.close()


@pytest.mark.parametrize("env_id", ["SearingSpotlights-v0", "Endless-SearingSpotlights-v0"])
@pytest.mark.parametrize("threshold,duration", [(128, 6), (200, 3), (40, 0), (0, 4), (255, 0), (300, 5), (-5, 0)])
def test_light_threshold(env_id, threshold, duration):
    """The dark layer's alpha ramps by int(255 / duration) while it is <= light_threshold (then stays: a partly lit
    board), or is set to the threshold at once when the duration is 0 (searing_spotlights.py:397,471-475); set_alpha clamps."""
    run_parity(env_id, dict(light_threshold=threshold, light_dim_off_duration=duration), n=32, steps=40)


@pytest.mark.parametrize("opts", [dict(agent_visible=True), dict(exit_visible=True), dict(agent_visible=True, exit_visible=True, coins_visible=True),
                                  dict(exit_visible=True, coins_visible=True), dict(agent_visible=True, coins_visible=True, light_threshold=120)])
def test_layers_above_the_dark_layer_finite(opts):
    """coins_visible / exit_visible / agent_visible move a layer from below the spotlight layer to above it
    (searing_spotlights.py:524-545); the agent lands on top of the top bar."""
    run_parity("SearingSpotlights-v0", dict(opts, sample_agent_position=False), n=48, steps=120, policy=coin_seeker, n_policy=24)


@pytest.mark.parametrize("opts", [dict(agent_visible=True), dict(agent_visible=True, coins_visible=True), dict(agent_visible=True, light_dim_off_duration=0, light_threshold=90)])
def test_layers_above_the_dark_layer_endless(opts):
    run_parity("Endless-SearingSpotlights-v0", opts, n=48, steps=200, policy=coin_seeker, n_policy=24)


@pytest.mark.parametrize("env_id,opts", [("SearingSpotlights-v0", dict(exit_scale=0.75, coin_scale=0.5)), ("SearingSpotlights-v0", dict(exit_scale=0.3, agent_scale=0.2)),
                                         ("Endless-SearingSpotlights-v0", dict(coin_scale=0.6, agent_scale=0.2))])
def test_scale_options(env_id, opts):
    run_parity(env_id, opts, n=48, steps=150, policy=coin_seeker, n_policy=24)


@pytest.mark.parametrize("n", [700, 3000])
def test_every_instance_truncated_in_the_same_step(n):
    """The fused raster / reset launch takes as few resets per serving workgroup as serve the queue in one round (1 .. 8, chosen in
    the kernel from the queue's length: csrc/mg_spot.hip SpotServeArgs): with max_steps = 9 every instance is truncated in steps 9,
    18 and 27 -- n resets at once, two and eight per workgroup, several rounds for the larger batch."""
    run_parity("SearingSpotlights-v0", dict(max_steps=9), n=n, steps=30, check_every=3)


@pytest.mark.parametrize("env_id", ["SearingSpotlights-v0", "Endless-SearingSpotlights-v0"])
@pytest.mark.parametrize("count", [1, 2, 3, 4, 5])
def test_reset_spotlights_from_one_batch_of_outputs(env_id, count):
    """new_spots_at_reset (csrc/mg_spot.hip): a reset's `initial_spawns` spotlights come from the generator's next 16 outputs at once
    (PCG64 jump-ahead across the instance's 16 lanes), and which half of which output feeds which draw depends on whether the stream
    arrives with a buffered 32-bit half.  Both layouts, every count the batch form takes (1..5): with `sample_agent_position` on, the
    seeded reset draws one more 32-bit number in front of the spotlights than with it off, so the two settings reach the spotlights in
    opposite buffer states; the auto-resets of the run arrive in either.  Checked: every reset frame (the spotlights are drawn), the
    generator's words of EVERY instance right after the seeded reset and at the end."""
    import memory_gym_amd
    import oracle_lib

    n, steps = 64, 40
    for sample in (True, False):
        opts = dict(initial_spawns=count, sample_agent_position=sample, agent_health=6, max_steps=17)  # (short episodes: every instance auto-resets twice)
        env = memory_gym_amd.make(env_id, num_envs=n, device=0)
        ref = oracle_lib.OracleBatch(env_id, n, options=opts)
        seeds = np.arange(n, dtype=np.int64) * 13 + 5 * count
        obs, _ = env.reset(seed=seeds, options=opts)
        assert np.array_equal(obs.cpu().numpy(), ref.reset(seeds)), "reset frames differ (count %d, sample_agent_position %s)" % (count, sample)
        buffered = 0
        for i in range(n):
            w = ref.envs[i].rng_words()
            assert np.array_equal(env.rng_words(i), w), "instance %d: generator differs right after the seeded reset" % i
            buffered += int(w[4])
        assert buffered in (0, n)  # the seeded reset's draw count is fixed by the options ...
        if sample:
            first = buffered
        else:
            assert buffered != first  # ... and differs by one between the two settings: both buffer states are covered
        prng = np.random.Generator(np.random.PCG64(17 + count))
        n_done = 0
        for t in range(steps):
            a = prng.integers(0, 3, (n, 2)).astype(np.int32)
            obs, rew, done, _, _ = env.step(a)
            o2, r2, d2 = ref.step(a, autoreset=True)
            d = done.cpu().numpy()
            assert np.array_equal(d, d2.astype(bool)) and np.array_equal(rew.cpu().numpy(), r2.astype(np.float32)), "step %d" % t
            assert np.array_equal(obs.cpu().numpy(), o2), "frames differ at step %d (count %d, sample_agent_position %s)" % (t, count, sample)
            n_done += int(d.sum())
        assert n_done >= 2 * n  # (the auto-resets arrive in either buffer state)
        for i in range(n):
            assert np.array_equal(env.rng_words(i), ref.envs[i].rng_words()), "instance %d: generator diverged" % i
        env.check_errors()
        env.close()
        ref.close()


@pytest.mark.slow
@pytest.mark.parametrize("env_id,opts,steps", [("Endless-SearingSpotlights-v0", ESS_OPTS[0], 700), ("Endless-SearingSpotlights-v0", ESS_OPTS[1], 700),
                                               ("SearingSpotlights-v0", SS_OPTS[0], 600), ("SearingSpotlights-v0", SS_OPTS[1], 600)])
def test_long_runs(env_id, opts, steps):
    """(marked slow: MEMGYM_FAST=1 leaves it out) long lock-step runs, every frame compared (a rejected Lemire draw, long episodes, many spawns)."""
    assert run_parity(env_id, opts, n=160, steps=steps, policy=coin_seeker, n_policy=64) > 0
