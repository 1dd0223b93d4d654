"""hide_chessboard / black_background (-m gpu): HIP path vs CPU oracle over a sequence of episodes on the SAME handles.

The reference repaints the two background surfaces an environment object keeps for its lifetime
(/root/reference/memory_gym/searing_spotlights.py:349-351, 234-235, 420-421; endless_searing_spotlights.py:313-315,
223-224, 376-377), so what one episode's options did to them is still there in the next, whatever its options; spotlights
spawned under black_background carry a white 1-px border (pygame_assets.py:62, 112-113).  Every phase passes the complete
option dictionary, like the reference's process_reset_params() does."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FAST = dict(spot_min_speed=0.01, spot_max_speed=0.03, initial_spawns=4, agent_health=1000)
PHASES = [
    (dict(), 25),
    (dict(FAST, black_background=True, light_dim_off_duration=0, light_threshold=150), 70),   # borders, blended at alpha 150
    (dict(FAST, hide_chessboard=True, light_dim_off_duration=2), 40),                            # bordered spotlights still alive
    (dict(FAST, black_background=True, visual_feedback=False), 40),
    (dict(), 40),                                                                                # defaults again: nothing comes back
]


def _full(env_id, opts):
    from memory_gym_amd.reset_params import DEFAULTS
    d = dict(DEFAULTS[env_id])
    d.update(opts)
    return d


@pytest.mark.parametrize("env_id", ["SearingSpotlights-v0", "Endless-SearingSpotlights-v0"])
def test_sticky_backgrounds_and_borders(env_id):
    import memory_gym_amd
    import oracle_lib

    n = 96
    env = memory_gym_amd.make(env_id, num_envs=n, device=0)
    ref = oracle_lib.OracleBatch(env_id, n)
    prng = np.random.Generator(np.random.PCG64(17))
    seen_border = seen_white = seen_black = False
    for ph, (opts, steps) in enumerate(PHASES):
        if env_id.startswith("Endless") and "agent_health" in opts:
            opts = dict(opts, spawn_interval=8)
        o = _full(env_id, opts)
        seeds = np.arange(n, dtype=np.int64) * 5 + 100 * ph
        obs, _ = env.reset(seed=seeds, options=o)
        ref.set_options(o)
        want = ref.reset(seeds)
        got = obs.cpu().numpy()
        assert np.array_equal(got, want), "phase %d: reset frames differ for envs %s" % (ph, np.nonzero((got != want).reshape(n, -1).any(1))[0][:8])
        for t in range(steps):
            a = prng.integers(0, 3, (n, 2)).astype(np.int32)
            obs, rew, done, _, _ = env.step(a)
            want, r2, d2 = ref.step(a, autoreset=True)
            got = obs.cpu().numpy()
            assert np.array_equal(done.cpu().numpy(), d2.astype(bool)), "phase %d step %d: done" % (ph, t)
            assert np.array_equal(rew.cpu().numpy(), r2.astype(np.float32)), "phase %d step %d: reward" % (ph, t)
            if not np.array_equal(got, want):
                bad = np.nonzero((got != want).reshape(n, -1).any(1))[0]
                px = np.argwhere((got[bad[0]] != want[bad[0]]).any(-1))
                raise AssertionError("phase %d step %d: frames differ for envs %s; env %d: %d px, first (x,y)=%s hip=%s oracle=%s" % (
                    ph, t, bad[:8], bad[0], len(px), px[0], got[bad[0]][tuple(px[0])], want[bad[0]][tuple(px[0])]))
            if t % 13 == 5:
                dbg = env.render_debug().cpu().numpy()
                for i in (0, n // 2, n - 1):
                    assert np.array_equal(dbg[i], ref.envs[i].debug_view()), "phase %d step %d: debug view of env %d" % (ph, t, i)
        body = got[:, :, 8:, :]  # below the top bar
        if ph == 1:
            # the dark layer over a black board leaves 0; a border pixel is white blended at alpha 150 over black: 150
            seen_border = bool(((body == 150).all(-1)).any())
            seen_black = bool((body.reshape(n, -1).max(1) <= 255).all() and ((body == 0).all(-1)).mean() > 0.3)
        if ph == 2:
            seen_white = bool(((body == 255).all(-1)).sum() > 200)  # the board is white wherever a spotlight shows it
    assert seen_border and seen_black and seen_white
    env.check_errors()
    env.close()
    ref.close()


def test_state_round_trip_keeps_borders():
    """A checkpoint taken while bordered spotlights are alive restores into a fresh handle that then draws them."""
    import memory_gym_amd
    import oracle_lib

    env_id, n = "Endless-SearingSpotlights-v0", 48
    o = _full(env_id, dict(FAST, spawn_interval=8, black_background=True, light_dim_off_duration=0, light_threshold=200))
    env = memory_gym_amd.make(env_id, num_envs=n, device=0)
    ref = oracle_lib.OracleBatch(env_id, n, options=o)
    seeds = np.arange(n, dtype=np.int64) + 9
    env.reset(seed=seeds, options=o)
    ref.reset(seeds)
    prng = np.random.Generator(np.random.PCG64(3))
    for t in range(30):
        a = prng.integers(0, 3, (n, 2)).astype(np.int32)
        env.step(a)
        ref.step(a, autoreset=True)
    sd = env.state_dict()
    env.close()
    env2 = memory_gym_amd.make(env_id, num_envs=n, device=0)
    env2.load_state_dict(sd)
    for t in range(30):
        a = prng.integers(0, 3, (n, 2)).astype(np.int32)
        obs, _, _, _, _ = env2.step(a)
        want, _, _ = ref.step(a, autoreset=True)
        assert np.array_equal(obs.cpu().numpy(), want), "step %d after the restore" % t
    env2.close()
    ref.close()
