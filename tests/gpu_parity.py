"""Shared driver for the -m gpu parity tests: HIP path (through the C ABI) vs CPU oracle in lock-step."""
import numpy as np


def _split(obs):
    """MortarMayhemB* return the reference's Dict observation."""
    if isinstance(obs, dict):
        return obs["visual_observation"], obs["vector_observation"]
    return obs, None


def _check_vec(env_id, vec, ref, where):
    if vec is None:
        return
    want = np.stack([e.get_list("vec") for e in ref.envs]).astype(np.float32)
    got = vec.cpu().numpy()
    assert got.shape == want.shape and np.array_equal(got, want), "%s: vector_observation differs %s: envs %s" % (
        env_id, where, np.nonzero((got != want).any(1))[0][:8])


def run_parity(env_id, options, n, steps, policy=None, n_policy=0, check_every=1, seed0=3, want_counters=()):
    import memory_gym_amd
    import oracle_lib

    env = memory_gym_amd.make(env_id, num_envs=n, device=0)
    ref = oracle_lib.OracleBatch(env_id, n, options=options)
    disc = env.action_dim == 1
    seeds = np.arange(n, dtype=np.int64) * 7 + seed0
    obs, info = env.reset(seed=seeds, options=options)
    obs, vec = _split(obs)
    o0 = ref.reset(seeds)
    _check_vec(env_id, vec, ref, "after reset")
    got = obs.cpu().numpy()
    if not np.array_equal(got, o0):
        bad = np.nonzero((got != o0).reshape(n, -1).any(1))[0]
        px = np.argwhere((got[bad[0]] != o0[bad[0]]).any(-1))
        raise AssertionError("%s: reset frames differ for envs %s; env %d: %d px, first (x,y)=%s hip=%s oracle=%s" % (
            env_id, bad[:8], bad[0], len(px), px[0], got[bad[0]][tuple(px[0])], o0[bad[0]][tuple(px[0])]))
    if env.gt_dim:
        gt_ref = np.stack([e.gt() for e in ref.envs]).astype(np.float32)
        assert np.array_equal(info["ground_truth"].cpu().numpy(), gt_ref), "reset ground_truth differs"
    prng = np.random.Generator(np.random.PCG64(99))
    n_done = 0
    for t in range(steps):
        a = (prng.integers(0, 4, (n, 1)) if disc else prng.integers(0, 3, (n, 2))).astype(np.int32)
        if policy is not None:
            for i in range(n_policy):
                act = policy(ref.envs[i], prng)
                a[i, :a.shape[1]] = act[:a.shape[1]]
        obs, rew, done, trunc, info = env.step(a[:, 0] if disc else a)
        obs, vec = _split(obs)
        o2, r2, d2 = ref.step(a[:, 0] if disc else a, autoreset=True, want_obs=(t % check_every == 0))
        _check_vec(env_id, vec, ref, "at step %d" % t)
        d = done.cpu().numpy()
        assert np.array_equal(d, d2.astype(bool)), "%s: done differs at step %d: envs %s" % (
            env_id, t, np.nonzero(d != d2.astype(bool))[0][:8])
        rg = rew.cpu().numpy()
        assert np.array_equal(rg, r2.astype(np.float32)), "%s: reward differs at step %d: envs %s" % (
            env_id, t, np.nonzero(rg != r2.astype(np.float32))[0][:8])
        assert not trunc.any()
        if t % check_every == 0:
            got = obs.cpu().numpy()
            if not np.array_equal(got, o2):
                env.check_errors()  # a capacity error flagged by the kernels explains a mismatch better than pixels do
                bad = np.nonzero((got != o2).reshape(n, -1).any(1))[0]
                px = np.argwhere((got[bad[0]] != o2[bad[0]]).any(-1))
                raise AssertionError("%s: frame differs at step %d for envs %s; env %d: %d px, first (x,y)=%s hip=%s oracle=%s" % (
                    env_id, t, bad[:8], bad[0], len(px), px[0], got[bad[0]][tuple(px[0])], o2[bad[0]][tuple(px[0])]))
        n_done += int(d.sum())
        if env.gt_dim:
            gt_ref = np.stack([e.gt() for e in ref.envs]).astype(np.float32)
            assert np.array_equal(info["ground_truth"].cpu().numpy(), gt_ref), "ground_truth differs at step %d" % t
    for i in sorted({0, min(1, n - 1), n // 2, n - 1}):
        assert np.array_equal(env.rng_words(i), ref.envs[i].rng_words()), "RNG stream of env %d diverged" % i
    env.check_errors()
    for name in want_counters:  # the arrangement under test really ran (mg_debug_counter)
        assert env.debug_counter(name) > 0, "%s: counter %s is zero after %d steps" % (env_id, name, steps)
    env.close()
    ref.close()
    return n_done


def check_terminal_info(env_id, n=64, steps=200, options=None):
    """End-of-episode info vs single-instance oracles (reward sum in double, length, per-env extras)."""
    import memory_gym_amd
    import oracle_lib

    env = memory_gym_amd.make(env_id, num_envs=n, device=0)
    refs = [oracle_lib.OracleEnv(env_id) for _ in range(n)]
    seeds = np.arange(n, dtype=np.int64) + 1000
    env.reset(seed=seeds, options=options)
    for i, r in enumerate(refs):
        r.reset(int(seeds[i]), options=options, want_obs=False)
    prng = np.random.Generator(np.random.PCG64(5))
    disc = env.action_dim == 1
    checked = 0
    for t in range(steps):
        a = (prng.integers(0, 4, (n, 1)) if disc else prng.integers(0, 3, (n, 2))).astype(np.int32)
        _, _, done, _, info = env.step(a[:, 0] if disc else a)
        done = done.cpu().numpy()
        for i, r in enumerate(refs):
            _, _, d = r.step(a[i], want_obs=False)
            assert d == done[i]
            if d:
                assert info["reward"][i].item() == r.get("info_reward"), (env_id, "reward")
                assert info["length"][i].item() == r.get("info_length"), (env_id, "length")
                for nm in env.info_names:
                    assert info[nm][i].item() == np.float32(r.get("info_" + nm)), (env_id, nm)
                checked += 1
                r.reset(None, want_obs=False)
    env.close()
    return checked
