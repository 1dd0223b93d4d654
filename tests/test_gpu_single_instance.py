"""GPU (-m gpu): the single-instance adapter (`memory_gym_amd.make(id)`), i.e. the reference's exact call shapes
(numpy observation, Python float reward, bool done, False, dict info; explicit reset() after a terminal step)
against a single-instance oracle -- BASELINE config C1's plumbing."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("env_id", ["MortarMayhem-Grid-v0", "Endless-SearingSpotlights-v0", "Endless-MysteryPath-v0", "MortarMayhemB-v0"])
def test_reference_shaped_loop(env_id):
    import memory_gym_amd
    import oracle_lib

    e = memory_gym_amd.make(env_id)
    r = oracle_lib.OracleEnv(env_id)
    vis = (lambda x: x["visual_observation"] if isinstance(x, dict) else x)
    o, info = e.reset(seed=5)
    assert isinstance(vis(o), np.ndarray) and vis(o).shape == (84, 84, 3) and np.array_equal(vis(o), r.reset(5))
    if e.vec.gt_dim:
        assert info["ground_truth"].dtype == np.float64 and np.array_equal(info["ground_truth"], r.gt())
    assert np.array_equal(e.render(), vis(o).transpose(1, 0, 2))
    prng = np.random.Generator(np.random.PCG64(1))
    episodes = 0
    for t in range(300):
        a = [int(prng.integers(0, 3)), int(prng.integers(0, 3))] if e.vec.action_dim == 2 else int(prng.integers(0, 4))
        o, rw, d, tr, info = e.step(a)
        o2, r2, d2 = r.step(a if isinstance(a, list) else [a, 0])
        assert isinstance(rw, float) and isinstance(d, bool) and tr is False
        assert np.array_equal(vis(o), o2) and rw == r2 and d == d2, (env_id, t)  # rw: the reference's Python float (a double)
        if "ground_truth" in info:  # the reference's float64 array, not a float32 rounding of it (0.6, not 0.60000002)
            assert info["ground_truth"].dtype == np.float64 and np.array_equal(info["ground_truth"], r.gt()), (env_id, t, info["ground_truth"], r.gt())
        if d:
            assert info["reward"] == r.get("info_reward") and info["length"] == r.get("info_length")
            episodes += 1
            o, info = e.reset()
            assert np.array_equal(vis(o), r.reset(None))
        else:
            assert "reward" not in info
    assert episodes > 0
    assert np.array_equal(e.vec.rng_words(0), r.rng_words())
    e.close()


@pytest.mark.parametrize("env_id", ["Endless-MortarMayhem-v0", "Endless-SearingSpotlights-v0", "Endless-MysteryPath-v0"])
def test_batched_ground_truth_in_float64(env_id):
    """mg_info_buffers.gt64_dev / mg_ground_truth64: info["ground_truth"] as the reference's doubles for every instance of a batch
    (ground_truth64=True), after resets, steps and same-step auto-resets; the default float32 tensor is its rounding."""
    import memory_gym_amd
    import oracle_lib

    n = 96
    env = memory_gym_amd.make(env_id, num_envs=n, device=0, ground_truth64=True)
    ref = oracle_lib.OracleBatch(env_id, n)
    seeds = np.arange(n, dtype=np.int64) + 3
    _, info = env.reset(seed=seeds)
    ref.reset(seeds)
    want = np.stack([e.gt() for e in ref.envs])
    assert info["ground_truth"].dtype.is_floating_point and info["ground_truth"].element_size() == 8
    assert np.array_equal(info["ground_truth"].cpu().numpy(), want)
    prng = np.random.Generator(np.random.PCG64(2))
    for t in range(150):
        a = (prng.integers(0, 4, n) if env.action_dim == 1 else prng.integers(0, 3, (n, 2))).astype(np.int32)
        _, _, done, _, info = env.step(a)
        ref.step(a, autoreset=True, want_obs=False)
        want = np.stack([e.gt() for e in ref.envs])
        assert np.array_equal(info["ground_truth"].cpu().numpy(), want), (env_id, t)
        assert np.array_equal(env.gt.cpu().numpy(), want.astype(np.float32)), (env_id, t)
    env.close()
    ref.close()
