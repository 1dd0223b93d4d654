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
    prng = np.random.Generator(np.random.PCG64(1))
    episodes = 0
    for t in range(300):
        a = [int(prng.integers(0, 3)), int(prng.integers(0, 3))] if e.vec.action_dim == 2 else int(prng.integers(0, 4))
        o, rw, d, tr, info = e.step(a)
        o2, r2, d2 = r.step(a if isinstance(a, list) else [a, 0])
        assert isinstance(rw, float) and isinstance(d, bool) and tr is False
        assert np.array_equal(vis(o), o2) and rw == np.float32(r2) and d == d2, (env_id, t)
        if "ground_truth" in info:
            assert np.array_equal(info["ground_truth"].astype(np.float32), r.gt().astype(np.float32))
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
