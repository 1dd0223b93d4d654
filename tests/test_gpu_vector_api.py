"""GPU (-m gpu): the gymnasium-0.29 vector convention -- mg_step with mg_info_buffers.final_obs_dev.

For every instance that finishes in a step the HIP path must deliver BOTH the terminal observation
(infos["final_observation"]) and the first observation of the next episode (obs), exactly what a loop over
single-instance reference envs produces with `obs_T = env.step(a)`, `obs_0 = env.reset()`; everything else (rewards,
dones, RNG consumption) must be identical to the fused same-step auto-reset path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [("MortarMayhem-Grid-v0", 1, 4, 120), ("Endless-MortarMayhem-v0", 2, 3, 120), ("MysteryPath-Grid-v0", 1, 4, 150),
         ("Endless-MysteryPath-v0", 1, 4, 120), ("Endless-SearingSpotlights-v0", 2, 3, 260), ("SearingSpotlights-v0", 2, 3, 200),
         ("MortarMayhemB-v0", 2, 3, 120)]


@pytest.mark.parametrize("env_id,adim,n_act,steps", CASES)
def test_final_observation_matches_oracle(env_id, adim, n_act, steps):
    import memory_gym_amd
    import oracle_lib

    n = 48
    envs = memory_gym_amd.GymnasiumVectorEnv(env_id, n, device=0)
    fused = memory_gym_amd.make(env_id, num_envs=n, device=0)  # same-step auto-reset without final observations
    refs = [oracle_lib.OracleEnv(env_id) for _ in range(n)]
    obs, _ = envs.reset(seed=100)
    fobs, _ = fused.reset(seed=100)
    vis = (lambda o: o["visual_observation"] if isinstance(o, dict) else o)
    for i, r in enumerate(refs):
        assert np.array_equal(vis(obs)[i].cpu().numpy(), r.reset(100 + i))
    prng = np.random.Generator(np.random.PCG64(8))
    n_final = 0
    for t in range(steps):
        a = prng.integers(0, n_act, (n, adim)).astype(np.int32)
        obs, rew, term, trunc, infos = envs.step(a[:, 0] if adim == 1 else a)
        o2, r2, d2, _, _ = fused.step(a[:, 0] if adim == 1 else a)
        assert np.array_equal(vis(obs).cpu().numpy(), vis(o2).cpu().numpy()) and np.array_equal(rew.cpu().numpy(), r2.cpu().numpy())
        assert np.array_equal(term.cpu().numpy(), d2.cpu().numpy()) and not trunc.any()
        got, fin, mask = vis(obs).cpu().numpy(), infos["final_observation"].cpu().numpy(), infos["_final_observation"].cpu().numpy()
        for i, r in enumerate(refs):
            o, rw, dn = r.step(a[i])
            assert dn == mask[i]
            if dn:
                assert np.array_equal(fin[i], o), "%s: terminal frame of env %d differs at step %d" % (env_id, i, t)
                assert infos["final_info"]["reward"][i].item() == r.get("info_reward")
                assert infos["final_info"]["length"][i].item() == r.get("info_length")
                o = r.reset(None)
                n_final += 1
            assert np.array_equal(got[i], o), "%s: observation of env %d differs at step %d" % (env_id, i, t)
    assert n_final > 0
    for i in (0, n - 1):
        assert np.array_equal(envs.env.rng_words(i), refs[i].rng_words()) and np.array_equal(fused.rng_words(i), refs[i].rng_words())
    envs.close()
    fused.close()


def test_numpy_mode_layout():
    """as_numpy=True reproduces gymnasium's host-side containers (object arrays with None for running envs)."""
    import memory_gym_amd

    envs = memory_gym_amd.GymnasiumVectorEnv("MortarMayhem-Grid-v0", 32, device=0, as_numpy=True)
    obs, infos = envs.reset(seed=0)
    assert isinstance(obs, np.ndarray) and obs.shape == (32, 84, 84, 3) and obs.dtype == np.uint8
    prng = np.random.Generator(np.random.PCG64(1))
    seen = False
    for _ in range(80):
        obs, rew, term, trunc, infos = envs.step(prng.integers(0, 4, 32))
        assert rew.dtype == np.float64 and term.dtype == np.bool_ and trunc.dtype == np.bool_
        if term.any():
            seen = True
            fo, fi = infos["final_observation"], infos["final_info"]
            assert fo.dtype == object and fi.dtype == object and np.array_equal(infos["_final_observation"], term)
            for i in range(32):
                if term[i]:
                    assert fo[i].shape == (84, 84, 3) and set(fi[i]) == {"reward", "length", "success", "commands_completed"}
                    assert isinstance(fi[i]["length"], int)
                else:
                    assert fo[i] is None and fi[i] is None
        else:
            assert "final_observation" not in infos
    assert seen
    envs.close()
