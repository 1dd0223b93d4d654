"""GPU (-m gpu): mg_get_state / mg_set_state for every family -- restoring a snapshot into a FRESH handle and replaying
the same actions must reproduce frames, rewards, dones and the RNG words (the reference cannot serialise an env)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [("MortarMayhem-Grid-v0", 1, 4, None), ("MortarMayhemB-v0", 2, 3, None), ("MysteryPath-v0", 2, 3, dict(max_steps=60)),
         ("Endless-MysteryPath-v0", 1, 4, None), ("SearingSpotlights-v0", 2, 3, None), ("Endless-SearingSpotlights-v0", 2, 3, None)]


@pytest.mark.parametrize("env_id,adim,n_act,options", CASES)
def test_restore_into_fresh_handle(env_id, adim, n_act, options):
    import memory_gym_amd
    import torch

    n = 192
    vis = (lambda o: o["visual_observation"] if isinstance(o, dict) else o)
    a_env = memory_gym_amd.make(env_id, num_envs=n, device=0)
    a_env.reset(seed=21, options=options)
    prng = np.random.Generator(np.random.PCG64(4))
    draw = (lambda: prng.integers(0, n_act, (n, adim)).astype(np.int32).squeeze(-1) if adim == 1 else prng.integers(0, n_act, (n, adim)).astype(np.int32))
    for _ in range(70):
        a_env.step(draw())
    sd = a_env.state_dict()
    b_env = memory_gym_amd.make(env_id, num_envs=n, device=0)
    b_env.reset(seed=999, options=options)  # different episodes until the snapshot is loaded (options belong to the handle)
    b_env.load_state_dict(sd)
    n_done = 0
    for t in range(90):
        a = draw()
        o1, r1, d1, _, i1 = a_env.step(a)
        o2, r2, d2, _, i2 = b_env.step(a)
        assert torch.equal(vis(o1), vis(o2)), "%s: frames differ %d steps after the restore" % (env_id, t)
        assert torch.equal(r1, r2) and torch.equal(d1, d2)
        m = d1.cpu().numpy()
        if m.any():
            assert torch.equal(i1["reward"][d1], i2["reward"][d2]) and torch.equal(i1["length"][d1], i2["length"][d2])
        n_done += int(m.sum())
    assert n_done > 0
    for i in (0, n // 2, n - 1):
        assert np.array_equal(a_env.rng_words(i), b_env.rng_words(i))
    a_env.close()
    b_env.close()


@pytest.mark.parametrize("env_id,adim,n_act,options", [CASES[0], CASES[2], ("SearingSpotlights-v0", 2, 3, dict(show_last_action=False, max_steps=80))])
def test_restore_into_never_reset_handle(env_id, adim, n_act, options):
    """the usual way to resume -- a fresh process creates the handle and loads the checkpoint, no reset.  The
    options in force travel with the checkpoint; auto-reset, reset(seed=None) and the final-observation path (step
    without auto-reset + masked reset(seed=None)) must work on the restored handle."""
    import memory_gym_amd
    import torch

    n = 96
    a_env = memory_gym_amd.make(env_id, num_envs=n, device=0, final_observation=True)
    a_env.reset(seed=5, options=options)
    prng = np.random.Generator(np.random.PCG64(9))
    draw = (lambda: prng.integers(0, n_act, (n, adim)).astype(np.int32).squeeze(-1) if adim == 1 else prng.integers(0, n_act, (n, adim)).astype(np.int32))
    for _ in range(50):
        a_env.step(draw())
    sd = a_env.state_dict()
    b_env = memory_gym_amd.make(env_id, num_envs=n, device=0, final_observation=True)
    b_env.load_state_dict(sd)  # no reset on this handle
    n_done = 0
    for t in range(120):
        a = draw()
        o1, r1, d1, _, i1 = a_env.step(a)
        o2, r2, d2, _, i2 = b_env.step(a)
        assert torch.equal(o1, o2), "%s: frames differ %d steps after the restore" % (env_id, t)
        assert torch.equal(r1, r2) and torch.equal(d1, d2)
        if bool(d1.any()):
            assert torch.equal(i1["final_observation"][d1], i2["final_observation"][d2])
        n_done += int(d1.sum())
    assert n_done > 0
    o1, _ = a_env.reset()  # seed=None on both: the streams continue
    o2, _ = b_env.reset()
    assert torch.equal(o1, o2)
    a_env.close()
    b_env.close()


def test_first_reset_without_seed():
    """gymnasium's common call pattern make(id).reset(): instances are seeded from OS entropy instead of raising."""
    import memory_gym_amd

    env = memory_gym_amd.make("MortarMayhem-Grid-v0", num_envs=8, device=0)
    obs, _ = env.reset()
    assert obs.shape == (8, 84, 84, 3) and bool(obs.any())
    env.step(np.zeros(8, np.int32))
    env.close()
    single = memory_gym_amd.make("Endless-SearingSpotlights-v0")
    o, info = single.reset()
    assert o.shape == (84, 84, 3)
    single.close()
