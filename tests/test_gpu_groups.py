"""GPU (-m gpu): instance groups (include/memgym.h: mg_set_groups) -- a handle whose instances are stepped in 2 or 4 blocks on
streams of their own, one block's logic kernel under the previous block's raster launch -- produce bit for bit what the same
handle produces with one block: observations, rewards (float32 and the reference's double), dones, ground truth, terminal
info, final observations and the RNG words of every instance; across same-step auto-resets, masked resets and a checkpoint."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [("MortarMayhem-Grid-v0", None, 2), ("MortarMayhem-Grid-v0", None, 4), ("Endless-MortarMayhem-v0", dict(max_steps=40), 2),
         ("MysteryPath-v0", dict(max_steps=24), 2), ("MysteryPath-Grid-v0", dict(max_steps=16), 4),
         ("Endless-MysteryPath-v0", None, 2), ("SearingSpotlights-v0", dict(max_steps=32, num_coins=[1, 2]), 2),
         ("Endless-SearingSpotlights-v0", dict(agent_health=3), 4), ("MortarMayhemB-Grid-v0", None, 2)]


def _vis(o):
    return o["visual_observation"] if isinstance(o, dict) else o


@pytest.mark.parametrize("env_id,options,groups", CASES, ids=["%s-g%d" % (c[0], c[2]) for c in CASES])
def test_groups_equal_one_block(env_id, options, groups):
    import memory_gym_amd

    n = 512
    a_env = memory_gym_amd.make(env_id, num_envs=n, device=0, final_observation=True)
    b_env = memory_gym_amd.make(env_id, num_envs=n, device=0, final_observation=True, groups=groups)
    assert b_env.groups == groups
    seeds = torch.arange(n, dtype=torch.int64, device="cuda") * 3 + 1
    oa, ia = a_env.reset(seed=seeds, options=options)
    ob, ib = b_env.reset(seed=seeds, options=options)
    assert torch.equal(_vis(oa), _vis(ob))
    if isinstance(oa, dict):
        assert torch.equal(oa["vector_observation"], ob["vector_observation"])
    g = torch.Generator(device="cuda").manual_seed(11)
    disc = a_env.action_dim == 1
    n_done = 0
    for t in range(70):
        a = torch.randint(0, 4 if disc else 3, (n,) if disc else (n, 2), device="cuda", generator=g, dtype=torch.int32)
        oa, ra, da, _, ia = a_env.step(a)
        ob, rb, db, _, ib = b_env.step(a)
        assert torch.equal(_vis(oa), _vis(ob)), "%s: observations differ at step %d" % (env_id, t)
        assert torch.equal(ra, rb) and torch.equal(a_env.reward64, b_env.reward64) and torch.equal(da, db)
        if a_env.gt_dim:
            assert torch.equal(ia["ground_truth"], ib["ground_truth"])
        if da.any():
            n_done += int(da.sum())
            assert torch.equal(ia["final_observation"][da], ib["final_observation"][db])
            for k in ["reward", "length"] + a_env.info_names:
                assert torch.equal(ia[k][da], ib[k][db]), k
        if t == 30:  # a masked reset(seed=None) of every third instance in the middle
            m = (torch.arange(n, device="cuda") % 3) == 0
            oa, _ = a_env.reset(mask=m)
            ob, _ = b_env.reset(mask=m)
            assert torch.equal(_vis(oa), _vis(ob))
    assert n_done > 0 or "Endless-Mystery" in env_id
    for i in (0, 1, n // groups - 1, n // groups, n - 1):
        assert np.array_equal(a_env.rng_words(i), b_env.rng_words(i)), "RNG stream of instance %d differs" % i
    a_env.check_errors()
    b_env.check_errors()
    # checkpoint: a grouped handle restores into a grouped handle and goes on identically; another grouping is refused
    sd = b_env.state_dict()
    c_env = memory_gym_amd.make(env_id, num_envs=n, device=0, groups=groups)
    c_env.load_state_dict(sd)
    a = torch.randint(0, 4 if disc else 3, (n,) if disc else (n, 2), device="cuda", generator=g, dtype=torch.int32)
    ob, rb, db, _, _ = b_env.step(a)
    oc, rc, dc, _, _ = c_env.step(a)
    assert torch.equal(_vis(ob), _vis(oc)) and torch.equal(rb, rc) and torch.equal(db, dc)
    with pytest.raises(RuntimeError, match="instance group"):
        a_env.load_state_dict(sd)
    for e in (a_env, b_env, c_env):
        e.close()


def test_grouping_is_fixed_by_the_first_reset_and_needs_divisibility():
    import ctypes as C

    import memory_gym_amd
    from memory_gym_amd import _native

    with pytest.raises(RuntimeError, match="divisible"):
        memory_gym_amd.make("MortarMayhem-Grid-v0", num_envs=6, device=0, groups=4)
    env = memory_gym_amd.make("MortarMayhem-Grid-v0", num_envs=8, device=0, groups=2)
    env.reset(seed=torch.arange(8, dtype=torch.int64, device="cuda"))
    assert _native.LIB.mg_groups(env._h) == 2
    assert _native.LIB.mg_set_groups(env._h, 4) != 0 and "fixed" in _native.last_error()
    env.close()
