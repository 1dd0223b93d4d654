"""GPU (-m gpu): the multi-GPU partition is "instance i is seeded i whatever the world size".  Emulated
on one GPU: one handle with N instances vs W handles holding shard_range(N, r, W) with shard_seeds -- concatenated
observations / rewards / dones must be identical at every step (including ragged shards, N % W != 0)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("env_id,adim,n_act,world", [("Endless-MortarMayhem-v0", 2, 3, 3), ("MysteryPath-Grid-v0", 1, 4, 2),
                                                     ("Endless-SearingSpotlights-v0", 2, 3, 4)])
def test_results_do_not_depend_on_world_size(env_id, adim, n_act, world):
    import memory_gym_amd
    import torch
    from memory_gym_amd.dist import shard_range, shard_seeds

    n = 1000  # not a multiple of 3 / 64: ragged shards, partial waves
    whole = memory_gym_amd.make(env_id, num_envs=n, device=0)
    whole.reset(seed=shard_seeds(n, 0, 1, base_seed=7, device="cuda"))
    parts = []
    for r in range(world):
        lo, hi = shard_range(n, r, world)
        e = memory_gym_amd.make(env_id, num_envs=hi - lo, device=0)
        e.reset(seed=shard_seeds(n, r, world, base_seed=7, device="cuda"))
        parts.append((lo, hi, e))
    assert parts[0][0] == 0 and parts[-1][1] == n
    g = torch.Generator(device="cuda").manual_seed(4)
    n_done = 0
    for t in range(150):
        a = torch.randint(0, n_act, (n,) if adim == 1 else (n, adim), device="cuda", generator=g, dtype=torch.int32)
        o, rw, d, _, _ = whole.step(a)
        outs = [e.step(a[lo:hi]) for lo, hi, e in parts]
        assert torch.equal(o, torch.cat([x[0] for x in outs])), "observations differ at step %d" % t
        assert torch.equal(rw, torch.cat([x[1] for x in outs])) and torch.equal(d, torch.cat([x[2] for x in outs]))
        n_done += int(d.sum().item())
    assert n_done > 0
    for lo, hi, e in parts:
        assert np.array_equal(e.rng_words(0), whole.rng_words(lo)) and np.array_equal(e.rng_words(hi - lo - 1), whole.rng_words(hi - 1))
        e.close()
    whole.close()
