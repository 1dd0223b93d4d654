"""GPU (-m gpu): EVERY instance of the BASELINE.json configurations against the CPU oracle -- rewards and dones after every
step, all frames every few steps and at the end -- so that a mis-indexed tail workgroup or a wrong grid stride beyond
the first few thousand instances cannot hide (round 1 compared 6-7 sampled instances at these sizes).  Instance i is
seeded i; uniform random actions; same-step auto-reset on both sides."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# (label, env id, instances, steps, compare frames every, options)
CONFIGS = [
    ("C2", "MortarMayhem-Grid-v0", 65536, 130, 26, None),
    ("C3", "MysteryPath-v0", 32768, 120, 20, dict(max_steps=24)),          # short episodes: thousands of A* resets inside the run
    ("C4", "Endless-SearingSpotlights-v0", 16384, 240, 30, None),
    ("C5 per-GPU shard", "Endless-MortarMayhem-v0", 32768, 160, 32, None),
    # the ids that are in no BASELINE config, at the sizes bench.py measures them ("other_workloads"); Endless-MysteryPath long enough
    # for several episodes per instance: lazy segments, records ahead of time and the instances' own resets on every instance
    ("Endless-MysteryPath", "Endless-MysteryPath-v0", 32768, 300, 30, None),
    ("SearingSpotlights", "SearingSpotlights-v0", 16384, 200, 25, dict(max_steps=40)),      # (every instance is reset inside the raster launch)
    ("MysteryPath-Grid", "MysteryPath-Grid-v0", 32768, 120, 20, dict(max_steps=24)),
    ("MortarMayhem", "MortarMayhem-v0", 32768, 160, 32, None),
    ("MortarMayhemB-Grid", "MortarMayhemB-Grid-v0", 32768, 100, 25, None),  # (Dict observation: the visual part here, the one-hot vector in test_gpu_mortar_b.py)
    ("MortarMayhemB", "MortarMayhemB-v0", 16384, 100, 25, None),
]


@pytest.mark.parametrize("label,env_id,n,steps,every,options", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_every_instance(label, env_id, n, steps, every, options):
    import memory_gym_amd
    import oracle_lib
    import torch

    env = memory_gym_amd.make(env_id, num_envs=n, device=0)
    ref = oracle_lib.OracleBatch(env_id, n, options=options)
    seeds = np.arange(n, dtype=np.int64)
    want = np.zeros((n, 84, 84, 3), np.uint8)
    rew, done = np.zeros(n, np.float64), np.zeros(n, np.uint8)
    host = torch.empty((n, 84, 84, 3), dtype=torch.uint8).pin_memory()

    def frames_equal(where):
        host.copy_(obs)
        got = host.numpy()
        if not np.array_equal(got, want):
            bad = np.nonzero((got != want).reshape(n, -1).any(1))[0]
            raise AssertionError("%s %s: %d of %d frames differ %s; first instances %s" % (label, env_id, len(bad), n, where, bad[:10]))

    def visual(o):
        return o["visual_observation"] if isinstance(o, dict) else o

    obs, _ = env.reset(seed=seeds, options=options)
    obs = visual(obs)
    ref.reset(seeds, out=want)
    frames_equal("after reset")
    g = torch.Generator(device="cuda").manual_seed(17)
    disc = env.action_dim == 1
    n_done = 0
    for t in range(steps):
        a = torch.randint(0, 4 if disc else 3, (n,) if disc else (n, 2), device="cuda", generator=g, dtype=torch.int32)
        obs, r, d, _, _ = env.step(a)
        obs = visual(obs)
        check = (t + 1) % every == 0 or t == steps - 1
        ref.step(a.cpu().numpy(), autoreset=True, want_obs=check, out=(want, rew, done))
        dg = d.cpu().numpy()
        assert np.array_equal(dg, done.astype(bool)), "%s: done differs at step %d for instances %s" % (label, t, np.nonzero(dg != done.astype(bool))[0][:10])
        assert np.array_equal(env.reward64.cpu().numpy(), rew), "%s: reward differs at step %d for instances %s" % (
            label, t, np.nonzero(env.reward64.cpu().numpy() != rew)[0][:10])
        n_done += int(dg.sum())
        if check:
            frames_equal("at step %d" % t)
    for i in (0, 1, n // 2, min(14335, n - 1), min(14336, n - 1), n - 2, n - 1):  # around the persistent grid's size and at both ends
        assert np.array_equal(env.rng_words(i), ref.envs[i].rng_words()), "%s: RNG words of instance %d" % (label, i)
    assert n_done > 0
    env.check_errors()
    env.close()
    ref.close()
