"""GPU (-m gpu): EVERY instance of the BASELINE.json configurations against the CPU oracle -- rewards and dones after every
step, all frames every few steps and at the end -- so that a mis-indexed tail workgroup or a wrong grid stride beyond
the first few thousand instances cannot hide (round 1 compared 6-7 sampled instances at these sizes).  Instance i is
seeded i; same-step auto-reset on both sides.  Two policies: uniform random actions (`test_every_instance`), under which
episodes are short and shallow (three of four MortarMayhem-Grid episodes die at the first verification), and COMPETENT play
(`test_every_instance_competent`, round 6: the oracle's expert action for every instance's current state, a random one with
probability 0.1) -- ten-command successes, command lists of a dozen entries, path segments appended far beyond the initial
three, opened exits -- on the launch arrangements that only exist at these sizes.  What the competent runs reached is asserted
from the HIP side: the end-of-episode info the kernels wrote and mg_debug_counter scans of the device's state records."""
import os

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


def _lock_step(label, env_id, n, steps, every, options, eps=None, stats=None):
    """eps None: uniform random actions drawn on the device; else the oracle's expert actions with eps random ones.  `stats`:
    a dict the run fills from the HIP side's end-of-episode info (sums / maxima over the finished episodes) and its counters."""
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
            env.check_errors()  # a capacity error flagged by the kernels explains a mismatch better than pixels do
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
    a_host = np.empty((n,) if disc else (n, 2), np.int32)
    n_done = 0
    acc = {}
    for t in range(steps):
        if eps is None:
            a = torch.randint(0, 4 if disc else 3, (n,) if disc else (n, 2), device="cuda", generator=g, dtype=torch.int32)
            a_np = a.cpu().numpy()
        else:
            a_np = ref.expert_actions(eps, 4711, t, out=a_host)
            a = torch.from_numpy(a_np).to("cuda")
        obs, r, d, _, info = env.step(a)
        obs = visual(obs)
        check = (t + 1) % every == 0 or t == steps - 1
        ref.step(a_np, autoreset=True, want_obs=check, out=(want, rew, done))
        dg = d.cpu().numpy()
        assert np.array_equal(dg, done.astype(bool)), "%s: done differs at step %d for instances %s" % (label, t, np.nonzero(dg != done.astype(bool))[0][:10])
        assert np.array_equal(env.reward64.cpu().numpy(), rew), "%s: reward differs at step %d for instances %s" % (
            label, t, np.nonzero(env.reward64.cpu().numpy() != rew)[0][:10])
        n_done += int(dg.sum())
        if stats is not None and dg.any():  # what the finished episodes reached, as the kernels reported it
            for name in env.info_names:
                v = info[name][d].double()
                s, m = acc.get(name, (0.0, -np.inf))
                acc[name] = (s + float(v.sum()), max(m, float(v.max())))
        if check:
            frames_equal("at step %d" % t)
    for i in (0, 1, n // 2, min(14335, n - 1), min(14336, n - 1), n - 2, n - 1):  # around the persistent grid's size and at both ends
        assert np.array_equal(env.rng_words(i), ref.envs[i].rng_words()), "%s: RNG words of instance %d" % (label, i)
    env.check_errors()
    if stats is not None:
        stats["episodes"] = n_done
        for name, (s, m) in acc.items():
            stats["sum_" + name], stats["max_" + name] = s, m
        for name in stats.pop("counters", ()):
            stats[name] = env.debug_counter(name)
    env.close()
    ref.close()
    return n_done


@pytest.mark.parametrize("label,env_id,n,steps,every,options", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_every_instance(label, env_id, n, steps, every, options):
    assert _lock_step(label, env_id, n, steps, every, options) > 0


# Competent play at the sizes bench.py measures.  (label, env id, instances, steps, compare frames every,
# options, counters to read at the end, what the HIP side must report having reached)
COMPETENT = [
    ("C2", "MortarMayhem-Grid-v0", 65536, 170, 34, None, (),
     lambda s, n: s["sum_success"] > n // 4),                                           # full ten-command episodes
    ("C3", "MysteryPath-v0", 32768, 150, 30, None, (),
     lambda s, n: s["sum_success"] > n),                                                # goals reached along the A* path
    ("C4", "Endless-SearingSpotlights-v0", 16384, 260, 26, None, (),
     lambda s, n: s["max_coins_collected"] >= 12 and s["sum_coins_collected"] > 4 * n),
    ("C5 per-GPU shard", "Endless-MortarMayhem-v0", 32768, 560, 80, None, ("cmd_list_max",),
     lambda s, n: s["cmd_list_max"] >= 6),                                              # (a dozen entries: the slow variant below)
    ("Endless-MysteryPath", "Endless-MysteryPath-v0", 32768, 400, 40, None, ("emp_segments_sum", "emp_segments_max"),
     lambda s, n: s["emp_segments_sum"] > 6 * n and s["emp_segments_max"] >= 10),       # > 3 n segments appended beyond the initial three
    ("SearingSpotlights", "SearingSpotlights-v0", 16384, 200, 25, None, (),
     lambda s, n: s["sum_success"] > 2 * n),                                            # every coin, then the opened exit
    ("MysteryPath-Grid", "MysteryPath-Grid-v0", 32768, 120, 20, None, (),
     lambda s, n: s["sum_success"] > 4 * n),                                            # ~2,000 finished paths per step: the lane generator inside the raster launch
    ("MortarMayhem", "MortarMayhem-v0", 32768, 290, 58, None, (),
     lambda s, n: s["sum_success"] > n // 2),
]


@pytest.mark.parametrize("label,env_id,n,steps,every,options,counters,reached", COMPETENT, ids=[c[0] for c in COMPETENT])
def test_every_instance_competent(label, env_id, n, steps, every, options, counters, reached):
    stats = {"counters": counters}
    assert _lock_step(label, env_id, n, steps, every, options, eps=0.1, stats=stats) > 0 or env_id == "Endless-MysteryPath-v0"
    assert reached(stats, n), "%s: the competent run stayed shallow: %s" % (label, stats)


@pytest.mark.slow
def test_endless_mortar_lists_of_a_dozen():
    """Endless-MortarMayhem-v0 under competent play until command lists hold a dozen entries and more on over
    1 % of the instances (endless_mortar_mayhem.py:311-333: every completed list grows by one command and is executed again from the
    start -- 1,628 steps for the eleven lists before the twelfth with the default timings)."""
    if os.environ.get("MEMGYM_FAST"):
        pytest.skip("MEMGYM_FAST")
    n = 16384  # (the mortar family's one launch is the same arrangement at every size; the C5 shard size itself: the competent run above)
    stats = {"counters": ("cmd_list_max", "cmd_list_ge12")}
    _lock_step("C5 long", "Endless-MortarMayhem-v0", n, 1760, 220, None, eps=0.1, stats=stats)
    assert stats["cmd_list_ge12"] > n // 100 and stats["max_max_command_sequence"] >= 11, stats
