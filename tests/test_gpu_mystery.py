"""GPU parity (-m gpu) for the Mystery Path family: HIP path through the C ABI vs the CPU oracle."""
import numpy as np
import pytest

from gpu_parity import check_terminal_info, run_parity

pytestmark = pytest.mark.gpu


def _toward(d):
    return 0 if d == 0 else (1 if d < 0 else 2)


def path_follower(e, prng):
    """MysteryPath-v0: walk along the oracle's path (end-first list) with occasional mistakes."""
    if prng.random() > 0.93:
        return [int(prng.integers(0, 3)), int(prng.integers(0, 3))]
    path = e.get_list("path").reshape(-1, 2)
    pos = (e.get("nx"), e.get("ny"))
    idx = None
    for k, (x, y) in enumerate(path):
        if (x, y) == pos:
            idx = k
            break
    if idx is None or idx == 0:
        return [0, 0]
    nx, ny = path[idx - 1]
    return [_toward(nx * 12 + 6 - e.get("ax")), _toward(ny * 12 + 6 - e.get("ay"))]


def endless_follower(e, prng):
    if prng.random() > 0.96:
        return [int(prng.integers(0, 4)), 0]
    path = e.get_list("path").reshape(-1, 2)
    cur = (e.get("cur_nx"), e.get("cur_ny"))
    k = next((j for j, (x, y) in enumerate(path) if (x, y) == cur), None)
    if k is None or k + 1 >= len(path):
        return [0, 0]
    nx, ny = path[k + 1]
    dx, dy = nx * 12 + 6 - e.get("ax"), ny * 12 + 6 - e.get("ay")
    if dy < 0:
        return [2, 0]
    if dy > 0:
        return [3, 0]
    if dx > 0:
        return [1, 0]
    return [0, 0]


def grid_follower(e, prng):
    """MysteryPath-Grid-v0: rotate towards / step onto the next path tile, with occasional mistakes."""
    if prng.random() > 0.93:
        return [int(prng.integers(0, 4)), 0]
    path = e.get_list("path").reshape(-1, 2)
    pos = (e.get("nx"), e.get("ny"))
    idx = next((k for k, (x, y) in enumerate(path) if (x, y) == pos), None)
    if idx is None or idx == 0:
        return [0, 0]
    dx, dy = path[idx - 1][0] - pos[0], path[idx - 1][1] - pos[1]
    want = 270 if dx > 0 else (90 if dx < 0 else (0 if dy < 0 else 180))
    rot = e.get("arot")
    if rot == want:
        return [3, 0]
    return [1 if (want - rot) % 360 in (90, 180) else 2, 0]


GRID_OPTS = [
    None,
    dict(max_steps=40, cardinal_origin_choice=[0, 3], show_origin=True, show_goal=True, reward_fall_off=-0.1, reward_step=-0.01,
         reward_goal=2.0, reward_path_progress=0.1),
]
MP_OPTS = [
    None,
    dict(max_steps=64, cardinal_origin_choice=[2], show_origin=True, show_goal=True, reward_fall_off=-0.1, reward_step=-0.01,
         reward_goal=2.0),
    dict(cardinal_origin_choice=[1, 3], visual_feedback=False),
]
EMP_OPTS = [
    None,
    dict(max_steps=300, stamina_level=12, reward_fall_off=-0.1, reward_path_progress_dense=0.05, reward_step=-0.001,
         camera_offset_scale=3.0),
    dict(show_stamina=True, show_past_path=False, visual_feedback=False),
    dict(show_background=True),                                      # icy columns scrolling with the agent
    dict(show_background=True, agent_speed=5.0, show_stamina=True),  # 5 px per step: phases 0, 5, 10, then 15 -> 0
    dict(show_background=True, agent_speed=2.0, camera_offset_scale=3.0, max_steps=200),
]


@pytest.mark.parametrize("opt_idx", range(len(MP_OPTS)))
def test_finite_parity(opt_idx):
    n_done = run_parity("MysteryPath-v0", MP_OPTS[opt_idx], n=160, steps=540 if opt_idx != 1 else 200, policy=path_follower, n_policy=64)
    assert n_done > 0


@pytest.mark.parametrize("opt_idx", range(len(GRID_OPTS)))
def test_grid_parity(opt_idx):
    n_done = run_parity("MysteryPath-Grid-v0", GRID_OPTS[opt_idx], n=160, steps=300, policy=grid_follower, n_policy=64)
    assert n_done > 0


@pytest.mark.parametrize("opt_idx", range(len(EMP_OPTS)))
def test_endless_parity(opt_idx):
    n_done = run_parity("Endless-MysteryPath-v0", EMP_OPTS[opt_idx], n=160, steps=260, policy=endless_follower, n_policy=64)
    assert n_done > 0


def test_terminal_info():
    assert check_terminal_info("MysteryPath-v0", steps=520) > 0
    assert check_terminal_info("Endless-MysteryPath-v0", steps=200) > 0
    assert check_terminal_info("MysteryPath-Grid-v0", steps=140) > 0


def test_full_size_sample():
    """BASELINE config C3 size (32,768 instances): a sample of instances must match single-instance oracles,
    including the A* path generation at reset (seed = instance index)."""
    import memory_gym_amd
    import oracle_lib
    import torch

    n = 32768
    env = memory_gym_amd.make("MysteryPath-v0", num_envs=n, device=0)
    obs, _ = env.reset(seed=0, options=dict(max_steps=40))
    sample = [0, 1, 255, 4095, 16384, 32767]
    refs = {i: oracle_lib.OracleEnv("MysteryPath-v0") for i in sample}
    first = obs[sample].cpu().numpy()
    for k, i in enumerate(sample):
        want = refs[i].reset(i, options=dict(max_steps=40))
        assert np.array_equal(first[k], want), "reset frame of instance %d differs in %d bytes (frame all zero: %s; placement %s)" % (
            i, int((first[k] != want).sum()), not first[k].any(), env.obs_placement_info)
    g = torch.Generator(device="cuda").manual_seed(0)
    for t in range(130):
        a = torch.randint(0, 3, (n, 2), device="cuda", generator=g, dtype=torch.int32)
        obs, rew, done, _, _ = env.step(a)
        ac = a[sample].cpu().numpy()
        got = obs[sample].cpu().numpy()
        for k, i in enumerate(sample):
            o, r, d = refs[i].step(ac[k])
            if d:
                o = refs[i].reset(None)
            assert np.array_equal(got[k], o), "instance %d differs at step %d" % (i, t)
    for i in sample:
        assert np.array_equal(env.rng_words(i), refs[i].rng_words())
    env.close()


def test_endless_full_size_sample():
    """32,768 Endless-MysteryPath instances -- the launch size at which a reset's segments are owed (lazy), the owed ones are
    generated one per lane beside the frames, and the NEXT episode's first segment is generated ahead of time so that a finishing
    instance is reset by its own step (EMP_PRE) -- against single-instance oracles for a sample of instances:
    frames, rewards, dones and ground truth at every step, the generator's words at the end and after a checkpoint round trip."""
    import memory_gym_amd
    import oracle_lib
    import torch

    n = 32768
    env = memory_gym_amd.make("Endless-MysteryPath-v0", num_envs=n, device=0)
    obs, _ = env.reset(seed=0)
    sample = [0, 1, 63, 64, 255, 4095, 16384, 20000, 32767]
    refs = {i: oracle_lib.OracleEnv("Endless-MysteryPath-v0") for i in sample}
    first = obs[sample].cpu().numpy()
    for k, i in enumerate(sample):
        assert np.array_equal(first[k], refs[i].reset(i)), "reset frame of instance %d differs" % i
    g = torch.Generator(device="cuda").manual_seed(0)
    episodes = 0

    far_seen = [0]
    followers = [i for k, i in enumerate(sample) if k % 2 == 1]  # these walk along their paths: new segments fall due, records ahead of time are dropped

    class _Never:  # (endless_follower's share of random actions: none here)
        def random(self):
            return 0.0

    _never = _Never()
    fidx = torch.tensor(followers, device="cuda")

    def run(steps, t0):
        nonlocal episodes
        for t in range(steps):
            a = torch.randint(0, 4, (n,), device="cuda", generator=g, dtype=torch.int32)
            if (t0 + t) % 200 < 130:  # (130 faultless steps = three to four segments, then random actions until the instance has been reset)
                a[fidx] = torch.tensor([endless_follower(refs[i], _never)[0] for i in followers], device="cuda", dtype=torch.int32)
            obs, rew, done, _, info = env.step(a)
            ac = a[sample].cpu().numpy()
            got, gr, gd = obs[sample].cpu().numpy(), rew[sample].cpu().numpy(), done[sample].cpu().numpy()
            ggt = info["ground_truth"][sample].cpu().numpy()
            for k, i in enumerate(sample):
                o, r, d = refs[i].step(int(ac[k]))
                assert bool(gd[k]) == bool(d) and gr[k] == np.float32(r), "instance %d: reward / done differ at step %d" % (i, t0 + t)
                if i in followers:
                    far_seen[0] = max(far_seen[0], int(refs[i].get("num_seg") or 0))
                if d:
                    o = refs[i].reset(None)
                    episodes += 1
                assert np.array_equal(got[k], o), "instance %d differs at step %d" % (i, t0 + t)
                assert np.array_equal(ggt[k], np.asarray(refs[i].gt(), dtype=np.float32)), "instance %d: ground truth differs at step %d" % (i, t0 + t)

    run(220, 0)
    assert episodes >= 20
    assert far_seen[0] >= 4, "no follower reached a fourth segment (appended when the agent enters the last but one)"
    assert env.debug_counter("emp_ahead_records") > n  # more than one episode per instance began with a record made ahead of time
    for i in sample:  # (looks at the state: what is owed is generated first; records ahead of time stay)
        assert np.array_equal(env.rng_words(i), refs[i].rng_words()), "RNG stream of instance %d diverged" % i
    sd = env.state_dict()
    env.close()
    env = memory_gym_amd.make("Endless-MysteryPath-v0", num_envs=n, device=0)
    env.load_state_dict(sd)
    run(60, 220)
    for i in sample:
        assert np.array_equal(env.rng_words(i), refs[i].rng_words()), "RNG stream of instance %d diverged after the checkpoint" % i
    env.check_errors()
    env.close()


@pytest.mark.slow
@pytest.mark.parametrize("env_id,opts,policy,steps", [("MysteryPath-v0", MP_OPTS[0], path_follower, 1100), ("MysteryPath-Grid-v0", GRID_OPTS[0], grid_follower, 600),
                                                      ("Endless-MysteryPath-v0", EMP_OPTS[0], endless_follower, 700), ("Endless-MysteryPath-v0", EMP_OPTS[1], endless_follower, 700)])
def test_long_runs(env_id, opts, policy, steps):
    """(marked slow: MEMGYM_FAST=1 leaves it out) one long lock-step run per variant, every frame compared (rare paths -- a long episode, owed and new
    segments falling into one step -- need many steps to occur)."""
    assert run_parity(env_id, opts, n=160, steps=steps, policy=policy, n_policy=64) > 0
