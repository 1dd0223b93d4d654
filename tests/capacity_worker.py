"""Worker of tests/test_gpu_capacity.py (one fresh process per scenario: the lab build of the library is chosen through MEMGYM_HIP_LIB
before the package loads, its capacity hooks through MEMGYM_EMP_SEG_CAP / MEMGYM_EMP_FALL_CAP / MEMGYM_EMM_CMD_CAP).

    python capacity_worker.py <scenario> <truncate|raise>

Lock-step against the oracle under the oracle's expert policy (eps as the scenario says).  An instance that the HIP path ends on a
capacity (info["capacity_exceeded"]) is where this build and the reference part ways BY DESIGN -- the reference's list would have grown
-- so it leaves the comparison there (the oracle's copy plays on with the longer list); every instance up to that step, and every
instance that never gets there, must equal the oracle."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "endless-memory-gym_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import memory_gym_amd  # noqa: E402
import oracle_lib  # noqa: E402

SCENARIOS = {
    # name: (env id, options, instances, steps, eps, oracle field that must stand at the capacity when an instance is ended, its value)
    "emp_segments": ("Endless-MysteryPath-v0", None, 192, 420, 0.05, "num_seg", int(os.environ.get("MEMGYM_EMP_SEG_CAP", "128"))),
    "emp_falloff": ("Endless-MysteryPath-v0", dict(stamina_level=60), 192, 700, 0.35, "n_falloff", int(os.environ.get("MEMGYM_EMP_FALL_CAP", "128"))),
    "emm_commands": ("Endless-MortarMayhem-v0", dict(explosion_delay=[6], explosion_duration=[2], command_show_duration=[1], command_show_delay=[0]),
                     96, 700, 0.0, "num_commands", int(os.environ.get("MEMGYM_EMM_CMD_CAP", "512"))),
    # spotlights that live 25 to 1,000 steps, one more every 4 steps, an agent that cannot die: more than 16 alive within ~100 steps
    "ess_slots": ("Endless-SearingSpotlights-v0", dict(spot_min_speed=0.001, spot_max_speed=0.04, spawn_interval=4, agent_health=100000, steps_per_coin=100000,
                                                      initial_spawns=3), 96, 260, 0.1, "n_spots", 16),
}


def vector_front_end():
    """gymnasium's vector convention: an episode this build ended on a capacity is TRUNCATED, not terminated (like a time limit); its
    final_observation / final_info are there as for any other end."""
    from memory_gym_amd.vector import GymnasiumVectorEnv
    n, env_id = 96, "Endless-MysteryPath-v0"
    envs = GymnasiumVectorEnv(env_id, n, on_capacity="truncate")
    ref = oracle_lib.OracleBatch(env_id, n)   # (only the policy's eyes: the oracle's copies play on where this build truncates)
    seeds = 7
    obs, info = envs.reset(seed=seeds)
    ref.reset(np.arange(n, dtype=np.int64) + seeds)
    ended = 0
    follow = torch.tensor([1.0, 2.0, 3.0], device="cuda")
    gt = info["ground_truth"]
    for t in range(420):
        a = (gt.float() @ follow).to(torch.int32)  # a perfect follower, from the environment's own ground truth
        obs, rew, term, trunc, infos = envs.step(a)
        gt = infos["ground_truth"]
        capx = infos["capacity_exceeded"]
        assert torch.equal(capx, trunc) and not (term & trunc).any(), "terminated and truncated exclude each other"
        if trunc.any():
            assert infos["_final_info"][trunc].all() and infos["_final_observation"][trunc].all()
            assert (infos["final_info"]["length"][trunc] > 100).all(), "a capacity end comes after a long walk"
            ended += int(trunc.sum())
    print(json.dumps({"scenario": "vector", "ended": ended}))


def main():
    if sys.argv[1] == "vector":
        return vector_front_end()
    name, mode = sys.argv[1], sys.argv[2]
    env_id, options, n, steps, eps, field, cap = SCENARIOS[name]
    capacity = None
    if os.environ.get("MEMGYM_TEST_CAPACITY"):  # "name=value": mg_set_capacity through make(capacity=...); the scenario's list is then that long
        what, value = os.environ["MEMGYM_TEST_CAPACITY"].split("=")
        capacity, cap = {what: int(value)}, int(value)
    env = memory_gym_amd.make(env_id, num_envs=n, device=0, on_capacity=mode, capacity=capacity)
    if capacity:
        assert env.capacity == capacity and memory_gym_amd._native.LIB.mg_capacity(env._h, list(capacity)[0].encode()) == cap
    ref = oracle_lib.OracleBatch(env_id, n, options=options)
    seeds = np.arange(n, dtype=np.int64) + 7
    obs, _ = env.reset(seed=seeds, options=options)
    assert np.array_equal(obs.cpu().numpy(), ref.reset(seeds))
    alive = np.ones(n, bool)   # still in lock-step with the oracle
    ended = []                 # (step, instance, oracle's field before the step)
    raised = None
    for t in range(steps):
        a = ref.expert_actions(eps, 99, t)
        before = ref.get_all(field)
        try:
            obs, rew, done, trunc, info = env.step(a)
        except RuntimeError as e:
            raised = (t, str(e))
            break
        o2, r2, d2 = ref.step(a, autoreset=True, want_obs=True)
        if mode == "raise":  # (the kernels end the instance in the step that reaches the capacity; step() raises at the latest one call later)
            continue
        d, r = done.cpu().numpy(), env.reward64.cpu().numpy()
        if mode == "truncate":
            capx = info["capacity_exceeded"].cpu().numpy()
            assert np.array_equal(capx, trunc.cpu().numpy()) and not (capx & ~d).any(), "a capacity end is a done and a truncation"
            for i in np.nonzero(capx & alive)[0]:
                assert before[i] >= cap - (2 if field == "num_seg" else 0), "instance %d was ended at step %d with %s = %s (capacity %d)" % (i, t, field, before[i], cap)
                # (the oracle's episode goes on with the longer list -- unless the same step ended it anyway, e.g. the list's last command failed)
                ended.append((t, int(i), float(before[i])))
            alive &= ~capx
        assert np.array_equal(d[alive], d2.astype(bool)[alive]), "done differs at step %d for %s" % (t, np.nonzero(alive & (d != d2.astype(bool)))[0][:8])
        assert np.array_equal(r[alive], r2[alive]), "reward differs at step %d" % t
        got = obs.cpu().numpy()
        same = (got == o2).reshape(n, -1).all(1)
        assert same[alive].all(), "frame differs at step %d for %s" % (t, np.nonzero(alive & ~same)[0][:8])
    if mode == "truncate":
        assert raised is None, raised
        env.check_errors()  # capacity bits are not errors in this mode
        for t in range(50):  # the batch goes on, ended instances included (their new episodes are ordinary ones)
            env.step(ref.expert_actions(1.0, 5, t))
        env.check_errors()
        print(json.dumps({"scenario": name, "ended": len(ended), "first": ended[:3], "still_in_lock_step": int(alive.sum()), "kinds": int(env.capacity_events)}))
    else:
        print(json.dumps({"scenario": name, "raised_at": None if raised is None else raised[0], "message": None if raised is None else raised[1]}))


if __name__ == "__main__":
    main()
