"""GPU (-m gpu): a capacity of this build ends ONE instance's episode, never the batch (missing #1 / next #3).  The
reference's lists grow without limit -- path segments, fall-off cells, the Endless-MortarMayhem command list, live spotlights --
here they hold 128 / 128 / 512 / 16 entries per instance.  The kernels end the episode of an instance that would need one more
(`done`, mg_info_buffers.capacity_dev) and raise the sticky error bit; `make(..., on_capacity="truncate")` reports the instance as
truncated (`truncated[i]`, info["capacity_exceeded"][i]) and the batch goes on, the default `"raise"` turns the bit into a
RuntimeError as before.  The lab build lowers the capacities so that a few hundred steps reach them (tests/capacity_worker.py);
every instance is in lock-step with the oracle up to the step that ends it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAB_LIB = os.path.join(ROOT, "endless-memory-gym_amd", "lib", "lab", "libmemgym_hip_lab.so")

CASES = {"emp_segments": dict(MEMGYM_EMP_SEG_CAP="6"), "emp_falloff": dict(MEMGYM_EMP_FALL_CAP="2"), "emm_commands": dict(MEMGYM_EMM_CMD_CAP="5"),
         "ess_slots": {},
         # the same lists at a capacity chosen through the public API (mg_set_capacity / make(capacity=...)): the episode goes on past the
         # lab hook's (or the default) length, in lock-step with the oracle, and ends where the NEW capacity is
         "emp_segments@capacity": dict(MEMGYM_EMP_SEG_CAP="6", MEMGYM_TEST_CAPACITY="path_segments=9"),
         "emm_commands@capacity": dict(MEMGYM_EMM_CMD_CAP="5", MEMGYM_TEST_CAPACITY="commands=7")}
BIT = {"emp_segments": 4, "emp_falloff": 8, "emm_commands": 32, "ess_slots": 1}


def _run(name, mode):
    env = dict(os.environ, MEMGYM_HIP_LIB=LAB_LIB, **CASES[name])
    name = name.split("@")[0]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "capacity_worker.py"), name, mode], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.parametrize("name", sorted(CASES))
def test_truncate_mode_ends_the_instance_not_the_batch(name):
    j = _run(name, "truncate")
    assert j["ended"] > 0, "no instance reached the capacity: %s" % j
    assert j["kinds"] & BIT[name.split("@")[0]], j
    assert j["still_in_lock_step"] >= 0


@pytest.mark.parametrize("name", ["emp_segments", "ess_slots"])
def test_raise_mode_still_raises(name):
    j = _run(name, "raise")
    assert j["raised_at"] is not None and ("0x%x" % BIT[name]) in j["message"] and "on_capacity='truncate'" in j["message"], j


def test_a_larger_segment_store_is_the_same_environment():
    """make(capacity={"path_segments": 1500}) changes where an instance's records lie (the stride of the segment store), nothing else:
    Endless-MysteryPath-v0 under a path-following policy in lock-step with the oracle; a checkpoint only loads into a handle of the same
    capacity; unknown names and values out of range are refused."""
    import numpy as np
    import memory_gym_amd
    import oracle_lib

    n, env_id = 384, "Endless-MysteryPath-v0"
    env = memory_gym_amd.make(env_id, num_envs=n, device=0, capacity={"path_segments": 1500})
    ref = oracle_lib.OracleBatch(env_id, n)
    seeds = np.arange(n, dtype=np.int64) + 3
    obs, _ = env.reset(seed=seeds)
    assert np.array_equal(obs.cpu().numpy(), ref.reset(seeds))
    for t in range(260):
        a = ref.expert_actions(0.05, 11, t)
        obs, rew, done, _, _ = env.step(a)
        o2, r2, d2 = ref.step(a, autoreset=True, want_obs=(t % 13 == 0))
        assert np.array_equal(done.cpu().numpy(), d2.astype(bool)) and np.array_equal(env.reward64.cpu().numpy(), r2), "step %d" % t
        if t % 13 == 0:
            assert np.array_equal(obs.cpu().numpy(), o2), "frames at step %d" % t
    assert env.debug_counter("emp_segments_max") > 3
    sd = env.state_dict()
    other = memory_gym_amd.make(env_id, num_envs=n, device=0)
    with pytest.raises(ValueError, match="capacity"):
        other.load_state_dict(sd)
    same = memory_gym_amd.make(env_id, num_envs=n, device=0, capacity={"path_segments": 1500})
    same.load_state_dict(sd)
    a = ref.expert_actions(0.05, 11, 260)
    o_a, _, _, _, _ = env.step(a)
    o_b, _, _, _, _ = same.step(a)
    assert np.array_equal(o_a.cpu().numpy(), o_b.cpu().numpy())
    with pytest.raises(ValueError, match="capacity"):
        memory_gym_amd.make(env_id, num_envs=8, device=0, capacity={"path_segments": 2})
    with pytest.raises(ValueError, match="capacity"):
        memory_gym_amd.make("MortarMayhem-Grid-v0", num_envs=8, device=0, capacity={"path_segments": 200})


def test_vector_front_end_reports_a_capacity_end_as_truncation():
    env = dict(os.environ, MEMGYM_HIP_LIB=LAB_LIB, MEMGYM_EMP_SEG_CAP="6")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "capacity_worker.py"), "vector"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    j = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["ended"] > 0, j
