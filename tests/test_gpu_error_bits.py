"""GPU (-m gpu): capacity limits are flagged, not silent.  Endless Mortar Mayhem's command list is unbounded in the reference
(endless_mortar_mayhem.py:316-318); the HIP path holds 512 commands per instance and, when an instance gets there, ends its
episode AND raises error bit 32 (include/memgym.h), which the Python step() turns into a RuntimeError at the latest one step
later.  512 commands are 131,328 correct tile visits; the test lowers the capacity (MEMGYM_EMM_CMD_CAP, read when the handle is
created) and plays perfectly until the list is full.  Until then everything equals the oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAB_LIB = os.path.join(ROOT, "endless-memory-gym_amd", "lib", "lab", "libmemgym_hip_lab.so")  # the hook exists in the -DMG_LAB build only (csrc/mg_lab.hpp)

WORKER = r'''
import os, sys
sys.path.insert(0, os.path.join(%(root)r, "endless-memory-gym_amd")); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, memory_gym_amd, oracle_lib
from test_gpu_mortar import expert_action_n
n, env_id = 4, "Endless-MortarMayhem-v0"
env = memory_gym_amd.make(env_id, num_envs=n, device=0)
ref = oracle_lib.OracleBatch(env_id, n)
seeds = np.arange(n, dtype=np.int64) + 40
env.reset(seed=seeds); ref.reset(seeds)
prng = np.random.Generator(np.random.PCG64(1))
raised = None
for t in range(4000):
    a = np.array([expert_action_n(env_id, ref.envs[i], prng, 2.0, 6) for i in range(n)], dtype=np.int32)
    full = [int(e.get("num_commands")) for e in ref.envs]
    try:
        obs, rew, done, _, info = env.step(a)
    except RuntimeError as e:
        raised = (t, str(e))
        break
    o2, r2, d2 = ref.step(a, autoreset=False, want_obs=True)
    d = done.cpu().numpy()
    over = [i for i in range(n) if full[i] == 5 and d[i] and not d2[i]]   # the list was full and would have grown: HIP ends the episode
    if over:
        continue_ok = True
        continue
    assert np.array_equal(d, d2.astype(bool)), "done differs at step %%d" %% t
    assert np.array_equal(rew.cpu().numpy(), r2.astype(np.float32))
    assert np.array_equal(obs.cpu().numpy(), o2), "frame differs at step %%d" %% t
    assert not d.any(), "a perfect player does not fail"
assert raised is not None, "no capacity error within 4000 steps (list lengths %%s)" %% [int(e.get("num_commands")) for e in ref.envs]
assert "command list" in raised[1] and "0x20" in raised[1], raised
print("OVERFLOW_FLAGGED at step", raised[0])
'''


def test_endless_mortar_mayhem_command_list_capacity_is_flagged():
    out = subprocess.run([sys.executable, "-c", WORKER % {"root": ROOT}], env=dict(os.environ, MEMGYM_EMM_CMD_CAP="5", MEMGYM_HIP_LIB=LAB_LIB), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "OVERFLOW_FLAGGED" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]
