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
         "ess_slots": {}}
BIT = {"emp_segments": 4, "emp_falloff": 8, "emm_commands": 32, "ess_slots": 1}


def _run(name, mode):
    env = dict(os.environ, MEMGYM_HIP_LIB=LAB_LIB, **CASES[name])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "capacity_worker.py"), name, mode], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.parametrize("name", sorted(CASES))
def test_truncate_mode_ends_the_instance_not_the_batch(name):
    j = _run(name, "truncate")
    assert j["ended"] > 0, "no instance reached the capacity: %s" % j
    assert j["kinds"] & BIT[name], j
    assert j["still_in_lock_step"] >= 0


@pytest.mark.parametrize("name", ["emp_segments", "ess_slots"])
def test_raise_mode_still_raises(name):
    j = _run(name, "raise")
    assert j["raised_at"] is not None and ("0x%x" % BIT[name]) in j["message"] and "on_capacity='truncate'" in j["message"], j
