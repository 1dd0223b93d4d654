"""The A/B switches of the library select whole launch arrangements (resets / path generation inside
the raster launch or not, full resets by lanes or by waves).  The defaults are what the rest of the suite runs; this runs
a lock-step parity check (HIP vs oracle, every frame) under the OTHER setting of each switch, in fresh processes.
The shipped library reads no switch (csrc/mg_lab.hpp); the workers load the -DMG_LAB build of the same sources."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
LAB_LIB = os.path.join(os.path.dirname(HERE), "endless-memory-gym_amd", "lib", "lab", "libmemgym_hip_lab.so")

CASES = [
    ("MEMGYM_EMP_FUSE", "0", "Endless-MysteryPath-v0", 160, 150),          # queue server as a launch of its own
    ("MEMGYM_EMP_RESET_LANES", "0", "Endless-MysteryPath-v0", 1024, 30),    # full reset one wave per instance (lanes: n >= 1,024)
    ("MEMGYM_EMP_BG_COOP", "0", "Endless-MysteryPath-v0", 640, 120),       # owed segments one per lane of frame workgroups (the default above ~20,000 instances)
    # the next episode's first segment ahead of time (the default above ~20,000 instances; it rides on the lane-per-path jobs) ...
    ("MEMGYM_EMP_BG_COOP=0 MEMGYM_SWITCH_WORKER_WANT=emp_own_resets MEMGYM_EMP_PRE", "1", "Endless-MysteryPath-v0", 640, 200),
    ("MEMGYM_EMP_BG_COOP=0 MEMGYM_EMP_PRE", "0", "Endless-MysteryPath-v0", 640, 120),  # ... and without it
    ("MEMGYM_MYSTERY_DEFER", "1", "MysteryPath-v0", 160, 150),              # reset paths inside the raster launch
    ("MEMGYM_MYSTERY_DEFER", "0", "MysteryPath-Grid-v0", 160, 150),         # ... and not, for the grid variant
    ("MEMGYM_SPOT_FUSE", "1", "Endless-SearingSpotlights-v0", 160, 200),    # resets inside the raster launch
    ("MEMGYM_SPOT_FUSE", "0", "SearingSpotlights-v0", 160, 200),            # ... and not, for the finite variant
    ("MEMGYM_SPOT_RESET_FALLBACK", "3", "Endless-SearingSpotlights-v0", 160, 200),  # every third instance: the reset's spotlights one after another (what a rejected draw falls back to)
    ("MEMGYM_SPOT_RESET_FALLBACK", "2", "SearingSpotlights-v0", 160, 200),
    ("MEMGYM_MORTAR_FUSE", "0", "MortarMayhem-Grid-v0", 300, 150),          # step and raster as two launches
    ("MEMGYM_MORTAR_FUSE", "0", "Endless-MortarMayhem-v0", 300, 150),
    ("MEMGYM_LAB_NONE", "1", "MortarMayhem-Grid-v0", 300, 60),             # the lab build itself, no switch set
]


# the same cases 1.5 x as long (rare paths need long runs) -- each one more fresh process, i.e. one more `import torch`
# (half a minute on a box with slow storage), so they are opt-in: MEMGYM_LONG_SWITCHES=1
LONG = [pytest.param(v, val, e, n, (st * 3 + 1) // 2 + 20, marks=pytest.mark.slow, id="long-%s=%s-%s" % (v.split()[-1], val, e))
        for v, val, e, n, st in CASES if st >= 70] if os.environ.get("MEMGYM_LONG_SWITCHES") else []


@pytest.mark.parametrize("var,value,env_id,n,steps", CASES + LONG)
def test_other_setting_is_bit_exact_too(var, value, env_id, n, steps):
    env = dict(os.environ, MEMGYM_HIP_LIB=LAB_LIB)
    *fixed, var = var.split()  # "A=1 B": A is set to 1, B to `value`
    for kv in fixed:
        env[kv.split("=")[0]] = kv.split("=")[1]
    env[var] = value
    r = subprocess.run([sys.executable, os.path.join(HERE, "switch_worker.py"), env_id, str(n), str(steps)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, "%s=%s, %s:\n%s\n%s" % (var, value, env_id, r.stdout[-2000:], r.stderr[-4000:])
    assert "ok:" in r.stdout
