"""GPU (-m gpu): frames of a masked reset drawn by the mask (round 6: raster_sparse_kernel, a workgroup per 32 instances) and terminal
observations copied instead of drawn again (mg_step with final_obs_dev) must be, byte for byte, what the dense persistent launches of
rounds 1-5 deliver: six env ids at sizes that are no multiple of anything, the gymnasium vector convention for 25-30 steps and masked
resets of 1 %, 50 % and 99.9 % of the instances, in two fresh processes of the lab build (MEMGYM_SPARSE_RASTER=0 / default).  Against
the ORACLE the new path runs in tests/test_gpu_vector_api.py (every terminal frame of seven ids) and wherever a test resets with a mask."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
LAB_LIB = os.path.join(os.path.dirname(HERE), "endless-memory-gym_amd", "lib", "lab", "libmemgym_hip_lab.so")


def digests(sparse, **more):
    env = dict(os.environ, MEMGYM_HIP_LIB=LAB_LIB, MEMGYM_SPARSE_RASTER="1" if sparse else "0", **more)
    r = subprocess.run([sys.executable, os.path.join(HERE, "sparse_worker.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok: all cases" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    if more:
        return [ln for ln in r.stdout.splitlines() if ln.startswith(("digest ", "own_resets ", "final_served "))]
    return [ln for ln in r.stdout.splitlines() if ln.startswith("digest ")]


def test_sparse_launches_equal_the_dense_ones():
    dense, sparse = digests(False), digests(True)
    assert len(dense) == 6 and dense == sparse, "\n".join(a + "\n" + b for a, b in zip(dense, sparse) if a != b)
    assert all(int(ln.rsplit("=", 1)[1]) > 0 for ln in dense)  # (episodes did finish in every case)


def test_emp_masked_resets_like_the_auto_reset_step():
    """Endless-MysteryPath-v0, 32,768 instances in the gymnasium vector convention: a masked reset(seed=None) resets an instance from the
    record generated ahead of time, like the auto-reset step (emp_masked_reset_kernel), or queues it for ONE lazily served segment --
    everything the caller sees (observations, terminal observations, rewards, dones, RNG words) equal to the run in which every masked
    instance is three cooperative paths of the queue server (lab MEMGYM_EMP_MASKED_FAST=0), and the fast path did reset instances itself."""
    # (MEMGYM_EMP_FINAL_FUSED=0: the vector convention through the generic path of mg_step, whose resets are masked resets)
    slow = digests(True, MEMGYM_SPARSE_CASES="emp_big", MEMGYM_EMP_MASKED_FAST="0", MEMGYM_EMP_FINAL_FUSED="0")
    fast = digests(True, MEMGYM_SPARSE_CASES="emp_big", MEMGYM_EMP_MASKED_FAST="1", MEMGYM_EMP_FINAL_FUSED="0")
    assert slow[0] == fast[0] and slow[0].startswith("digest Endless-MysteryPath-v0 32768"), slow[0] + "\n" + fast[0]
    assert int(fast[0].rsplit("=", 1)[1]) > 32768  # (finished episodes: several per instance)
    assert int(slow[1].split()[1]) == 0 and int(fast[1].split()[1]) > 1000, (slow[1], fast[1])


def test_mortar_one_launch_keeps_terminal_observations():
    """The mortar family's one-launch step keeps terminal observations itself (round 6: the frame workgroup of a finishing instance draws the
    terminal frame into final_obs_dev from a second descriptor, then the reset frame): everything the caller of the gymnasium vector convention
    sees -- observations, TERMINAL observations, rewards, dones, generator words -- equal to the generic path of mg_step (step without
    auto-reset, rows copied, masked reset; lab MEMGYM_MORTAR_FINAL_FUSED=0), all five mortar ids, MortarMayhem-Grid-v0 at 65,536 instances.
    (Against the oracle: tests/test_gpu_vector_api.py, terminal frame by terminal frame.)"""
    generic = digests(True, MEMGYM_SPARSE_CASES="mortar", MEMGYM_MORTAR_FINAL_FUSED="0")
    fused = digests(True, MEMGYM_SPARSE_CASES="mortar", MEMGYM_MORTAR_FINAL_FUSED="1")
    g = [ln for ln in generic if ln.startswith("digest ")]
    f = [ln for ln in fused if ln.startswith("digest ")]
    assert len(g) == 5 and g == f, "\n".join(a + "\n" + b for a, b in zip(g, f) if a != b)
    assert all(int(ln.rsplit("=", 1)[1]) > 0 for ln in g)


def test_spot_fused_launch_keeps_terminal_observations():
    """The spotlight family's fused raster / reset launch keeps terminal observations itself (round 6: the step kernel leaves a finishing
    instance's descriptor "as after any other step" -- the terminal frame's -- and the service workgroup draws it into final_obs_dev before it
    resets the instance): equal to the generic path of mg_step (lab MEMGYM_SPOT_FINAL_FUSED=0) in everything the caller sees incl. the terminal
    observations -- both variants, below and above the size at which the endless variant takes the fused launch on its own, and the border
    composer (black_background)."""
    generic = [ln for ln in digests(True, MEMGYM_SPARSE_CASES="spot", MEMGYM_SPOT_FINAL_FUSED="0") if ln.startswith("digest ")]
    fused = [ln for ln in digests(True, MEMGYM_SPARSE_CASES="spot", MEMGYM_SPOT_FINAL_FUSED="1") if ln.startswith("digest ")]
    assert len(generic) == 4 and generic == fused, "\n".join(a + "\n" + b for a, b in zip(generic, fused) if a != b)
    assert all(int(ln.rsplit("=", 1)[1]) > 0 for ln in generic)


def test_mystery_launches_keep_terminal_observations():
    """The finite Mystery Path ids keep terminal observations in the step's own two launches (round 6: mystery_step_kernel<false, true> leaves a
    finishing instance's terminal descriptor in io.tdesc, mystery_raster_paths_kernel<u8, true> draws it into final_obs_dev before the reset
    frame; the reset's path is generated beside the frames as in the auto-reset step): equal to the generic path of mg_step (lab
    MEMGYM_MYSTERY_FINAL_FUSED=0) in everything the caller sees incl. the terminal observations; episodes shortened with max_steps so that
    every instance finishes several times (a step in which ALL instances are truncated at once included)."""
    generic = [ln for ln in digests(True, MEMGYM_SPARSE_CASES="mystery", MEMGYM_MYSTERY_FINAL_FUSED="0") if ln.startswith("digest ")]
    fused = [ln for ln in digests(True, MEMGYM_SPARSE_CASES="mystery", MEMGYM_MYSTERY_FINAL_FUSED="1") if ln.startswith("digest ")]
    assert len(generic) == 4 and generic == fused, "\n".join(a + "\n" + b for a, b in zip(generic, fused) if a != b)
    assert all(int(ln.rsplit("=", 1)[1]) > 0 for ln in generic)


def test_emp_launches_keep_terminal_observations():
    """Endless-MysteryPath-v0 keeps terminal observations with the auto-reset step's own launches (round 6: emp_step_kernel<false, true> and the
    service waves of emp_raster_serve_kernel<u8, *, true> leave a finishing instance's terminal frame DESCRIPTOR in io.tdesc before anybody resets
    it; one sparse raster launch behind them draws those frames into final_obs_dev): equal to the generic path of mg_step (lab
    MEMGYM_EMP_FINAL_FUSED=0) in everything the caller sees incl. the terminal observations -- random agents and path followers, truncation of
    many instances in one step, both arrangements of launches."""
    generic = [ln for ln in digests(True, MEMGYM_SPARSE_CASES="emp", MEMGYM_EMP_FINAL_FUSED="0") if ln.startswith("digest ")]
    assert len(generic) == 4 and all(int(ln.rsplit("=", 1)[1]) > 0 for ln in generic)
    # (appended segments lazily -- the default: no step waits for a path, the service waves see resets only -- and, lab
    # MEMGYM_EMP_LAZY_APPEND=0, by the service waves within the step that needs them: those finish the instance's step and can meet its end)
    for lazy_append in ("1", "0"):
        out = digests(True, MEMGYM_SPARSE_CASES="emp", MEMGYM_EMP_FINAL_FUSED="1", MEMGYM_EMP_LAZY_APPEND=lazy_append)
        fused = [ln for ln in out if ln.startswith("digest ")]
        assert generic == fused, "\n".join(a + "\n" + b for a, b in zip(generic, fused) if a != b)
        served = [int(ln.split()[1]) for ln in out if ln.startswith("final_served ")]
        assert len(served) == 4 and (sum(served[1:]) > 0) == (lazy_append == "0"), served
