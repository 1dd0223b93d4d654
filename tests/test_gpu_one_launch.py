"""GPU (-m gpu): the mortar family's one-launch step (csrc/mg_mortar.hip mortar_step_raster_kernel) never shows a stale frame.

The reference's step() returns the frame of THIS step, always (mortar_mayhem_grid.py:280-375).  In the one launch a frame's
workgroup waits for the descriptor the step's workgroups of the same launch publish; since round 4 a frame wave that waits
too long steps the instances itself (claim words), so the launch is correct whatever order the hardware dispatches its
workgroups in.  Checked here:
  (i)   with the step workgroups at the END of the grid (lab build, MEMGYM_LAB_LOGIC_LAST=1): every frame workgroup is resident
        before any step workgroup, i.e. every slot is stepped by a frame wave -- frames, rewards, dones, RNG streams bit-exact
        vs the oracle, and the rescue counter says the path was taken;
  (ii)  65,536 MortarMayhem-Grid instances stepped while a second stream runs back-to-back convolutions (the
        examples/rollout_with_policy.py situation): sampled instances vs the oracle, step for step;
  (iii) two handles on two streams stepping concurrently.
MEMGYM_SOAK_STEPS=20000 makes (ii) the long soak (run through gpurun, not in the driver's suite).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LAB_LIB = os.path.join(ROOT, "endless-memory-gym_amd", "lib", "lab", "libmemgym_hip_lab.so")

WORKER = r'''
import sys
sys.path[:0] = [%(here)r, %(pkg)r]
import numpy as np
from gpu_parity import run_parity
import memory_gym_amd
env_id, n, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
# run_parity closes its handle; count the rescues on a handle of our own first
env = memory_gym_amd.make(env_id, num_envs=n, device=0)
env.reset(seed=np.arange(n, dtype=np.int64))
import torch
for t in range(5):
    env.step(torch.zeros((n, env.action_dim), dtype=torch.int32, device="cuda").squeeze(-1))
env.check_errors()
print("RESCUES", env.debug_counter("one_launch_rescues"))
env.close()
done = run_parity(env_id, None, n=n, steps=steps, check_every=1)
print("ok:", env_id, n, steps, done)
'''


@pytest.mark.parametrize("env_id,n,steps", [("MortarMayhem-Grid-v0", 4096, 40), ("Endless-MortarMayhem-v0", 4133, 30), ("MortarMayhem-v0", 2500, 30)])
def test_step_workgroups_dispatched_last(env_id, n, steps):
    env = dict(os.environ, MEMGYM_HIP_LIB=LAB_LIB, MEMGYM_LAB_LOGIC_LAST="1")
    r = subprocess.run([sys.executable, "-c", WORKER % {"here": HERE, "pkg": os.path.join(ROOT, "endless-memory-gym_amd")}, env_id, str(n), str(steps)],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok:" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    rescues = int(r.stdout.split("RESCUES")[1].split()[0])
    # more frames than the chip holds workgroups (1,792): the step workgroups cannot be resident before frame waves wait
    assert rescues > 0, "the frame waves never had to step a slot themselves: the test did not exercise the path"


def _sampled_oracle(env_id, idx):
    import oracle_lib
    ref = oracle_lib.OracleBatch(env_id, len(idx))
    ref.reset(np.asarray(idx, dtype=np.int64))  # instance i of the handle is seeded i
    return ref


def _lockstep(envs, refs, idxs, steps, frames_every, between=None):
    """Step every handle of `envs` (each on its own stream when several) with its own action stream; compare the sampled
    instances with their oracles after every step."""
    import torch

    n = envs[0].num_envs
    gens = [torch.Generator(device="cuda").manual_seed(11 + k) for k in range(len(envs))]
    streams = [torch.cuda.Stream() for _ in envs] if len(envs) > 1 else [torch.cuda.current_stream()]
    sel = [torch.as_tensor(ix, device="cuda") for ix in idxs]
    for t in range(steps):
        pending, outs = [], []
        for k, env in enumerate(envs):  # enqueue every handle's step first (no host synchronisation in between) ...
            with torch.cuda.stream(streams[k]):
                a = torch.randint(0, 4 if env.action_dim == 1 else 3, (n, env.action_dim), generator=gens[k], device="cuda", dtype=torch.int32)
                obs, rew, done, _, _ = env.step(a.squeeze(-1) if env.action_dim == 1 else a)
                pending.append((a, obs, rew, done))
            if between is not None:
                between()
        for k, (a, obs, rew, done) in enumerate(pending):  # ... then fetch the sampled instances
            with torch.cuda.stream(streams[k]):
                outs.append((a[sel[k]].cpu().numpy(), rew[sel[k]].cpu().numpy(), done[sel[k]].cpu().numpy(),
                             obs[sel[k]].cpu().numpy() if t % frames_every == 0 or t == steps - 1 else None))
        for k, (a, rew, done, frames) in enumerate(outs):
            o2, r2, d2 = refs[k].step(a[:, 0] if envs[k].action_dim == 1 else a, autoreset=True, want_obs=frames is not None)
            assert np.array_equal(done, d2.astype(bool)), "handle %d: done differs at step %d" % (k, t)
            assert np.array_equal(rew, r2.astype(np.float32)), "handle %d: reward differs at step %d" % (k, t)
            if frames is not None:
                bad = np.nonzero((frames != o2).reshape(len(frames), -1).any(1))[0]
                assert len(bad) == 0, "handle %d: frames of sampled instances %s differ at step %d" % (k, [int(idxs[k][b]) for b in bad[:8]], t)
    for env in envs:
        env.check_errors()


def test_under_a_concurrent_stream():
    import memory_gym_amd
    import torch

    steps = int(os.environ.get("MEMGYM_SOAK_STEPS", "800"))
    n = 65536
    env = memory_gym_amd.make("MortarMayhem-Grid-v0", num_envs=n, device=0)
    idx = np.unique(np.concatenate([np.arange(0, 64), np.arange(n - 64, n), np.random.Generator(np.random.PCG64(5)).integers(0, n, 64)]))
    ref = _sampled_oracle("MortarMayhem-Grid-v0", idx)
    obs, _ = env.reset(seed=np.arange(n, dtype=np.int64))
    assert np.array_equal(obs[torch.as_tensor(idx, device="cuda")].cpu().numpy(), ref.reset(idx.astype(np.int64)))
    # the policy side of a rollout: a CNN over the previous observations on a stream of its own, never joined with the env's
    side = torch.cuda.Stream()
    conv = torch.nn.Conv2d(3, 32, 8, stride=4).cuda().half()
    x = torch.randn(2048, 3, 84, 84, device="cuda", dtype=torch.half)

    def policy_work():
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):
                conv(x)

    _lockstep([env], [ref], [idx], steps, frames_every=50, between=policy_work)
    torch.cuda.synchronize()
    assert env.debug_counter("one_launch_rescues") >= 0  # (how many is the hardware's business; the frames above are what counts)
    env.close()
    ref.close()


def test_two_handles_on_two_streams():
    import memory_gym_amd
    import torch

    n = 16384
    ids = ["MortarMayhem-Grid-v0", "Endless-MortarMayhem-v0"]
    envs = [memory_gym_amd.make(i, num_envs=n, device=0) for i in ids]
    idx = np.unique(np.concatenate([np.arange(0, 32), np.arange(n - 32, n)]))
    refs = [_sampled_oracle(i, idx) for i in ids]
    for env in envs:
        env.reset(seed=np.arange(n, dtype=np.int64))
    torch.cuda.synchronize()
    _lockstep(envs, refs, [idx, idx], 400, frames_every=40)
    torch.cuda.synchronize()
    for e in envs:
        e.close()
    for r in refs:
        r.close()
