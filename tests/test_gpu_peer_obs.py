"""GPU (-m gpu): the peer-mapped observation gather (memory_gym_amd.dist.PeerObsBuffer).  Two processes -- on the 1-GPU
box both on cuda:0, gloo for control -- rasterise their shards straight into rank 0's observation tensor through HIP
IPC; rank 0's tensor must equal what one process computes for all instances (world-size invariance, tests/test_dist_gloo.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("env_id", ["Endless-MortarMayhem-v0", "MysteryPath-Grid-v0"])
def test_two_ranks_write_into_rank0_memory(env_id, tmp_path):
    import torch

    import memory_gym_amd

    n_total, steps, out = 96, 20, str(tmp_path / "frames.pt")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29581", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "peer_obs_worker.py"), env_id, str(n_total), str(steps), out],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\\n".join(logs)
    got = torch.load(out)

    env = memory_gym_amd.make(env_id, num_envs=n_total, device=0)
    obs, _ = env.reset(seed=0)
    want = [obs.cpu().clone()]
    want_r, want_d = [], []
    g = torch.Generator(device="cuda").manual_seed(5)
    for t in range(steps):
        a = torch.randint(0, 4 if env.action_dim == 1 else 3, (n_total,) if env.action_dim == 1 else (n_total, 2), device="cuda", generator=g, dtype=torch.int32)
        o, r, d, _, _ = env.step(a)
        want.append(o.cpu().clone())
        want_r.append(r.cpu().clone())
        want_d.append(d.cpu().clone())
    env.close()
    assert torch.equal(got, torch.stack(want))
    # rank 0 also holds every rank's rewards and dones of each step ("obs (+reward, done)")
    got_r, got_d = torch.load(out + ".scalars")
    assert torch.equal(got_r, torch.stack(want_r)) and got_d.dtype == torch.bool and torch.equal(got_d, torch.stack(want_d))
