"""GPU (-m gpu): the RCCL legs of BASELINE config 5 executed ONCE on the nccl backend -- a one-rank process group on the one
GPU a test box has (RCCL refuses two ranks on one device, so world size 1 is what a 1-GPU box can run; the 8-GPU curve is the
driver's):

  * memory_gym_amd.dist.gather_to_rank0 through RCCL == the local observations, byte for byte;
  * memory_gym_amd.dist.ObsGatherer (double-buffered: the gather of step t runs beside step t + 1) delivers, for every
    step, exactly the frames a plain env.step() sequence on a second handle produces;
  * `bench.py --gather rccl` end to end with a one-rank nccl group: one JSON line, gather_check true.
Each part runs in a subprocess: a process group is process-global state."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, os.path.join(%(root)r, "endless-memory-gym_amd"))
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
import memory_gym_amd
from memory_gym_amd.dist import gather_to_rank0, ObsGatherer
n = 512
env = memory_gym_amd.make("Endless-MortarMayhem-v0", num_envs=n, device=0)
ref = memory_gym_amd.make("Endless-MortarMayhem-v0", num_envs=n, device=0)
seeds = torch.arange(n, dtype=torch.int64, device="cuda")
obs, _ = env.reset(seed=seeds)
ref.reset(seed=seeds)
g = gather_to_rank0(obs)
torch.cuda.synchronize()
assert g.shape == obs.shape and torch.equal(g, obs), "gather_to_rank0 over RCCL differs from the local observations"
gen = torch.Generator(device="cuda").manual_seed(3)
gat = ObsGatherer(env)
prev = None
for t in range(40):
    a = torch.randint(0, 3, (n, 2), device="cuda", generator=gen, dtype=torch.int32)
    o, r, d, _, _ = gat.step(a)
    o2, r2, d2, _, _ = ref.step(a)
    want = o2.clone()
    got, grew, gdone = gat.gathered_step()
    torch.cuda.synchronize()
    assert len(got) == 1 and torch.equal(got[0], want), "gathered frames of step %%d differ" %% t
    assert torch.equal(r, r2) and torch.equal(d, d2)
    # the step's rewards and dones travel with the frames (BASELINE.md section 3, C5; one packed 5-B-per-instance collective)
    assert torch.equal(grew[0], r2) and gdone[0].dtype == torch.bool and torch.equal(gdone[0], d2), "gathered rewards / dones of step %%d differ" %% t
    if prev is not None:  # the other buffer still holds step t - 1's frames: the step did not write into it
        assert torch.equal(gat.bufs[(t - 1) & 1], prev)
    prev = want
gat.drain()
dist.destroy_process_group()
print("RCCL_WORLD1_OK")
'''


# RCCL reading an mg_obs_alloc range (HIP virtual memory: hipMemCreate pieces mapped into one reserved range): BASELINE config 5's
# per-GPU shard, 32,768 Endless-MortarMayhem instances = 694 MB of observations, the size at which the Python mirror takes its
# buffers from mg_obs_alloc (the 8-GPU run must not be the first time RCCL sees such memory)
WORKER_VMM = r'''
import os, sys
sys.path.insert(0, os.path.join(%(root)r, "endless-memory-gym_amd"))
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import memory_gym_amd
from memory_gym_amd.dist import gather_to_rank0, ObsGatherer
from memory_gym_amd.vec_env import is_balanced_buffer
n = 32768
env = memory_gym_amd.make("Endless-MortarMayhem-v0", num_envs=n, device=0)
info = env.obs_placement_info
print("PLACEMENT", info)
if info is None or info["pieces"] == 0:
    print("PLAIN_ALLOCATION")   # one zone only within reach on this box: nothing to test
    sys.exit(0)
assert is_balanced_buffer(env.obs)
obs, _ = env.reset(seed=torch.arange(n, dtype=torch.int64, device="cuda"))
g = gather_to_rank0(obs)
torch.cuda.synchronize()
assert torch.equal(g, obs), "gather_to_rank0 of a balanced (virtual-memory) buffer differs from the local observations"
gen = torch.Generator(device="cuda").manual_seed(3)
gat = ObsGatherer(env)
assert is_balanced_buffer(gat.bufs[1]), "the second buffer of the gatherer is balanced like the first"
for t in range(12):
    a = torch.randint(0, 3, (n, 2), device="cuda", generator=gen, dtype=torch.int32)
    o = gat.step(a)[0]
    got = gat.gathered()
    torch.cuda.synchronize()
    assert torch.equal(got[0], o), "gathered frames of step %%d differ from the step's observation buffer" %% t
gat.drain()
env.check_errors()
dist.destroy_process_group()
print("RCCL_VMM_OK")
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env():
    return dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")


def test_gather_through_rccl_equals_local_observations():
    out = subprocess.run([sys.executable, "-c", WORKER % {"root": ROOT}], env=_env(), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "RCCL_WORLD1_OK" in out.stdout, out.stderr[-3000:]


def test_rccl_reads_a_balanced_virtual_memory_buffer():
    out = subprocess.run([sys.executable, "-c", WORKER_VMM % {"root": ROOT}], env=_env(), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and ("RCCL_VMM_OK" in out.stdout or "PLAIN_ALLOCATION" in out.stdout), out.stdout[-1500:] + out.stderr[-3000:]
    if "PLAIN_ALLOCATION" in out.stdout:
        pytest.skip("mg_obs_alloc found one memory zone only on this box: " + out.stdout[-300:])


def test_bench_gather_rccl_one_rank():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gather", "rccl", "--env", "Endless-MortarMayhem-v0", "--envs-per-gpu", "8192",
                          "--steps", "16", "--warmup", "4", "--settle", "20", "--no-cpu-baseline", "--no-secondary", "--no-traffic", "--no-c1"],
                         env=_env(), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["value"] > 1e5
    assert "gather" in j["config"]["parallelism"] and j["gather_check"] is True
    assert j["timing"].startswith("host clock")
