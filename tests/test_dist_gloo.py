"""world_size-2 gloo test (CPU) of the multi-GPU plumbing in memory_gym_amd/dist.py: shard ranges / seeds are a
partition that does not depend on the world size, and the rank-0 gather is byte-exact (also for ragged shards)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_dist_module():
    # import the helper without importing the package __init__ (which loads the HIP library)
    import importlib.util
    spec = importlib.util.spec_from_file_location("mg_dist", os.path.join(ROOT, "endless-memory-gym_amd", "memory_gym_amd", "dist.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _frames_for(lo, hi):
    """Deterministic fake observations for global instances [lo, hi): frame i is filled with a pattern of i."""
    idx = torch.arange(lo, hi, dtype=torch.int64)
    f = (idx[:, None] * 31 + torch.arange(84 * 84 * 3, dtype=torch.int64)[None, :] * 7) % 251
    return f.to(torch.uint8).reshape(-1, 84, 84, 3)


def _worker(rank, world, port, n_total, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _load_dist_module()
    lo, hi = m.shard_range(n_total, rank, world)
    seeds = m.shard_seeds(n_total, rank, world, base_seed=5)
    assert seeds.tolist() == list(range(lo + 5, hi + 5))
    obs = _frames_for(lo, hi)
    rew = torch.arange(lo, hi, dtype=torch.float32) * 0.5
    g_obs = m.gather_to_rank0(obs)
    g_rew = m.gather_to_rank0(rew)
    if rank == 0:
        assert g_obs.shape[0] == n_total and torch.equal(g_obs, _frames_for(0, n_total))
        assert torch.equal(g_rew, torch.arange(0, n_total, dtype=torch.float32) * 0.5)
        open(os.path.join(out_dir, "ok"), "w").write("ok")
    else:
        assert g_obs is None and g_rew is None
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_ranges_partition():
    m = _load_dist_module()
    for n in (1, 7, 64, 65536, 262144, 1000):
        for w in (1, 2, 3, 8):
            r = [m.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_gather_world2_even_and_ragged(tmp_path):
    for n_total in (8, 9):
        d = tmp_path / ("n%d" % n_total)
        d.mkdir()
        mp.spawn(_worker, args=(2, _free_port(), n_total, str(d)), nprocs=2, join=True)
        assert (d / "ok").exists()


def _reward_for(lo, hi, t):
    return torch.arange(lo, hi, dtype=torch.float32) * 0.125 + 0.1 * t


def _done_for(lo, hi, t):
    return ((torch.arange(lo, hi) + t) % 3 == 0).to(torch.uint8)


class _FakeEnv:
    """What ObsGatherer needs of a VecMemoryGym: `obs`, use_obs_buffer(), use_step_buffers(), step() writing the frames, rewards
    and dones of step t of this rank's instances into the buffers in use."""

    def __init__(self, lo, hi):
        self.lo, self.hi, self.t = lo, hi, 0
        self.obs = torch.zeros((hi - lo, 84, 84, 3), dtype=torch.uint8)
        self.reward = torch.zeros(hi - lo, dtype=torch.float32)
        self.done_u8 = torch.zeros(hi - lo, dtype=torch.uint8)

    def use_obs_buffer(self, t):
        self.obs = t

    def use_step_buffers(self, reward, done_u8):
        assert reward.dtype == torch.float32 and done_u8.dtype == torch.uint8 and reward.shape == done_u8.shape == (self.hi - self.lo,)
        self.reward, self.done_u8 = reward, done_u8

    def step(self, actions):
        self.obs.copy_((_frames_for(self.lo, self.hi).to(torch.int64) + 3 * self.t + int(actions)).remainder(256).to(torch.uint8))
        self.reward.copy_(_reward_for(self.lo, self.hi, self.t))
        self.done_u8.copy_(_done_for(self.lo, self.hi, self.t))
        self.t += 1
        return self.obs, self.reward, self.done_u8.view(torch.bool), None, {}


def _gatherer_worker(rank, world, port, n_total, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _load_dist_module()
    lo, hi = m.shard_range(n_total, rank, world)
    env = _FakeEnv(lo, hi)
    own = (env.obs, env.reward, env.done_u8)
    g = m.ObsGatherer(env)
    assert g.bufs[0] is not g.bufs[1]
    for t in range(7):
        obs, rew, done, *_ = g.step(t % 3)
        assert obs is g.bufs[t & 1]  # the environment alternates between the two buffers
        assert rew.data_ptr() == g.packed[t & 1][1].data_ptr() and done.data_ptr() == g.packed[t & 1][2].data_ptr()  # ... and the two packed reward / done buffers
        assert torch.equal(rew, _reward_for(lo, hi, t)) and torch.equal(done, _done_for(lo, hi, t).bool())
        if t >= 1:  # step t - 1's frames are still intact in the other buffer while their gather may be running
            want_prev = (_frames_for(lo, hi).to(torch.int64) + 3 * (t - 1) + (t - 1) % 3).remainder(256).to(torch.uint8)
            assert torch.equal(g.bufs[(t - 1) & 1], want_prev)
            assert torch.equal(g.packed[(t - 1) & 1][1], _reward_for(lo, hi, t - 1))
        if t % 2 == 0 or t == 6:  # join only some steps: the others are overtaken by the wait inside step t + 2
            got = g.gathered_step()
            if rank == 0:
                frames, rewards, dones = got
                full = torch.cat(frames, 0)
                want = (_frames_for(0, n_total).to(torch.int64) + 3 * t + t % 3).remainder(256).to(torch.uint8)
                assert torch.equal(full, want), "gathered frames of step %d differ" % t
                # "gather of obs (+reward, done)"
                assert torch.equal(torch.cat(rewards), _reward_for(0, n_total, t)), "gathered rewards of step %d differ" % t
                assert dones[0].dtype == torch.bool and torch.equal(torch.cat(dones), _done_for(0, n_total, t).bool()), "gathered dones of step %d differ" % t
                assert g.gathered() is frames
            else:
                assert got is None
    g.close()  # joins the gathers and gives the environment its own tensors back
    assert env.obs is own[0] and env.reward is own[1] and env.done_u8 is own[2]
    kept = g.packed[0][1].clone()
    env.step(1)
    assert torch.equal(g.packed[0][1], kept), "a step after close() must not touch the gatherer's buffers"
    if rank == 0:
        open(os.path.join(out_dir, "ok"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_double_buffered_gatherer_world2(tmp_path):
    for n_total in (8, 6):  # (6: 3 instances per rank -- the packed reward / done tensor is padded to 16 bytes)
        d = tmp_path / ("n%d" % n_total)
        d.mkdir()
        mp.spawn(_gatherer_worker, args=(2, _free_port(), n_total, str(d)), nprocs=2, join=True)
        assert (d / "ok").exists()


def test_packed_scalars_layout():
    m = _load_dist_module()
    for n in (1, 3, 16, 1000):
        flat, rew, done = m.packed_scalars(n, "cpu")
        assert flat.numel() % 16 == 0 and flat.numel() >= 5 * n and rew.shape == (n,) and done.shape == (n,)
        rew.copy_(torch.arange(n, dtype=torch.float32) + 0.5)
        done.copy_((torch.arange(n) % 2).to(torch.uint8))
        r2, d2 = m.unpack_scalars(flat.clone(), n)
        assert torch.equal(r2, rew) and d2.dtype == torch.bool and torch.equal(d2, done.bool())


def test_bench_self_launch_spawns_one_rank_per_gpu(tmp_path, monkeypatch):
    """`python bench.py --gpus N` without a launcher spawns N ranks with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set and
    lets only rank 0 write to stdout (the driver's SCALE runs call bench.py exactly like that)."""
    import subprocess
    import sys
    import types

    sys.path.insert(0, ROOT)
    import bench

    script = tmp_path / "fake_rank.py"
    script.write_text("import os\nprint('rank', os.environ['RANK'], os.environ['LOCAL_RANK'], os.environ['WORLD_SIZE'], "
                      "os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'], os.environ['HSA_ENABLE_IPC_MODE_LEGACY'], flush=True)\n")
    monkeypatch.setattr(bench.os.path, "abspath", lambda p: str(script))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "3"])
    import torch
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 3)
    seen = []
    real_popen = subprocess.Popen

    def spy(cmd, env=None, stdout=None, **kw):
        seen.append((env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"], stdout))
        return real_popen(cmd, env=env, stdout=subprocess.DEVNULL, **kw)

    monkeypatch.setattr(bench.subprocess, "Popen", spy)
    assert bench.self_launch(types.SimpleNamespace(gpus=3)) == 0
    assert [(r, lr, w) for r, lr, w, _ in seen] == [("0", "0", "3"), ("1", "1", "3"), ("2", "2", "3")]
    assert seen[0][3] is None and all(s[3] == subprocess.DEVNULL for s in seen[1:])
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    assert bench.self_launch(types.SimpleNamespace(gpus=3)) == 2  # fewer GPUs than ranks: refuse, do not time 1 GPU as 3
