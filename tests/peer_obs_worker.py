"""Worker of tests/test_gpu_peer_obs.py (one process per rank, gloo for control, both ranks on cuda:0 of the 1-GPU box):
every rank rasterises its shard straight into rank 0's observation tensor (memory_gym_amd.dist.PeerObsBuffer)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "endless-memory-gym_amd"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import memory_gym_amd  # noqa: E402
from memory_gym_amd.dist import PeerObsBuffer, shard_range, shard_seeds  # noqa: E402


def main():
    env_id, n_total, steps, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    buf = PeerObsBuffer(n_total, device="cuda:0")
    lo, hi = shard_range(n_total, rank, world)
    env = memory_gym_amd.make(env_id, num_envs=hi - lo, device=0, obs_buffer=buf.local)
    env.reset(seed=shard_seeds(n_total, rank, world, base_seed=0, device="cuda:0"))
    buf.fence()
    buf.bind_scalars(env)  # the step's rewards / dones ride on the collective that orders the streams
    frames = [buf.full.clone()] if rank == 0 else None
    rewards, dones = [], []
    g = torch.Generator(device="cuda").manual_seed(5)
    adim = env.action_dim
    for t in range(steps):
        a_all = torch.randint(0, 4 if adim == 1 else 3, (n_total,) if adim == 1 else (n_total, 2), device="cuda", generator=g, dtype=torch.int32)
        env.step(a_all[lo:hi].contiguous())
        got = buf.fence_with_scalars()
        if rank == 0:
            frames.append(buf.full.clone())
            rewards.append(torch.cat(got[0]).clone())
            dones.append(torch.cat(got[1]).clone())
        else:
            assert got is None
        dist.barrier()  # rank 0 has copied the step's frames before anyone overwrites them
    if rank == 0:
        torch.save(torch.stack(frames).cpu(), out)
        torch.save((torch.stack(rewards).cpu(), torch.stack(dones).cpu()), out + ".scalars")
    env.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
