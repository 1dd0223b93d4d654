#!/usr/bin/env python3
"""tools/soak.py -- long run of one workload (default 20,000 steps of random agents): no device error flags, finite
rewards, episode bookkeeping consistent, memory use flat.  Usage (GPU box): python tools/soak.py ENV_ID [N] [STEPS]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "endless-memory-gym_amd"))
import torch  # noqa: E402

import memory_gym_amd  # noqa: E402

env_id = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
env = memory_gym_amd.make(env_id, num_envs=n, device=0)
env.reset(seed=0)
g = torch.Generator(device="cuda").manual_seed(0)
hi = 4 if env.action_dim == 1 else 3
acts = [torch.randint(0, hi, (n,) if env.action_dim == 1 else (n, 2), device="cuda", generator=g, dtype=torch.int32) for _ in range(64)]
episodes = torch.zeros((), dtype=torch.int64, device="cuda")
ret_sum = torch.zeros((), dtype=torch.float64, device="cuda")
len_sum = torch.zeros((), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
mem0 = torch.cuda.memory_allocated()
t0 = time.perf_counter()
for t in range(steps):
    obs, rew, done, _, info = env.step(acts[t % 64])
    episodes += done.sum()
    ret_sum += torch.where(done, info["reward"], torch.zeros_like(info["reward"])).sum()
    len_sum += torch.where(done, info["length"], torch.zeros_like(info["length"])).sum()
    if t % 2000 == 1999:
        assert torch.isfinite(rew).all()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
env.check_errors()
e = int(episodes.item())
print("%s: %d instances x %d steps in %.1f s (%.1f M env-steps/s incl. the bookkeeping above); %d episodes, mean return %.3f, mean length %.1f; "
      "allocated memory %+d B over the run; no error flags" % (env_id, n, steps, dt, n * steps / dt / 1e6, e, float(ret_sum.item()) / max(e, 1),
                                                                   float(len_sum.item()) / max(e, 1), torch.cuda.memory_allocated() - mem0))
# every step of every instance belongs to exactly one episode: finished ones + the ones still running
assert int(len_sum.item()) <= n * steps
env.close()
