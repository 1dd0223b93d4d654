#!/usr/bin/env python3
"""tools/soak.py -- long run of one workload (default 20,000 steps of random agents): no device error flags, finite
rewards, episode bookkeeping consistent, memory use flat.  Usage (GPU box): python tools/soak.py ENV_ID [N] [STEPS] [POLICY] [ON_CAPACITY]
POLICY "follower:EPS" (ids whose ground truth names the way: Endless-MysteryPath-v0) = a path-following agent with EPS random actions --
the regime in which episodes outlive the build's capacities (128 path segments = 1,024 tiles); ON_CAPACITY "truncate" then counts the
instances whose episode was ended on a capacity (info["capacity_exceeded"]) instead of raising (the default "raise").  Round 6."""
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
policy = sys.argv[4] if len(sys.argv) > 4 else "random"
on_capacity = sys.argv[5] if len(sys.argv) > 5 else "raise"
capacity = dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in sys.argv[6].split(",")) if len(sys.argv) > 6 else None  # e.g. path_segments=1024
env = memory_gym_amd.make(env_id, num_envs=n, device=0, on_capacity=on_capacity, capacity=capacity)
obs, info = env.reset(seed=0)
eps = float(policy.split(":")[1]) if policy.startswith("follower") else None
way = torch.tensor([1.0, 2.0, 3.0], device="cuda")
capacity_ends = torch.zeros((), dtype=torch.int64, device="cuda")
longest = torch.zeros((), dtype=torch.int32, device="cuda")
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
    a = acts[t % 64]
    if eps is not None:
        a = torch.where(torch.rand(n, device="cuda", generator=g) < eps, a, (info["ground_truth"].float() @ way).to(torch.int32))
    obs, rew, done, trunc, info = env.step(a)
    capacity_ends += trunc.sum()
    longest = torch.maximum(longest, torch.where(done, info["length"], torch.zeros_like(info["length"])).max())
    episodes += done.sum()
    ret_sum += torch.where(done, info["reward"], torch.zeros_like(info["reward"])).sum()
    len_sum += torch.where(done, info["length"], torch.zeros_like(info["length"])).sum()
    if t % 2000 == 1999:
        assert torch.isfinite(rew).all()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
env.check_errors()
e = int(episodes.item())
if eps is not None or on_capacity != "raise":
    print("policy %s, on_capacity=%s: %d episodes ended on a capacity of the build (kinds 0x%x), longest finished episode %d steps" % (
        policy, on_capacity, int(capacity_ends.item()), getattr(env, "capacity_events", 0), int(longest.item())))
print("%s: %d instances x %d steps in %.1f s (%.1f M env-steps/s incl. the bookkeeping above); %d episodes, mean return %.3f, mean length %.1f; "
      "allocated memory %+d B over the run; no error flags" % (env_id, n, steps, dt, n * steps / dt / 1e6, e, float(ret_sum.item()) / max(e, 1),
                                                                   float(len_sum.item()) / max(e, 1), torch.cuda.memory_allocated() - mem0))
# every step of every instance belongs to exactly one episode: finished ones + the ones still running
assert int(len_sum.item()) <= n * steps
env.close()
