#!/usr/bin/env python3
"""tools/emp_policy_bench.py [N] [STEPS] [EPS] -- Endless-MysteryPath under an agent that FOLLOWS its path (what a trained policy
does, unlike bench.py's uniform random actions): the action is read off the ground-truth info the environment itself returns
(one-hot: right / up / down to the next path node; endless_mystery_path.py:92-97), with a share EPS of random actions.  Such an
agent appends a segment every ~8 tiles and rarely resets: the regime in which a step's work is new segments, not resets.
Prints env-steps/s and the error flags; MEMGYM_HIP_LIB selects the library (lab switches apply to the lab build)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "endless-memory-gym_amd"))
import memory_gym_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
eps = float(sys.argv[3]) if len(sys.argv) > 3 else 0.02
env = memory_gym_amd.make("Endless-MysteryPath-v0", num_envs=n, device=0)
obs, info = env.reset(seed=torch.arange(n, dtype=torch.int64, device="cuda"))
g = torch.Generator(device="cuda").manual_seed(1)


def act(info):
    a = info["ground_truth"].argmax(1).to(torch.int32) + 1
    r = torch.rand(n, device="cuda", generator=g) < eps
    return torch.where(r, torch.randint(0, 4, (n,), device="cuda", generator=g, dtype=torch.int32), a)


for t in range(300):  # settle
    obs, rew, done, _, info = env.step(act(info))
torch.cuda.synchronize()
ndone = torch.zeros((), dtype=torch.int64, device="cuda")
t0 = time.perf_counter()
for t in range(steps):
    obs, rew, done, _, info = env.step(act(info))
    ndone += done.sum()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
env.check_errors()
print("Endless-MysteryPath-v0 x%d, path-following agent (eps %.2f): %.1f M env-steps/s, %.3f ms per step, %.1f resets per step" % (
    n, eps, n * steps / dt / 1e6, dt / steps * 1e3, float(ndone.item()) / steps))
